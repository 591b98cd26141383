"""The one-pass three-term sweep (knn_sweep16b<filter>: north_star's "dense GEMM + top-K" kernel, what a voice with units in no
order runs) alone on the GPU: B* database as generated and permuted, `rows` query rows per call; time per launch, what the bf16
pipe issues (3 terms x 2 rows N 64) against its dense peak, list lengths; results compared between the filters.
    python tools/onepass_time.py [rows]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets, variant_database, BF16_MFMA_PEAK_TFLOPS
N, Dt, Dj, K = 1048576, 61, 8, 100
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 9600
F0, JC0 = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
for kind in ('compact', 'permuted', 'speechlike'):
    F_unw, JC_unw = (F0, JC0) if kind == "compact" else variant_database(kind, N, Dt, F0, JC0)[:2]
    # a permuted voice keeps the speech its units were cut from: the rows follow the walk through the ORIGINAL order
    src = F0 if kind == 'permuted' else F_unw
    U = np.vstack([synthetic_targets(src, 600, seed=1 + s) * wt for s in range((rows + 599) // 600)])[:rows]
    ref = None
    for two_pass in (0, 1):
        eng = snickery_amd.HipSearchEngine(0)
        eng.set_option('prefilter_two_pass', two_pass)
        eng.set_option('reorder', 0)          # the kernel on the voice AS STORED (the engine would otherwise reorder a permuted one)
        eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
        for _ in range(4):
            cand, dist = eng.knn(U, K)
        if ref is None: ref = (cand, dist)
        eng.reset_timers()
        for _ in range(5): eng.knn(U, K)
        tm = eng.timers()
        f = tm['knn_filter'][0] / tm['knn_filter'][1]
        issued = 3 * 2.0 * rows * N * 64 / (f * 1e-3) / 1e12
        st = {k: round(v[0] / v[1], 3) for k, v in tm.items() if v[1]}
        # the issued figure prices the ONE-pass sweep (3 terms over every pair): printed only where that is what ran -- a two-pass
        # filter skips most pairs, and its time against the one-pass work is no fraction of a peak (r06's first log printed 1.08)
        ran_onepass = two_pass == 0 or eng.info('filter_onepass') == 1
        what = ('one-pass issued %.0f TFLOP/s = %.3f of the bf16 peak' % (issued, issued / BF16_MFMA_PEAK_TFLOPS)) if ran_onepass else \
               'two-pass filter (%s): most tile pairs skipped, no one-pass figure' % ('coarse sweep + refine' if eng.info('filter_coarse') == 1 else 'balls + refine')
        print('%s two_pass %d: filter %.3f ms per launch of %d rows (%s)  coarse %d onepass %d  same=%s  list mean %.0f  %s' % (
            kind, two_pass, f, rows, what, eng.info('filter_coarse'), eng.info('filter_onepass'),
            np.array_equal(ref[0], cand) and np.array_equal(ref[1], dist), eng.info('last_list_mean'), st), flush=True)
        eng.close()
