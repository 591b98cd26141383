"""B* batch step time against the stage-A sample fraction (sampled group minima -> per-row thresholds)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
N, Dt, Dj, T, K, U = 1048576, 61, 302, 600, 100, 32
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
utts = [synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(U)]
ref = None
for frac in (1 / 8., 1 / 12., 1 / 16., 1 / 24., 1 / 32., 1 / 48.):
    eng = snickery_amd.HipSearchEngine(0)
    eng.set_option('sample_fraction', frac)
    eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
    paths, costs = eng.knn_viterbi_batch(utts, K)
    if ref is None: ref = costs
    eng.reset_timers()
    t0 = time.time()
    for _ in range(3): eng.knn_viterbi_batch(utts, K)
    dt = (time.time() - t0) / 3
    tm = eng.timers()
    st = {k: round(v[0] / 3, 2) for k, v in tm.items() if v[1] and k.startswith('knn')}
    print('1/%d: %.2f ms/step  same=%s  list mean %.0f max %.0f  %s' % (round(1 / frac), dt * 1e3, np.array_equal(ref, costs),
          eng.info('last_list_mean'), eng.info('last_list_max'), st), flush=True)
    eng.close()
