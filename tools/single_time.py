"""One utterance through snk_knn_viterbi at the B* database (T = 600, K = 100): wall time per call and the stage table,
per Viterbi mode (2 = automatic, 1 = sparse kernels forced, 0 = dense)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
N, Dt, Dj, K = 1048576, 61, 302, 100
T = int(sys.argv[1]) if len(sys.argv) > 1 else 600
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
U = synthetic_targets(F_unw, T, seed=1) * wt
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
ref = None
import itertools
for mode, waves, chunk, warm in [(0, 1, 0, 32), (1, 4, 0, 32), (1, 1, 0, 32)] + [(1, 1, c, w) for c in (32, 48, 64, 96) for w in (8, 16, 32, 48)]:
    eng.set_option('viterbi_mode', mode)
    eng.set_option('viterbi_sparse_waves', waves)
    eng.set_option('viterbi_lb_chunk', chunk)
    eng.set_option('viterbi_lb_warm', warm)
    out = eng.knn_viterbi(U, K)
    if ref is None: ref = out
    for _ in range(3): eng.knn_viterbi(U, K)
    eng.reset_timers()
    st0 = [eng.info(x) for x in ('dense_cells', 'dense_steps', 'dense_exact_costs', 'set_overflows')]
    n = 20
    t0 = time.time()
    for _ in range(n): out = eng.knn_viterbi(U, K)
    dt = (time.time() - t0) / n
    tm = eng.timers()
    st = {k: round(v[0] / n, 3) for k, v in tm.items() if v[1]}
    same = all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(out, ref))
    st1 = [eng.info(x) for x in ('dense_cells', 'dense_steps', 'dense_exact_costs', 'set_overflows')]
    print('viterbi_mode %d waves %d chunk %d warm %d: %.3f ms/call same=%s %s' % (mode, waves, chunk, warm, dt * 1e3, same, st), 'per call: refined cells %.1f, steps with a refinement %.1f, exact costs there %.1f, set overflows %.1f' % tuple((b - a) / n for a, b in zip(st0, st1)), flush=True)
eng.close()
