"""K-NN stages alone (nothing else on the GPU): one call of 6400 query rows against the B* database, per stage-A
sample fraction and prefilter."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
N, Dt, Dj, K = 1048576, 61, 302, 100
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
U = synthetic_targets(F_unw, rows, seed=1) * wt
ref = None
configs = ((0, 1 / 16.), (1, 1 / 16.), (1, 1 / 8.), (1, 1 / 4.))
if len(sys.argv) > 2:
    configs = ((int(sys.argv[2]), 1.0 / float(sys.argv[3]) if len(sys.argv) > 3 else 1 / 16.),)
for pre, frac in configs:
    eng = snickery_amd.HipSearchEngine(0)
    eng.set_option('sample_fraction', frac)
    eng.set_option('prefilter', pre)
    eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
    cand, dist = eng.knn(U, K)
    if ref is None: ref = (cand, dist)
    eng.reset_timers()
    t0 = time.time()
    for _ in range(3): eng.knn(U, K)
    dt = (time.time() - t0) / 3
    tm = eng.timers()
    st = {k: round(v[0] / 3, 3) for k, v in tm.items() if v[1]}
    print('prefilter %d sample 1/%d: %.2f ms/call  same=%s  list mean %.0f max %.0f  %s' % (pre, round(1 / frac), dt * 1e3,
          np.array_equal(ref[0], cand) and np.array_equal(ref[1], dist), eng.info('last_list_mean'), eng.info('last_list_max'), st),
          'pairs', eng.info('coarse_pairs'), 'filter_coarse', eng.info('filter_coarse'), flush=True)
    eng.close()
    # (the last engine's tile-pair counts of the two-pass filter)
