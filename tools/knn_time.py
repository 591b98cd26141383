"""K-NN stages alone (nothing else on the GPU): calls of `rows` query rows against the B* database, with and without the
ball bound of the thresholds (stage A', prefilter_ball_bound), per prefilter; results compared with the first configuration.
    python tools/knn_time.py [rows]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
N, Dt, Dj, K = 1048576, 61, 302, 100
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 9600
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
U = np.vstack([synthetic_targets(F_unw, 600, seed=1 + s) * wt for s in range((rows + 599) // 600)])[:rows]
ref = None
for pre, bb, sb in ((0, 0, 1), (1, 0, 0), (1, 0, 1), (1, 1, 1)):
    eng = snickery_amd.HipSearchEngine(0)
    eng.set_option('prefilter', pre)
    eng.set_option('prefilter_ball_bound', bb)
    eng.set_option('prefilter_super_balls', sb)
    eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
    cand, dist = eng.knn(U, K)
    if ref is None: ref = (cand, dist)
    eng.reset_timers()
    t0 = time.time()
    for _ in range(3): eng.knn(U, K)
    dt = (time.time() - t0) / 3
    tm = eng.timers()
    st = {k: round(v[0] / 3, 3) for k, v in tm.items() if v[1]}
    print('prefilter %d ball bound %d super balls %d: %.2f ms/call  same=%s  list mean %.0f max %.0f  %s' % (pre, bb, sb, dt * 1e3,
          np.array_equal(ref[0], cand) and np.array_equal(ref[1], dist), eng.info('last_list_mean'), eng.info('last_list_max'), st),
          'pairs', eng.info('coarse_pairs'), 'filter_coarse', eng.info('filter_coarse'),
          'margin rows', eng.info('prefilter_margin_rows'), 'min margin', eng.info('prefilter_min_margin'),
          'fallbacks', eng.info('f16_fallbacks'), flush=True)
    eng.close()
