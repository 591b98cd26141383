import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
N, Dt, Dj, T, K = 1048576, 61, 302, 600, 100
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
U = synthetic_targets(F_unw, T, seed=1) * wt
for nt in (4, 8):
    eng.set_option('db_tiles_per_wave', nt)
    for dbg in (0, 1, 2):
        eng.set_option('sweep_debug', dbg)
        try:
            eng.knn(U, K)
        except Exception as e:
            pass
        eng.reset_timers()
        for _ in range(3):
            try:
                eng.knn(U, K)
            except Exception as e:
                print('   exc:', str(e)[:150])
        t = eng.timers()
        print('NT=%d dbg=%d filter %.3f ms  minima %.3f ms' % (nt, dbg, t['knn_filter'][0] / max(t['knn_filter'][1], 1), t['knn_minima'][0] / max(t['knn_minima'][1], 1)))
eng.set_option('sweep_debug', 0)
