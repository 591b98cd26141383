"""Small driver for PMC passes: a few full-size K-NN calls only."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
N, Dt, Dj, T, K = 1048576, 61, 302, 600, 100
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 8
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
eng.set_option('db_tiles_per_wave', nt)
for s in range(3):
    U = synthetic_targets(F_unw, T, seed=1 + s) * wt
    eng.knn_viterbi(U, K)
print(eng.timers())
