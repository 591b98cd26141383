"""Small driver for PMC passes: two B* batch steps (32 utterances) through the batch entry point."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
N, Dt, Dj, T, K, U = 1048576, 61, 302, 600, 100, 32
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
utts = [synthetic_targets(F_unw, T, seed=1 + s) * wt for s in range(U)]
for s in range(2):
    eng.knn_viterbi_batch(utts, K)
print(eng.timers())
