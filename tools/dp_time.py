"""Scratch: standalone stage times of join costs + Viterbi recursion (no concurrent K-NN)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import snickery_amd
from bench import synthetic_db, synthetic_targets

N, Dt, Dj = 262144, 61, 302
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
for T, K in ((600, 100), (600, 50), (600, 128), (600, 200), (1500, 100)):
    U = synthetic_targets(F_unw, T, seed=3) * wt
    cand, dist = eng.knn(U, K)
    eng.viterbi(cand, dist)
    eng.reset_timers()
    for rep in range(5):
        path, cost = eng.viterbi(cand, dist)
    tm = eng.timers()
    print('T=%d K=%d: join %.3f ms  dp %.3f ms (%.2f us/step)' % (T, K, tm['join_costs'][0] / 5, tm['viterbi_dp'][0] / 5,
          tm['viterbi_dp'][0] / 5 / (T - 1) * 1e3))
