"""Standalone stage times of the Viterbi side (no concurrent K-NN): dense exact path and sparse path."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import snickery_amd
from bench import synthetic_db, synthetic_targets

N, Dt, Dj = 1048576, 61, 302
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
if len(sys.argv) > 1:
    eng.set_option('join_beta', float(sys.argv[1]))
for T, K in ((600, 100), (600, 50), (600, 128), (600, 200), (1500, 100)):
    U = synthetic_targets(F_unw, T, seed=3) * wt
    cand, dist = eng.knn(U, K)
    for mode in (0, 1):
        eng.set_option('viterbi_mode', mode)
        eng.viterbi(cand, dist)
        eng.reset_timers()
        st0 = [eng.info(n) for n in ('dense_cells', 'dense_steps', 'dense_exact_costs')]
        t0 = time.perf_counter()
        for rep in range(5):
            path, cost = eng.viterbi(cand, dist)
        wall = (time.perf_counter() - t0) / 5 * 1e3
        tm = eng.timers()
        keys = ('join_costs', 'viterbi_dp') if mode == 0 else ('join_lower_bounds', 'viterbi_lower_bound', 'join_exact_sparse', 'viterbi_sparse')
        st = [(eng.info(n) - b) / 5 for n, b in zip(('dense_cells', 'dense_steps', 'dense_exact_costs'), st0)]
        print('T=%d K=%d mode %d: wall %.3f ms  ' % (T, K, mode, wall) + '  '.join('%s %.3f' % (k, tm[k][0] / 5) for k in keys) + ('  refined cells/steps/costs %s' % st if mode else ''))
