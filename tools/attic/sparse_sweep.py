"""Margin sweep of the sparse Viterbi path on the B* workload: refinement statistics and stage times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import snickery_amd
from bench import synthetic_db, synthetic_targets

N, Dt, Dj, T, K, U = 1048576, 61, 302, 600, 100, 32
F, JC = synthetic_db(N, Dt, Dj, 0)
wt, wj = np.full(Dt, 0.4), np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F, JC); eng.set_weights(wt, wj)
utts = [synthetic_targets(F, T, 1 + u) * wt for u in range(U)]
batch = snickery_amd.QueryBatch(utts)
names = ('dense_cells', 'dense_steps', 'dense_exact_costs', 'set_overflows')
ref = None
for beta in [float(b) for b in (sys.argv[1:] or ['2e-3', '1e-3', '5e-4', '2e-4', '0'])]:
    eng.set_option('join_beta', beta)
    eng.knn_viterbi_batch(batch, K)
    before = [eng.info(n) for n in names]
    eng.reset_timers()
    t0 = time.perf_counter()
    for _ in range(3):
        paths, costs = eng.knn_viterbi_batch(batch, K)
    dt = (time.perf_counter() - t0) / 3
    st = [(eng.info(n) - b) / 3 for n, b in zip(names, before)]
    tm = eng.timers()
    if ref is None:
        ref = (paths, costs)
    same = all(np.array_equal(a, b) for a, b in zip(paths, ref[0])) and np.array_equal(costs, ref[1])
    print('beta %g: %.2f ms/step  cells refined %.0f (%.2f%%)  steps %.0f (%.1f%%)  exact in refinement %.0f  overflows %.0f  same=%s' % (
        beta, dt * 1e3, st[0], 100 * st[0] / (U * T * K), st[1], 100 * st[1] / (U * T), st[2], st[3], same))
    print('   ', {k: round(v[0] / 3, 2) for k, v in tm.items() if v[1] and k in ('join_lower_bounds', 'viterbi_lower_bound', 'join_exact_sparse', 'viterbi_sparse', 'knn_filter')})
