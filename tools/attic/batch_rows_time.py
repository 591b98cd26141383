"""Scratch: B* batch throughput against the batch_rows option (rows per K-NN call)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import snickery_amd
from bench import synthetic_db, synthetic_targets

N, Dt, Dj, T, K, U = 1048576, 61, 302, 600, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 16
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
utts = [synthetic_targets(F_unw, T, seed=s) * wt for s in range(1, U + 1)]
ref = None
for rows, nst in ((0, 2), (600, 2), (1200, 2), (2400, 2), (3600, 2), (4800, 2), (8192, 2)):
    eng.set_option('batch_rows', rows)
    eng.knn_viterbi_batch(utts, K)
    eng.reset_timers()
    t0 = time.time()
    for rep in range(3):
        paths, costs = eng.knn_viterbi_batch(utts, K)
    dt = (time.time() - t0) / 3
    if ref is None: ref = (paths, costs)
    same = all(np.array_equal(a, b) for a, b in zip(paths, ref[0])) and np.array_equal(costs, ref[1])
    print('batch_rows=%5d dp_streams=%d: %.2f ms/step  %.0f frames/s  same=%s redos=%d' % (rows, nst, dt * 1e3, U * T / dt, same, eng.info('batch_redos')))
    print('    ' + '  '.join('%s %.3f/%d' % (k, ms, n) for k, (ms, n) in eng.timers().items() if n))
