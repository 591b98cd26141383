"""One-call batches against two batches in flight (submit i+1 before collect i) on the B* workload."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
import snickery_amd
from bench import synthetic_db, synthetic_targets
N, Dt, Dj, T, K, U = 1048576, 61, 302, 600, 100, 32
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
batches = [snickery_amd.QueryBatch([synthetic_targets(F_unw, T, seed=100 * b + u) * wt for u in range(U)]) for b in range(3)]
if len(sys.argv) > 1 and sys.argv[1] == 'pin':
    for b in batches: b.pin()
ref = [eng.knn_viterbi_batch(b, K) for b in batches]
steps = 8
t0 = time.time()
for i in range(steps): eng.knn_viterbi_batch(batches[i % 3], K)
t_sync = (time.time() - t0) / steps
t0 = time.time()
prev = None; got = []
for i in range(steps):
    t = eng.knn_viterbi_batch_submit(batches[i % 3], K)
    if prev is not None: got.append(eng.knn_viterbi_batch_collect(prev))
    prev = t
got.append(eng.knn_viterbi_batch_collect(prev))
t_async = (time.time() - t0) / steps
ok = all(all(np.array_equal(a, b) for a, b in zip(got[i][0], ref[i % 3][0])) and np.array_equal(got[i][1], ref[i % 3][1]) for i in range(steps))
print('sync %.2f ms/step (%.0f frames/s), pipelined %.2f ms/step (%.0f frames/s), identical results: %s' % (
    t_sync * 1e3, U * T / t_sync, t_async * 1e3, U * T / t_async, ok))
