"""snk_sharded_knn_viterbi_batch on a single-rank RCCL communicator (the whole database on this GPU; the collectives
degenerate to copies): one step at a time against two steps in flight (submit / collect) -- what running the Viterbi
side of step i beside the K-NN of step i + 1 is worth in the sharded entry point; next to the plain batch pipeline."""
import sys, os, time
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
utts = snickery_amd.QueryBatch([synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(U)])
utts.pin()
ref = eng.knn_viterbi_batch(utts, K)
eng.comm_init(1, 0, eng.comm_unique_id())
steps = 20
p, c = eng.sharded_knn_viterbi_batch(utts, K)
t0 = time.perf_counter()
for _ in range(steps):
    p, c = eng.sharded_knn_viterbi_batch(utts, K)
dt1 = (time.perf_counter() - t0) / steps
same = all(np.array_equal(a, b) for a, b in zip(p, ref[0])) and np.array_equal(c, ref[1])
print('sharded, one step at a time: %.2f ms per step  same=%s' % (dt1 * 1e3, same))
t0 = time.perf_counter()
pending = None
for _ in range(steps):
    tk = eng.sharded_knn_viterbi_batch_submit(utts, K)
    if pending is not None:
        p, c = eng.sharded_knn_viterbi_batch_collect(pending)
    pending = tk
p, c = eng.sharded_knn_viterbi_batch_collect(pending)
dt2 = (time.perf_counter() - t0) / steps
same = all(np.array_equal(a, b) for a, b in zip(p, ref[0])) and np.array_equal(c, ref[1])
print('sharded, two steps in flight: %.2f ms per step  same=%s' % (dt2 * 1e3, same))
eng.comm_destroy()
t0 = time.perf_counter()
pending = None
for _ in range(steps):
    tk = eng.knn_viterbi_batch_submit(utts, K)
    if pending is not None:
        eng.knn_viterbi_batch_collect(pending)
    pending = tk
eng.knn_viterbi_batch_collect(pending)
print('snk_knn_viterbi_batch_submit / _collect, two in flight: %.2f ms per step' % ((time.perf_counter() - t0) / steps * 1e3))
