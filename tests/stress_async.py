"""Stress of the batches-in-flight API (three workspaces): random ragged batches, random collect order, compared with
the one-call form.  python tests/stress_async.py [iterations] [seed] [join_bounds_delay]
(join_bounds_delay 3 / 4: the Viterbi side of every group behind a point inside the next group's K-NN call whatever the shape,
a batch's last group queued by the next submit or by its own collect -- the orders this script draws at random)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import snickery_amd
import snk_oracle as o

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
N, Dt, Dj = 40000, 61, 40
F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=9)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
if len(sys.argv) > 3: eng.set_option('join_bounds_delay', int(sys.argv[3]))
pending = []        # (ticket, reference result)
bad = 0
for it in range(n_iter):
    eng.set_option('batch_rows', int(rng.choice([32, 64, 200, 8192])))
    K = int(rng.choice([5, 20, 50]))
    lens = [int(rng.randint(1, 80)) for _ in range(int(rng.randint(1, 7)))]
    utts = [o.synthetic_targets(F_unw, T, seed=int(rng.randint(1 << 30))) * wt for T in lens]
    while len(pending) == 3 or (pending and rng.rand() < 0.3):
        t, ref = pending.pop(int(rng.randint(len(pending))))
        got = eng.knn_viterbi_batch_collect(t)
        ok = all(np.array_equal(a, b) for a, b in zip(got[0], ref[0])) and np.array_equal(got[1], ref[1])
        bad += 0 if ok else 1
    if not pending:
        ref = eng.knn_viterbi_batch(utts, K)              # one-call form only with nothing in flight
    else:
        ref = None
    t = eng.knn_viterbi_batch_submit(utts, K)
    if ref is None:                                      # reference computed after the fact
        got = eng.knn_viterbi_batch_collect(t)
        while pending:
            t2, r2 = pending.pop()
            g2 = eng.knn_viterbi_batch_collect(t2)
            bad += 0 if (all(np.array_equal(a, b) for a, b in zip(g2[0], r2[0])) and np.array_equal(g2[1], r2[1])) else 1
        ref = eng.knn_viterbi_batch(utts, K)
        bad += 0 if (all(np.array_equal(a, b) for a, b in zip(got[0], ref[0])) and np.array_equal(got[1], ref[1])) else 1
    else:
        pending.append((t, ref))
for t, ref in pending:
    got = eng.knn_viterbi_batch_collect(t)
    bad += 0 if (all(np.array_equal(a, b) for a, b in zip(got[0], ref[0])) and np.array_equal(got[1], ref[1])) else 1
print('%d iterations, %d mismatches' % (n_iter, bad))
sys.exit(1 if bad else 0)
