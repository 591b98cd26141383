"""Scratch timing of the shard-local K-NN (one rank of an 8-way row-sharded B* database) on one GPU:
per-utterance calls vs one call over the concatenated rows of the whole batch."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import snickery_amd
from bench import synthetic_db, synthetic_targets

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, Dt, Dj, T, K, U = 1048576, 61, 302, 600, 100, 16
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4)
eng = snickery_amd.HipSearchEngine(0)
lo, hi = 0, N // G
eng.upload_target_only(F_unw[lo:hi]); eng.set_shard(lo, N); eng.set_weights(wt, None)
utts = [synthetic_targets(F_unw, T, seed=s) * wt for s in range(1, U + 1)]
allq = np.vstack(utts)
dev = torch.device('cuda', 0)
d2 = torch.empty(U * T, K, dtype=torch.float64, device=dev)
ids = torch.empty(U * T, K, dtype=torch.int64, device=dev)
for rep in range(2):
    eng.reset_timers()
    torch.cuda.synchronize(); t0 = time.time()
    for u, q in enumerate(utts):
        eng.knn_local_dev(q, K, d2[u * T:(u + 1) * T].data_ptr(), ids[u * T:(u + 1) * T].data_ptr())
    torch.cuda.synchronize(); dt = time.time() - t0
print('per-utterance calls: %.3f ms for %d utts' % (dt * 1e3, U))
for k, (ms, n) in eng.timers().items():
    if n: print('   %-18s %8.3f ms total over %d' % (k, ms, n))
ref_d2, ref_ids = d2.clone(), ids.clone()
for rows in (2400, 9600):
    for rep in range(2):
        eng.reset_timers()
        torch.cuda.synchronize(); t0 = time.time()
        for s in range(0, U * T, rows):
            eng.knn_local_dev(allq[s:s + rows], K, d2[s:s + rows].data_ptr(), ids[s:s + rows].data_ptr())
        torch.cuda.synchronize(); dt = time.time() - t0
    print('%d-row calls: %.3f ms   same=%s retries=%d f32fallbacks=%d' % (rows, dt * 1e3, bool(torch.equal(d2, ref_d2) and torch.equal(ids, ref_ids)),
          eng.info('last_knn_retries'), eng.info('f16_fallbacks')))
    for k, (ms, n) in eng.timers().items():
        if n: print('   %-18s %8.3f ms total over %d' % (k, ms, n))

# the batch entry points: step 1 on this "rank", step 2 for the 16/G utterances it would own
eng.upload_join_only(JC_unw); eng.set_weights(wt, np.full(Dj, 0.05))
for rep in range(3):
    eng.reset_timers()
    torch.cuda.synchronize(); t0 = time.time()
    eng.knn_local_batch_dev(utts, K, d2.data_ptr(), ids.data_ptr())
    t1 = time.time()
print('knn_local_batch_dev: %.3f ms  same=%s redos=%d' % ((t1 - t0) * 1e3, bool(torch.equal(d2, ref_d2) and torch.equal(ids, ref_ids)), eng.info('batch_redos')))
for k, (ms, n) in eng.timers().items():
    if n: print('   %-18s %8.3f ms total over %d' % (k, ms, n))
n_own = max(1, U // G)
fake = torch.stack([d2[:n_own * T]] + [d2[:n_own * T] + 1e3] * (G - 1)).contiguous()
fid = torch.stack([ids[:n_own * T]] * G).contiguous()
for rep in range(3):
    eng.reset_timers()
    torch.cuda.synchronize(); t0 = time.time()
    paths, costs = eng.merge_viterbi_batch_dev(fake.data_ptr(), fid.data_ptr(), G, [T] * n_own, K)
    t1 = time.time()
print('merge_viterbi_batch_dev (%d utts): %.3f ms' % (n_own, (t1 - t0) * 1e3))
for k, (ms, n) in eng.timers().items():
    if n: print('   %-18s %8.3f ms total over %d' % (k, ms, n))
