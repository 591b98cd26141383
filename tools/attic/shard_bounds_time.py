"""One GPU standing in for the ranks of a row-sharded search, one after the other: step 1 (shard-local
top-K of all rows of the batch) with each shard's own thresholds against step 1 with the bounds shared
over all shards (stage A everywhere, element-wise minimum = what the all-reduce delivers, then filter
+ re-rank against it).  B* database, G shards, 32 utterances per GPU.

    python tools/shard_bounds_time.py [G] [utts_per_gpu]
"""
import os
import sys
import time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from snickery_amd.dist import HipShardEngine, shard_bounds
from bench import synthetic_db, synthetic_targets

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
UPG = int(sys.argv[2]) if len(sys.argv) > 2 else 32
MIN_SLABS = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N, Dt, Dj, T, K = 1048576, 61, 302, 600, 100
F_unw, _ = synthetic_db(N, Dt, 2, seed=0)
wt = np.full(Dt, 0.4)
utts = snickery_amd.QueryBatch([synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(UPG * G)])
R = int(sum(utts.lengths))
dev = torch.device('cuda', 0)
shards = []
for r in range(G):
    lo, hi = shard_bounds(N, G, r)
    eng = snickery_amd.HipSearchEngine(0)
    if MIN_SLABS:
        eng.set_option('min_sample_slabs', MIN_SLABS)
    eng.upload_target_only(F_unw[lo:hi])
    eng.set_shard(lo, N)
    eng.set_weights(wt, None)
    shards.append(HipShardEngine(eng, dev))


def stages(eng):
    tm = eng.timers()
    return {k: round(v[0], 2) for k, v in tm.items() if v[1] and (k.startswith('knn') or k.startswith('h2d') or k.startswith('prep'))}


d2 = torch.empty(R, K, dtype=torch.float64, device=dev)
ids = torch.empty(R, K, dtype=torch.int64, device=dev)
ref = []
e0 = shards[0]
e0.knn_local_batch(utts, K, d2, ids)                        # warm-up
e0.engine.reset_timers()
t0 = time.time(); e0.knn_local_batch(utts, K, d2, ids); torch.cuda.synchronize(); t_plain = time.time() - t0
print('shard 0 of %d, %d rows, own thresholds : %.1f ms  %s  list mean %.0f' % (
    G, R, t_plain * 1e3, stages(e0.engine), e0.engine.info('last_list_mean')), flush=True)
plain = (d2.clone(), ids.clone())

bounds = []
t_bounds = 0.0
for r, e in enumerate(shards):
    b = torch.empty(R, dtype=torch.float64, device=dev)
    if r == 0:
        e.knn_local_batch_bounds(utts, K, b)                # warm-up
        e.engine.reset_timers()
    t0 = time.time(); e.knn_local_batch_bounds(utts, K, b); torch.cuda.synchronize()
    if r == 0:
        t_bounds = time.time() - t0
        st_bounds = stages(e.engine)
    bounds.append(b)
shared = torch.stack(bounds).min(dim=0).values
e0.knn_local_batch_bounds(utts, K, torch.empty(R, dtype=torch.float64, device=dev))      # its own call: the batch is resident on shard 0
e0.knn_local_batch_bounded(utts, K, shared, d2, ids)        # warm-up
e0.engine.reset_timers()
t0 = time.time(); e0.knn_local_batch_bounded(utts, K, shared, d2, ids); torch.cuda.synchronize(); t_bounded = time.time() - t0
print('shard 0: stage A only (bounds)           : %.1f ms  %s' % (t_bounds * 1e3, st_bounds), flush=True)
print('shard 0: filter against shared bounds    : %.1f ms  %s  list mean %.0f' % (
    t_bounded * 1e3, stages(e0.engine), e0.engine.info('last_list_mean')), flush=True)
print('step 1 per rank: %.1f ms -> %.1f ms (+ one all-reduce of %d x 8 bytes)' % (t_plain * 1e3, (t_bounds + t_bounded) * 1e3, R))
# the bounded list of a row is the head of the plain list (same order), the rest padding
pid, bid = plain[1].cpu().numpy(), ids.cpu().numpy()
n_kept = (bid >= 0).sum(axis=1)
ok = all(np.array_equal(bid[t, :n_kept[t]], pid[t, :n_kept[t]]) and np.all(bid[t, n_kept[t]:] == -1) for t in range(0, R, 97))
print('bounded lists are prefixes of the plain lists: %s; entries kept per row: mean %.1f of %d' % (ok, n_kept.mean(), K))
