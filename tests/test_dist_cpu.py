"""world_size-2 gloo test of the sharded search path (snickery_amd/dist.py) on CPU.
The GPU engine is replaced by a stand-in built on the oracle; what is under test is the
sharding arithmetic, the all-gather / all-to-all exchanges, the merge rule and the utterance->rank
assignment of the Viterbi."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardEngine(object):
    """CPU stand-in with the HipShardEngine interface."""
    device = 'cpu'

    def __init__(self, F_shard, offset, E, S):
        self.F, self.offset, self.E, self.S = F_shard, offset, E, S

    def alloc(self, *shape, dtype):
        return torch.empty(*shape, dtype=dtype)

    def knn_local(self, U, K, d2_out, id_out):
        import snk_oracle as o
        cand, d = o.knn_bruteforce(self.F, U, K)
        d2 = np.full(cand.shape, 1e30)
        for t in range(U.shape[0]):
            ok = cand[t] >= 0
            d2[t, ok] = o.sqdist_rows(self.F[cand[t, ok]], U[t])
        ids = np.where(cand >= 0, cand + self.offset, -1)
        d2_out.copy_(torch.from_numpy(d2))
        id_out.copy_(torch.from_numpy(ids))

    def merge(self, d2_all, id_all, G, T, K):
        d2 = d2_all.numpy().reshape(G, T, K)
        ids = id_all.numpy().reshape(G, T, K)
        cand = np.full((T, K), -1, dtype=np.int64)
        dd = np.full((T, K), 1e15)
        for t in range(T):
            k = d2[:, t, :].ravel()
            i = ids[:, t, :].ravel()
            ok = i >= 0
            order = np.lexsort((i[ok], k[ok]))[:K]
            cand[t, :order.size] = i[ok][order]
            dd[t, :order.size] = np.sqrt(k[ok][order])
        return cand, dd

    def viterbi(self, cand, d):
        import snk_oracle as o
        return o.viterbi(cand, d, self.E, self.S)

    def knn_local_batch(self, utterances, K, d2_out, id_out):
        r0 = 0
        for U in utterances:
            self.knn_local(U, K, d2_out[r0:r0 + U.shape[0]], id_out[r0:r0 + U.shape[0]])
            r0 += U.shape[0]

    def knn_local_batch_bounds(self, utterances, K, bound_out):
        # an upper bound of the K-th nearest squared distance of this shard: the exact one, loosened
        import snk_oracle as o
        r0 = 0
        for U in utterances:
            cand, d = o.knn_bruteforce(self.F, U, K)
            for t in range(U.shape[0]):
                ok = cand[t] >= 0
                full = int(ok.sum()) == K
                bound_out[r0 + t] = float(o.sqdist_rows(self.F[cand[t, ok]], U[t]).max()) * 1.5 if full else float('inf')
            r0 += U.shape[0]

    def knn_local_batch_bounded(self, utterances, K, bound_in, d2_out, id_out):
        self.knn_local_batch(utterances, K, d2_out, id_out)
        keep = d2_out <= bound_in.reshape(-1, 1)          # what a shard filtering against the bound returns
        d2_out[~keep] = 1e30
        id_out[~keep] = -1

    def merge_viterbi_batch(self, d2_all, id_all, G, lengths, K):
        R = sum(lengths)
        cand, dd = self.merge(d2_all, id_all, G, R, K)
        paths, costs, r0 = [], [], 0
        for T in lengths:
            p, c = self.viterbi(cand[r0:r0 + T], dd[r0:r0 + T])
            paths.append(p)
            costs.append(c)
            r0 += T
        return paths, costs


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import snk_oracle as o
    from snickery_amd.dist import ShardedSearch, shard_bounds
    N, Dt, Dj, K = 3001, 20, 12, 9
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=3)
    wt = np.full(Dt, 0.5)
    wj = np.full(Dj, 0.2)
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    lo, hi = shard_bounds(N, world, rank)
    eng = OracleShardEngine(F[lo:hi], lo, E, S)
    search = ShardedSearch(eng)
    utts = [o.synthetic_targets(F_unw, T, seed=s) * wt for s, T in [(1, 14), (2, 9), (3, 11)]]
    cand, d = search.knn(utts[0], K)
    paths, costs = search.knn_viterbi_batch(utts, K)
    oc, od = o.knn_bruteforce(F, utts[0], K)
    ok = bool(np.array_equal(cand, oc) and np.array_equal(d, od))
    for u, U in enumerate(utts):
        c, dd = o.knn_bruteforce(F, U, K)
        p, cost = o.viterbi(c, dd, E, S)
        ok = ok and list(paths[u]) == p and costs[u] == cost
    # fewer utterances than ranks: the last rank owns nothing and still takes part in the exchange
    paths1, costs1 = search.knn_viterbi_batch(utts[1:2], K)
    c, dd = o.knn_bruteforce(F, utts[1], K)
    p, cost = o.viterbi(c, dd, E, S)
    ok = ok and len(paths1) == 1 and list(paths1[0]) == p and costs1[0] == cost
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover():
    sys.path.insert(0, ROOT)
    from snickery_amd.dist import shard_bounds
    for n, g in [(10, 3), (1048576, 8), (7, 8), (1001, 2)]:
        spans = [shard_bounds(n, g, r) for r in range(g)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(g - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_sharded_search_gloo_world2():
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] is True and out[1] is True


def test_library_shard_plan_is_the_twin_of_dist_shard_bounds():
    """snk_shard_plan (the split snk_sharded_knn_viterbi_batch uses for database rows and utterance blocks)
    against dist.shard_bounds (the split of the torch-side path): same blocks for every (n, world, rank),
    so the two exchange paths move the same rows to the same owners.  No GPU: a pure function of the C ABI."""
    import snickery_amd
    from snickery_amd.dist import shard_bounds
    from snickery_amd.engine import shard_plan
    for n in (0, 1, 2, 7, 8, 31, 32, 100, 1048576, 1500001):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_plan(n, world, r) for r in range(world)]
            assert blocks == [shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
    with pytest.raises(snickery_amd.SnkError):
        shard_plan(10, 2, 2)
