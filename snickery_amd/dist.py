"""Multi-GPU search: the unit database is row-sharded over the ranks of one node
(one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Exchange step (the only collective on the data path): every rank computes the
top-K of ALL query rows against ITS shard, the per-rank lists (T, K) of (squared distance,
global unit id) are all-gathered, and each rank merges the G*K candidates of every row to the
global top-K ordered by (distance, id).  The Viterbi of an utterance then runs on ONE GPU
(utterance u on rank u mod G) against a replicated join matrix.

The class is engine-agnostic: the product passes a ``snickery_amd.HipSearchEngine`` and CUDA
tensors; the CPU tests (gloo, world_size 2) pass a stand-in engine built on the oracle.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_units, world_size, rank):
    """Contiguous row shard [lo, hi) of rank; sizes differ by at most one unit."""
    base, rem = divmod(int(n_units), int(world_size))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class HipShardEngine(object):
    """Adapter: HipSearchEngine + torch CUDA buffers for the exchange step."""

    def __init__(self, engine, device):
        self.engine = engine
        self.device = device

    def alloc(self, *shape, dtype):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def knn_local(self, U, K, d2_out, id_out):
        self.engine.knn_local_dev(U, K, d2_out.data_ptr(), id_out.data_ptr())

    def merge(self, d2_all, id_all, G, T, K):
        torch.cuda.synchronize(self.device)
        return self.engine.merge_topk_dev(d2_all.data_ptr(), id_all.data_ptr(), G, T, K)

    def viterbi(self, cand, dist_):
        return self.engine.viterbi(cand, dist_)


class ShardedSearch(object):
    def __init__(self, shard_engine, rank=None, world_size=None, group=None):
        self.e = shard_engine
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world_size is None else world_size

    def knn(self, U, K):
        """Global (candidates, distances) of every row of U; identical on all ranks."""
        T = U.shape[0]
        G = self.world
        d2 = self.e.alloc(T, K, dtype=torch.float64)
        ids = self.e.alloc(T, K, dtype=torch.int64)
        self.e.knn_local(U, K, d2, ids)
        if G == 1:
            return self.e.merge(d2, ids, 1, T, K)
        d2_all = self.e.alloc(G, T, K, dtype=torch.float64)
        id_all = self.e.alloc(G, T, K, dtype=torch.int64)
        # one fused gather each for distances and ids: (T*K*8 B per rank, latency bound)
        dist.all_gather(list(d2_all.unbind(0)), d2, group=self.group)
        dist.all_gather(list(id_all.unbind(0)), ids, group=self.group)
        return self.e.merge(d2_all, id_all, G, T, K)

    def knn_viterbi_batch(self, utterances, K):
        """Paths of all utterances (list of int64 arrays) and costs, gathered on every rank.
        K-NN is sharded over the database; utterance u's Viterbi runs on rank u mod G."""
        G = self.world
        mine = {}
        for u, U in enumerate(utterances):
            cand, d = self.knn(U, K)
            if u % G == self.rank:
                path, cost = self.e.viterbi(cand, d)
                mine[u] = (np.asarray(path, dtype=np.int64), float(cost))
        if G == 1:
            return [mine[u][0] for u in range(len(utterances))], np.array([mine[u][1] for u in range(len(utterances))])
        # results exchange (small, host side): pad paths to a common length
        lens = [int(np.shape(U)[0]) for U in utterances]
        Lmax = max(lens)
        buf = torch.full((len(utterances), Lmax + 2), -1.0, dtype=torch.float64)
        for u, (p, c) in mine.items():
            buf[u, 0] = float(len(p))
            buf[u, 1] = c
            buf[u, 2:2 + len(p)] = torch.from_numpy(p.astype(np.float64))
        buf = buf.to(self.e.device) if str(self.e.device) != 'cpu' else buf
        allb = [torch.empty_like(buf) for _ in range(G)]
        dist.all_gather(allb, buf, group=self.group)
        paths, costs = [], []
        for u in range(len(utterances)):
            row = allb[u % G][u].cpu().numpy()
            n = int(row[0])
            paths.append(row[2:2 + n].astype(np.int64))
            costs.append(row[1])
        return paths, np.array(costs)
