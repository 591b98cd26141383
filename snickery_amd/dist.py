"""Multi-GPU search: the unit database is row-sharded over the ranks of one node
(one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Exchange step (the only collective on the data path): every rank computes the
top-K of ALL query rows against ITS shard; the per-rank lists (T, K) of (squared distance,
global unit id) are exchanged and merged to the global top-K ordered by (distance, id).  A single
utterance all-gathers its lists (every rank gets the candidates); a batch assigns utterances to
ranks in contiguous blocks and uses an all-to-all, so that every list travels only to the ONE GPU
that runs that utterance's Viterbi against a replicated join matrix.

The class is engine-agnostic: the product passes a ``snickery_amd.HipSearchEngine`` and CUDA
tensors; the CPU tests (gloo, world_size 2) pass a stand-in engine built on the oracle.
"""
import numpy as np
import torch
import torch.distributed as dist


def _staged(group):
    """True when the collectives must go through host memory: backend gloo with device buffers.
    Only the functional tests do that (several ranks sharing the one GPU of a test box, where RCCL
    refuses two ranks on one device); production runs nccl (= RCCL) on the device buffers."""
    return dist.get_backend(group) == 'gloo'


def _all_to_all(out, inp, out_splits, in_splits, group):
    if _staged(group) and out.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)


def _all_gather(outs, inp, group):
    if _staged(group) and inp.is_cuda:
        tmp = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        dist.all_gather(tmp, inp.cpu(), group=group)
        for o, t in zip(outs, tmp):
            o.copy_(t)
    else:
        dist.all_gather(outs, inp, group=group)


def _all_reduce_min(t, group):
    if _staged(group) and t.is_cuda:
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.MIN, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)


def _lengths(utterances):
    """Row counts of a list of per-utterance matrices or of a snickery_amd.QueryBatch."""
    if hasattr(utterances, 'lengths'):
        return list(utterances.lengths)
    return [int(np.shape(U)[0]) for U in utterances]


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

    def knn_local_batch(self, utterances, K, d2_out, id_out):
        self.engine.knn_local_batch_dev(utterances, K, d2_out.data_ptr(), id_out.data_ptr())

    def knn_local_batch_bounds(self, utterances, K, bound_out):
        self._lengths = _lengths(utterances)
        _, self._columns = self.engine.knn_local_batch_bounds_dev(utterances, K, bound_out.data_ptr())

    def knn_local_batch_bounded(self, utterances, K, bound_in, d2_out, id_out):
        torch.cuda.synchronize(self.device)          # the all-reduce of the bounds ran on torch's stream
        self.engine.knn_local_batch_bounded_dev(self._lengths, self._columns, K, bound_in.data_ptr(),
                                                d2_out.data_ptr(), id_out.data_ptr())

    def merge_viterbi_batch(self, d2_all, id_all, G, lengths, K):
        torch.cuda.synchronize(self.device)          # the exchange ran on torch's stream
        return self.engine.merge_viterbi_batch_dev(d2_all.data_ptr(), id_all.data_ptr(), G, lengths, K)


class ShardedSearch(object):
    def __init__(self, shard_engine, rank=None, world_size=None, group=None, shared_bounds=True):
        self.e = shard_engine
        self.group = group
        self.shared_bounds = shared_bounds
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
        _all_gather(list(d2_all.unbind(0)), d2, self.group)
        _all_gather(list(id_all.unbind(0)), ids, self.group)
        return self.e.merge(d2_all, id_all, G, T, K)

    def knn_viterbi_batch(self, utterances, K):
        """Paths of all utterances (list of int64 arrays) and costs, gathered on every rank.

        Step 1 (every rank): shard-local top-K of ALL rows of the batch.  Exchange: utterances are
        owned by ranks in contiguous blocks, so the (R, K) list matrices are already ordered by
        destination and ONE all-to-all per matrix hands every owner the lists of its utterances
        from every shard as (G, R_own, K) -- (G-1)/G * R*K*16 B sent per rank instead of the
        G-fold all-gather volume.  Step 2 (owner): merge, join costs, Viterbi."""
        G = self.world
        n = len(utterances)
        lens = _lengths(utterances)
        owned = [shard_bounds(n, G, r) for r in range(G)]          # utterance blocks per rank
        rows_to = [sum(lens[a:b]) for a, b in owned]
        R = sum(lens)
        d2 = self.e.alloc(R, K, dtype=torch.float64)
        ids = self.e.alloc(R, K, dtype=torch.int64)
        if G > 1 and self.shared_bounds:
            # every shard bounds the K-th nearest key of ITS units from a sample; the smallest of
            # those bounds still bounds the K-th nearest key of the whole database, and filtering every
            # shard against it leaves about 1/G of the survivors (R x 8 bytes, one all-reduce)
            bound = self.e.alloc(R, dtype=torch.float64)
            self.e.knn_local_batch_bounds(utterances, K, bound)
            _all_reduce_min(bound, self.group)
            self.e.knn_local_batch_bounded(utterances, K, bound, d2, ids)
        else:
            self.e.knn_local_batch(utterances, K, d2, ids)
        lo, hi = owned[self.rank]
        r_own = rows_to[self.rank]
        if G == 1:
            d2_all, id_all = d2, ids
        else:
            d2_all = self.e.alloc(G * r_own, K, dtype=torch.float64)
            id_all = self.e.alloc(G * r_own, K, dtype=torch.int64)
            _all_to_all(d2_all, d2, [r_own] * G, rows_to, self.group)
            _all_to_all(id_all, ids, [r_own] * G, rows_to, self.group)
        if hi > lo:
            own_paths, own_costs = self.e.merge_viterbi_batch(d2_all, id_all, G, lens[lo:hi], K)
        else:
            own_paths, own_costs = [], []
        if G == 1:
            return [np.asarray(p, dtype=np.int64) for p in own_paths], np.asarray(own_costs, dtype=np.float64)
        # results exchange (small): one fixed-size record per owned-utterance slot
        slots = max(b - a for a, b in owned)
        Lmax = max(lens)
        buf = torch.full((slots, Lmax + 2), -1.0, dtype=torch.float64)
        for j, (p, c) in enumerate(zip(own_paths, own_costs)):
            buf[j, 0] = float(len(p))
            buf[j, 1] = float(c)
            buf[j, 2:2 + len(p)] = torch.from_numpy(np.asarray(p, dtype=np.float64))
        buf = buf.to(self.e.device) if str(self.e.device) != 'cpu' else buf
        allb = [torch.empty_like(buf) for _ in range(G)]
        _all_gather(allb, buf, self.group)
        allb = [b.cpu().numpy() for b in allb]
        paths, costs = [], []
        for r, (a, b) in enumerate(owned):
            for j in range(b - a):
                row = allb[r][j]
                paths.append(row[2:2 + int(row[0])].astype(np.int64))
                costs.append(row[1])
        return paths, np.array(costs)


# ------------------------------------------------------------------------------------------------
# The same sharded search with the collectives INSIDE libsnkhip.so (include/snk.h: snk_comm_init,
# snk_sharded_knn_viterbi_batch): RCCL on the engine's own stream, no host synchronisation between the
# bounds, the exchange and the merge.  torch.distributed only carries the 128-byte RCCL id to the ranks.
# ------------------------------------------------------------------------------------------------
def gloo_transport(world_size, group=None):
    """A snickery_amd.engine.TransportCallbacks over torch.distributed collectives on HOST tensors: the
    functional-test transport for several ranks sharing one GPU (RCCL refuses two ranks on one device)."""
    from .engine import TransportCallbacks
    G = int(world_size)

    def all_reduce_min(a):
        t = torch.from_numpy(np.array(a, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return t.numpy()

    def all_gather(a):
        t = torch.from_numpy(np.array(a, dtype=np.uint8))
        outs = [torch.empty_like(t) for _ in range(G)]
        dist.all_gather(outs, t, group=group)
        return torch.cat(outs).numpy()

    def all_to_all_v(send, soff, sbytes, roff, rbytes, recv_total):
        ins = [torch.from_numpy(np.array(send[soff[p]:soff[p] + sbytes[p]], dtype=np.uint8)) for p in range(G)]
        outs = [torch.empty(rbytes[p], dtype=torch.uint8) for p in range(G)]
        # gloo has no all_to_all on every build: one broadcast-free exchange from point-to-point pairs
        me = dist.get_rank(group)
        reqs = []
        for p in range(G):
            if p == me:
                outs[p].copy_(ins[p])
                continue
            dst = dist.get_global_rank(group, p) if group is not None else p
            if sbytes[p]:
                reqs.append(dist.isend(ins[p], dst, group=group))
            if rbytes[p]:
                reqs.append(dist.irecv(outs[p], dst, group=group))
        for r in reqs:
            r.wait()
        out = np.zeros(recv_total, dtype=np.uint8)
        for p in range(G):
            out[roff[p]:roff[p] + rbytes[p]] = outs[p].numpy()
        return out

    return TransportCallbacks(G, all_reduce_min, all_gather, all_to_all_v, torch.cuda.synchronize)


class LibraryShardedSearch(object):
    """Row-sharded search through snk_sharded_knn_viterbi_batch.  `engine` holds this rank's shard
    (upload_target_only + set_shard), the full join matrix (upload_join_only) and, optionally, the
    replicated global sample (upload_global_sample) -- all before set_weights.
    transport 'rccl': the library's own RCCL communicator (one GPU per rank); 'gloo': host-staged
    collectives for functional tests."""

    def __init__(self, engine, rank=None, world_size=None, group=None, transport='rccl'):
        self.engine = engine
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world_size is None else world_size
        if transport == 'rccl':
            box = [engine.comm_unique_id() if self.rank == 0 else None]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            engine.comm_init(self.world, self.rank, box[0])
        else:
            engine.comm_init_transport(self.world, self.rank, gloo_transport(self.world, group))

    def knn_viterbi_batch(self, utterances, K):
        return self.engine.sharded_knn_viterbi_batch(utterances, K)

    def submit(self, utterances, K):
        """Queue a step (at most two in flight, the same sequence on every rank); see collect."""
        return self.engine.sharded_knn_viterbi_batch_submit(utterances, K)

    def collect(self, ticket):
        return self.engine.sharded_knn_viterbi_batch_collect(ticket)


def global_sample(F_unw, stride=16):
    """Every stride-th unit of the whole database: what every rank uploads with upload_global_sample."""
    return np.ascontiguousarray(F_unw[::int(stride)])
