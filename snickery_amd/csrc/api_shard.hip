// C ABI of libsnkhip.so, part 5: the database row-sharded over ranks -- device-pointer entry points, collectives inside the
// library (RCCL, loaded on demand, or caller-provided functions) and the sharded step.
#include "snk_engine.h"

#include <rccl/rccl.h>

// ---------------------------------------------------------------------------
// multi-GPU device-pointer entry points
// ---------------------------------------------------------------------------
int snk_knn_local_dev(snk_handle h, const double *Q, int64_t T, int D, int K, double *d2_dev_out,
                      int64_t *id_dev_out)
{
    CHK(check_ready(h, true, false));
    HIPCHK(hipSetDevice(h->device));
    if (!d2_dev_out || !id_dev_out) return fail("snk_knn_local_dev: null output");
    CHK(upload_queries(h, Q, T, D));
    CHK(knn_device(h, h->Qraw.as<double>(), T, K, nullptr, id_dev_out, nullptr, d2_dev_out));
    HIPCHK(hipStreamSynchronize(h->stream));
    collect_timers(h);
    return 0;
}

int snk_merge_topk_dev(snk_handle h, const double *d2_dev, const int64_t *id_dev, int G, int64_t T, int K,
                       int64_t *cand_out, double *dist_out)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!d2_dev || !id_dev || !cand_out || !dist_out) return fail("snk_merge_topk_dev: null argument");
    if (G < 1 || (int64_t)G * K > 8192) return fail("snk_merge_topk_dev: G*K=%d exceeds 8192", G * K);
    UttSlot &s = h->slot[0];
    CHK(s.cand.ensure((size_t)T * K * sizeof(int64_t)));
    CHK(s.tdist.ensure((size_t)T * K * sizeof(double)));
    launch_merge_topk(d2_dev, id_dev, G, T, K, s.cand.as<int64_t>(), s.tdist.as<double>(), h->stream);
    HIPCHK(hipGetLastError());
    {
        D2HPart parts[2] = {{cand_out, s.cand.p, (size_t)T * K * sizeof(int64_t)}, {dist_out, s.tdist.p, (size_t)T * K * sizeof(double)}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    return 0;
}

// Batch form of snk_knn_local_dev: the rows of all utterances against this rank's shard, first
// attempts enqueued back to back (status words checked once at the end, the rare overflowed
// utterance redone with the exact f64 sweep).  Results are complete when the call returns.
// Shared body of the three shard-local batch calls.  bound_out: stage A only (per-row bounds);
// bound_in: filter against the caller's bounds; Q == nullptr: the rows of the previous call are
// still resident (the bounds call and the bounded call of one step see the same batch).
static int upload_batch_queries(snk_engine *h, const double *Q, int64_t total, int D)
{
    CHK(h->Qall.ensure((size_t)total * D * sizeof(double)));
    StageTimer t(h, h->stream, TM_H2D);
    CHK(h2d(h, h->Qall.p, Q, (size_t)total * D * sizeof(double), h->stream));
    if (!h->tsel.empty()) launch_mask_columns(h->Qall.as<double>(), total, D, h->tmask.as<double>(), h->stream);
    h->qall_rows = total;
    return 0;
}

// defer != nullptr: nothing is synchronised; *defer receives the number of status words left in
// h->res_status (the caller checks them when it next touches the host, and redoes the step if any is set).
static int knn_local_batch(snk_engine *h, const char *who, const double *Q, const int64_t *row_offsets, int n_utts,
                           int D, int K, const double *bound_in, double *bound_out, double *d2_dev_out,
                           int64_t *id_dev_out, int *defer = nullptr, bool refine = false)
{
    CHK(check_ready(h, true, false));
    HIPCHK(hipSetDevice(h->device));
    if (!row_offsets || n_utts < 1 || (!bound_out && (!d2_dev_out || !id_dev_out)))
        return fail("%s: null/empty argument", who);
    if (D != h->Dt) return fail("query matrix has %d columns, database has %d", D, h->Dt);
    const int64_t total = row_offsets[n_utts];
    for (int u = 0; u < n_utts; ++u)
        if (row_offsets[u + 1] - row_offsets[u] < 1) return fail("%s: utterance %d has no rows", who, u);
    // the search is per row: fixed-size row groups, whatever the utterance boundaries.  Filtering against a shared
    // bound (a shard of a row-sharded database: many rows, few units) takes the largest calls there are: every launch
    // sweeps the shard once, and on 131 072 units a wavefront gets two work items per launch
    const int64_t step = bound_in ? SNK_KNN_MAX_ROWS : h->batch_rows > 0 ? h->batch_rows : total;
    const int n_groups = (int)((total + step - 1) / step);
    if (Q) {
        CHK(upload_batch_queries(h, Q, total, D));
    } else if (h->qall_rows != total) {
        return fail("%s: no query matrix given and the resident one has %lld rows, not %lld", who,
                    (long long)h->qall_rows, (long long)total);
    }
    CHK(h->res_status.ensure((size_t)n_groups * sizeof(int)));
    for (int g = 0; g < n_groups; ++g) {
        const int64_t r0 = g * step, rows = (r0 + step <= total) ? step : total - r0;
        CHK(knn_device(h, h->Qall.as<double>() + r0 * D, rows, K, nullptr, id_dev_out ? id_dev_out + r0 * K : nullptr,
                       nullptr, d2_dev_out ? d2_dev_out + r0 * K : nullptr, h->res_status.as<int>() + g,
                       bound_in ? bound_in + r0 : nullptr, bound_out ? bound_out + r0 : nullptr, false, refine));
    }
    if (defer) { *defer = n_groups; HIPCHK(hipGetLastError()); return 0; }
    if (bound_out) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipGetLastError());
        collect_timers(h);
        return 0;
    }
    std::vector<int> st((size_t)n_groups);
    CHK(d2h_sync(h, st.data(), h->res_status.p, (size_t)n_groups * sizeof(int), h->stream));
    HIPCHK(hipGetLastError());
    if ((h->ball_pass_ran || h->coarse_pass_ran || h->probe_ran) && h->cpairctl.p) {
        unsigned int ctl[4] = {0u, 0u, 0u, 0u};   // (of the last group's call: enough to judge the voice)
        CHK(d2h_sync(h, ctl, h->cpairctl.p, sizeof(ctl), h->stream));
        judge_filter(h, h->ball_pass_ran, h->ball_limit, h->coarse_pass_ran, h->coarse_limit, ctl[0], h->probe_ran, h->probe_limit, ctl[2]);
    }
    for (int g = 0; g < n_groups; ++g) {
        if (st[g] == 0) continue;
        if (st[g] & 2) h->tie_overflow = 1;
        const int64_t r0 = g * step, rows = (r0 + step <= total) ? step : total - r0;
        const int saved = h->precision;
        if (!((st[g] & ~3) == 0 && h->knn_level < 2)) h->precision = 0;      // (a lone list overflow: the voice's ladder, as in the batch pipeline)
        const int rc = knn_device(h, h->Qall.as<double>() + r0 * D, rows, K, nullptr, id_dev_out + r0 * K, nullptr,
                                  d2_dev_out + r0 * K);
        h->precision = saved;
        if (rc) return rc;
        h->batch_redos += 1;
    }
    collect_timers(h);
    return 0;
}

int snk_knn_local_batch_dev(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D, int K,
                            double *d2_dev_out, int64_t *id_dev_out)
{
    if (!Q) return fail("snk_knn_local_batch_dev: null query matrix");
    return knn_local_batch(h, "snk_knn_local_batch_dev", Q, row_offsets, n_utts, D, K, nullptr, nullptr, d2_dev_out, id_dev_out);
}

int snk_knn_local_batch_bounds_dev(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D, int K,
                                   double *bound_dev_out)
{
    if (!Q || !bound_dev_out) return fail("snk_knn_local_batch_bounds_dev: null argument");
    return knn_local_batch(h, "snk_knn_local_batch_bounds_dev", Q, row_offsets, n_utts, D, K, nullptr, bound_dev_out,
                           nullptr, nullptr);
}

int snk_knn_local_batch_bounded_dev(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D, int K,
                                    const double *bound_dev_in, double *d2_dev_out, int64_t *id_dev_out)
{
    if (!bound_dev_in) return fail("snk_knn_local_batch_bounded_dev: null bounds");
    return knn_local_batch(h, "snk_knn_local_batch_bounded_dev", Q, row_offsets, n_utts, D, K, bound_dev_in, nullptr,
                           d2_dev_out, id_dev_out);
}

// Second half of the sharded search on the rank that owns the utterances: merge the G shard-local
// lists of every row (the exchange step delivered them as (G, R, K)), then join costs on the main
// stream and the T-step recursions on the side streams, as in snk_knn_viterbi_batch.
int snk_merge_viterbi_batch_dev(snk_handle h, const double *d2_dev, const int64_t *id_dev, int G,
                                const int64_t *row_offsets, int n_utts, int K,
                                int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    CHK(check_ready(h, false, true));
    CHK(no_batch_in_flight(h, "snk_merge_viterbi_batch_dev"));
    HIPCHK(hipSetDevice(h->device));
    if (!d2_dev || !id_dev || !row_offsets || n_utts < 1 || !path_out || !path_len_out || !cost_out)
        return fail("snk_merge_viterbi_batch_dev: null/empty argument");
    if (K < 1 || K > 208) return fail("viterbi: n_candidates=%d outside 1..208", K);
    if (G < 1 || (int64_t)G * K > 8192) return fail("snk_merge_viterbi_batch_dev: G*K=%d exceeds 8192", G * K);
    const int64_t total = row_offsets[n_utts];
    for (int u = 0; u < n_utts; ++u)
        if (row_offsets[u + 1] - row_offsets[u] < 1) return fail("snk_merge_viterbi_batch_dev: utterance %d has no rows", u);
    CHK(h->mcand.ensure((size_t)total * K * sizeof(int64_t)));
    CHK(h->mdist.ensure((size_t)total * K * sizeof(double)));
    CHK(h->res_path.ensure((size_t)total * sizeof(int64_t)));
    CHK(h->res_plen.ensure((size_t)n_utts * sizeof(int64_t)));
    CHK(h->res_cost.ensure((size_t)n_utts * sizeof(double)));
    {
        StageTimer t(h, h->stream, TM_MERGE);
        launch_merge_topk(d2_dev, id_dev, G, total, K, h->mcand.as<int64_t>(), h->mdist.as<double>(), h->stream);
    }
    {
        const std::vector<int> first = group_utterances(h, row_offsets, n_utts, false, K);
        for (int g = 0; g + 1 < (int)first.size(); ++g)
            CHK(viterbi_group(h, g, row_offsets, first[g], first[g + 1], K, h->mcand.as<int64_t>(), h->mdist.as<double>(), true,
                              nullptr, nullptr, nullptr, n_utts));
    }
    for (int i = 0; i < 2; ++i) HIPCHK(hipStreamSynchronize(h->dp_stream[i]));
    HIPCHK(hipGetLastError());
    {
        StageTimer t(h, h->stream, TM_D2H);
        D2HPart parts[3] = {{path_out, h->res_path.p, (size_t)total * sizeof(int64_t)},
                            {path_len_out, h->res_plen.p, (size_t)n_utts * sizeof(int64_t)},
                            {cost_out, h->res_cost.p, (size_t)n_utts * sizeof(double)}};
        CHK(staged_d2h(h, h->stream, parts, 3));
    }
    collect_timers(h);
    return 0;
}

// ---------------------------------------------------------------------------
// collectives inside the library: RCCL (loaded when a communicator is first asked for: a single-GPU caller
// never maps its 500 MB) or caller-provided functions
// ---------------------------------------------------------------------------
namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load()
{
    if (g_rccl.lib) return 0;
    const char *names[] = {getenv("SNK_LIBRCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) {
        if (!n || !*n) continue;
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return fail("snk_comm: cannot load librccl (set SNK_LIBRCCL): %s", dlerror());
#define SNK_SYM(field, name)                                                                    \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name));                  \
    if (!g_rccl.field) return fail("snk_comm: librccl has no symbol %s", name)
    SNK_SYM(GetUniqueId, "ncclGetUniqueId"); SNK_SYM(CommInitRank, "ncclCommInitRank"); SNK_SYM(CommDestroy, "ncclCommDestroy"); SNK_SYM(CommAbort, "ncclCommAbort");
    SNK_SYM(AllReduce, "ncclAllReduce"); SNK_SYM(AllGather, "ncclAllGather"); SNK_SYM(Send, "ncclSend");
    SNK_SYM(Recv, "ncclRecv"); SNK_SYM(GroupStart, "ncclGroupStart"); SNK_SYM(GroupEnd, "ncclGroupEnd");
    SNK_SYM(GetErrorString, "ncclGetErrorString");
#undef SNK_SYM
    g_rccl.lib = lib;
    return 0;
}
#define NCCLCHK(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) return fail("%s failed: %s", #expr, g_rccl.GetErrorString(r_));     \
    } while (0)

void shard_plan(int64_t n, int G, int r, int64_t *lo, int64_t *hi)
{
    const int64_t base = n / G, rem = n % G;
    *lo = r * base + (r < rem ? r : rem);
    *hi = *lo + base + (r < rem ? 1 : 0);
}

// (comm_all_reduce_min: behind this namespace -- the K-NN pipeline of api_knn.hip calls it too)
int comm_all_gather(snk_engine *h, const void *send, void *recv, int64_t bytes)
{
    if (h->comm_ranks <= 1) { HIPCHK(hipMemcpyAsync(recv, send, (size_t)bytes, hipMemcpyDeviceToDevice, h->stream)); return 0; }
    if (h->have_transport) {
        if (h->transport.all_gather(h->transport.ctx, send, recv, bytes, h->stream)) return fail("transport all_gather failed");
        return 0;
    }
    NCCLCHK(g_rccl.AllGather(send, recv, (size_t)bytes, ncclChar, (ncclComm_t)h->nccl_comm, h->stream));
    return 0;
}
int comm_all_to_all_v(snk_engine *h, const void *send, const int64_t *soff, const int64_t *sbytes, void *recv,
                      const int64_t *roff, const int64_t *rbytes)
{
    const int G = h->comm_ranks;
    if (G <= 1) { HIPCHK(hipMemcpyAsync((char *)recv + roff[0], (const char *)send + soff[0], (size_t)sbytes[0], hipMemcpyDeviceToDevice, h->stream)); return 0; }
    if (h->have_transport) {
        if (h->transport.all_to_all_v(h->transport.ctx, send, soff, sbytes, recv, roff, rbytes, h->stream)) return fail("transport all_to_all_v failed");
        return 0;
    }
    // one fused group of point-to-point transfers: xGMI is point to point, every pair has its own link
    NCCLCHK(g_rccl.GroupStart());
    for (int p = 0; p < G; ++p) {
        if (sbytes[p]) NCCLCHK(g_rccl.Send((const char *)send + soff[p], (size_t)sbytes[p], ncclChar, p, (ncclComm_t)h->nccl_comm, h->stream));
        if (rbytes[p]) NCCLCHK(g_rccl.Recv((char *)recv + roff[p], (size_t)rbytes[p], ncclChar, p, (ncclComm_t)h->nccl_comm, h->stream));
    }
    NCCLCHK(g_rccl.GroupEnd());
    return 0;
}
}  // namespace

// the three collectives of the sharded search, on the engine's stream (this one is shared with api_knn.hip)
int comm_all_reduce_min(snk_engine *h, double *buf, int64_t n)
{
    if (h->comm_ranks <= 1) return 0;
    if (h->have_transport) {
        if (h->transport.all_reduce_min_f64(h->transport.ctx, buf, n, h->stream)) return fail("transport all_reduce_min_f64 failed");
        return 0;
    }
    NCCLCHK(g_rccl.AllReduce(buf, buf, (size_t)n, ncclDouble, ncclMin, (ncclComm_t)h->nccl_comm, h->stream));
    return 0;
}

int snk_shard_plan(int64_t n_items, int nranks, int rank, int64_t *lo_out, int64_t *hi_out)
{
    if (n_items < 0 || nranks < 1 || rank < 0 || rank >= nranks || !lo_out || !hi_out) return fail("snk_shard_plan: bad arguments");
    shard_plan(n_items, nranks, rank, lo_out, hi_out);
    return 0;
}

int snk_comm_unique_id(void *id_out, int capacity, int *bytes_out)
{
    if (!id_out || capacity < (int)sizeof(ncclUniqueId)) return fail("snk_comm_unique_id: need a buffer of %d bytes", (int)sizeof(ncclUniqueId));
    CHK(rccl_load());
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    if (bytes_out) *bytes_out = (int)sizeof(id);
    return 0;
}

int snk_comm_init(snk_handle h, int nranks, int rank, const void *unique_id)
{
    if (!h) return fail("null handle");
    if (nranks < 1 || rank < 0 || rank >= nranks || !unique_id) return fail("snk_comm_init: bad arguments");
    CHK(snk_comm_destroy(h));
    HIPCHK(hipSetDevice(h->device));
    CHK(rccl_load());
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    NCCLCHK(g_rccl.CommInitRank(&comm, nranks, id, rank));
    h->nccl_comm = comm;
    h->comm_ranks = nranks; h->comm_rank = rank; h->have_transport = false;
    return 0;
}

int snk_comm_init_transport(snk_handle h, int nranks, int rank, const snk_transport *t)
{
    if (!h) return fail("null handle");
    if (nranks < 1 || rank < 0 || rank >= nranks || !t || !t->all_reduce_min_f64 || !t->all_gather || !t->all_to_all_v)
        return fail("snk_comm_init_transport: bad arguments");
    CHK(snk_comm_destroy(h));
    h->transport = *t;
    h->have_transport = true;
    h->comm_ranks = nranks; h->comm_rank = rank;
    return 0;
}

int snk_comm_destroy(snk_handle h)
{
    if (!h) return 0;
    if (h->nccl_comm && g_rccl.CommDestroy && !h->comm_dead) {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        (void)g_rccl.CommDestroy((ncclComm_t)h->nccl_comm);
    }
    h->nccl_comm = nullptr;
    h->have_transport = false;
    h->comm_ranks = 0; h->comm_rank = 0;
    h->comm_dead = false;
    return 0;
}

// (no engine in these two signatures: a process-wide pinned bounce buffer, one transfer at a time)
static std::mutex g_bounce_lock;
static HostBuf g_bounce;

int snk_copy_to_host(void *dst_host, const void *src_dev, int64_t bytes)
{
    if (bytes < 0 || (bytes && (!dst_host || !src_dev))) return fail("snk_copy_to_host: bad arguments");
    std::lock_guard<std::mutex> lock(g_bounce_lock);
    const size_t chunk = (size_t)64 << 20;
    for (size_t off = 0; off < (size_t)bytes; off += chunk) {
        const size_t n = (size_t)bytes - off < chunk ? (size_t)bytes - off : chunk;
        CHK(g_bounce.ensure(n));
        HIPCHK(hipMemcpy(g_bounce.p, (const char *)src_dev + off, n, hipMemcpyDeviceToHost));
        memcpy((char *)dst_host + off, g_bounce.p, n);
    }
    return 0;
}

int snk_copy_to_device(void *dst_dev, const void *src_host, int64_t bytes)
{
    if (bytes < 0 || (bytes && (!dst_dev || !src_host))) return fail("snk_copy_to_device: bad arguments");
    std::lock_guard<std::mutex> lock(g_bounce_lock);
    const size_t chunk = (size_t)64 << 20;
    for (size_t off = 0; off < (size_t)bytes; off += chunk) {
        const size_t n = (size_t)bytes - off < chunk ? (size_t)bytes - off : chunk;
        CHK(g_bounce.ensure(n));
        memcpy(g_bounce.p, (const char *)src_host + off, n);
        HIPCHK(hipMemcpy((char *)dst_dev + off, g_bounce.p, n, hipMemcpyHostToDevice));
    }
    return 0;
}

// One sharded step (see include/snk.h), in two halves so that two steps can be in flight: submit queues everything
// up to the recursions of the owned utterances (K-NN of all rows, the two all-reduces, the exchange, merge, the four
// Viterbi passes on the side streams) and returns; collect waits for those recursions, gathers the results of all
// ranks and hands them out.  A caller that submits step i + 1 before collecting step i runs the Viterbi side of step i
// -- as long as a third of the step at G = 8 -- beside the K-NN of step i + 1, as the single-GPU batch pipeline does.
// safe: the exact float64 sweep with per-shard thresholds -- the path every rank takes again, together, when any
// rank's fast path reported a list overflow (rare; decided from the gathered status words, so all ranks agree).
// Query rows of a sharded step: every rank needs all of them, and every rank was handed all of them.  Each rank
// uploads only the rows of the utterances it owns and the ranks pass them on over xGMI (seven links in parallel
// against one PCIe upload of G times the bytes: 75 MB per rank and step at G = 8, B*).
static int upload_queries_gathered(snk_engine *h, const ShardTicket &t, const double *Q)
{
    const int G = t.G, me = t.me, D = t.D;
    CHK(h->Qall.ensure((size_t)t.R * D * sizeof(double)));
    {
        StageTimer tm(h, h->stream, TM_H2D);
        const int64_t a = t.row0[(size_t)me];
        if (t.r_own > 0)
            CHK(h2d(h, h->Qall.as<double>() + a * D, Q + a * D, (size_t)t.r_own * D * sizeof(double), h->stream));
    }
    std::vector<int64_t> soff((size_t)G), sb((size_t)G), roff((size_t)G), rb((size_t)G);
    for (int p = 0; p < G; ++p) {
        soff[(size_t)p] = t.row0[(size_t)me] * D * 8; sb[(size_t)p] = p == me ? 0 : t.r_own * D * 8;
        roff[(size_t)p] = t.row0[(size_t)p] * D * 8;  rb[(size_t)p] = p == me ? 0 : t.rows_to[(size_t)p] * D * 8;
    }
    CHK(comm_all_to_all_v(h, h->Qall.p, soff.data(), sb.data(), h->Qall.p, roff.data(), rb.data()));
    if (!h->tsel.empty()) launch_mask_columns(h->Qall.as<double>(), t.R, D, h->tmask.as<double>(), h->stream);
    h->qall_rows = t.R;
    return 0;
}

static int sharded_submit(snk_engine *h, ShardTicket &t, const double *Q, const int64_t *row_offsets, int n_utts, int D, int K,
                          bool safe)
{
    const int G = h->comm_ranks, me = h->comm_rank;
    const int64_t R = row_offsets[n_utts];
    t.G = G; t.me = me; t.n_utts = n_utts; t.K = K; t.D = D; t.R = R; t.safe = safe; t.Q = Q;
    t.offs.assign(row_offsets, row_offsets + n_utts + 1);
    t.ulo.assign((size_t)G, 0); t.uhi.assign((size_t)G, 0); t.rows_to.assign((size_t)G, 0); t.row0.assign((size_t)G, 0);
    for (int r = 0; r < G; ++r) {
        shard_plan(n_utts, G, r, &t.ulo[(size_t)r], &t.uhi[(size_t)r]);
        t.row0[(size_t)r] = row_offsets[t.ulo[(size_t)r]];
        t.rows_to[(size_t)r] = row_offsets[t.uhi[(size_t)r]] - t.row0[(size_t)r];
    }
    const int64_t r_own = t.rows_to[(size_t)me];
    const int n_own = (int)(t.uhi[(size_t)me] - t.ulo[(size_t)me]);
    t.r_own = r_own; t.n_own = n_own;
    int64_t slots = 0, Lmax = 0;
    for (int r = 0; r < G; ++r) slots = std::max(slots, t.uhi[(size_t)r] - t.ulo[(size_t)r]);
    for (int u = 0; u < n_utts; ++u) Lmax = std::max(Lmax, row_offsets[u + 1] - row_offsets[u]);
    t.slots = slots; t.rec = Lmax + 3;                             // per slot: path length, cost bits, K-NN status, path
    CHK(h->sh_d2.ensure((size_t)R * K * sizeof(double)));
    CHK(h->sh_id.ensure((size_t)R * K * sizeof(int64_t)));
    CHK(h->sh_bound.ensure((size_t)R * sizeof(double)));
    CHK(h->sh_rd2.ensure((size_t)G * (r_own > 0 ? r_own : 1) * K * sizeof(double)));
    CHK(h->sh_rid.ensure((size_t)G * (r_own > 0 ? r_own : 1) * K * sizeof(int64_t)));
    // what the recursions of THIS step read and write (the step submitted next has its own)
    CHK(t.mcand.ensure((size_t)(r_own > 0 ? r_own : 1) * K * sizeof(int64_t)));
    CHK(t.mdist.ensure((size_t)(r_own > 0 ? r_own : 1) * K * sizeof(double)));
    CHK(t.res_path.ensure((size_t)(r_own > 0 ? r_own : 1) * sizeof(int64_t)));
    CHK(t.res_plen.ensure((size_t)(n_own > 0 ? n_own : 1) * sizeof(int64_t)));
    CHK(t.res_cost.ensure((size_t)(n_own > 0 ? n_own : 1) * sizeof(double)));
    double *d2 = h->sh_d2.as<double>();
    int64_t *ids = h->sh_id.as<int64_t>();
    int n_status = 0;
    struct PrecisionGuard { snk_engine *e; int v; ~PrecisionGuard() { e->precision = v; } } guard{h, h->precision};
    if (safe) h->precision = 0;
    const double *Qk = Q;                       // what the K-NN calls are handed: nullptr = the rows are resident already
    if (G > 1 && h->shard_gather_queries) {
        CHK(upload_queries_gathered(h, t, Q));
        Qk = nullptr;
    }
    if (G > 1 && !safe) {
        // bounds of the K-th nearest key, one all-reduce (MIN) of R doubles
        double *bound = h->sh_bound.as<double>();
        if (h->gs_ready) {
            // own share of the rows against the replicated global sample; the others' entries stay +max
            if (Qk) CHK(upload_batch_queries(h, Q, R, D));
            launch_fill_threshold(bound, R, R, DBL_MAX, h->stream);
            const int64_t step = h->batch_rows > 0 ? h->batch_rows : r_own;
            for (int64_t r0 = 0; r0 < r_own; r0 += step) {
                const int64_t rows = (r0 + step <= r_own) ? step : r_own - r0;
                const int64_t a = t.row0[(size_t)me] + r0;
                CHK(knn_device(h, h->Qall.as<double>() + a * D, rows, K, nullptr, nullptr, nullptr, nullptr, nullptr,
                               nullptr, bound + a, true));
            }
        } else {
            CHK(knn_local_batch(h, "snk_sharded_knn_viterbi_batch", Qk, row_offsets, n_utts, D, K, nullptr, bound,
                                nullptr, nullptr, &n_status));
        }
        CHK(comm_all_reduce_min(h, bound, R));
        CHK(knn_local_batch(h, "snk_sharded_knn_viterbi_batch", nullptr, row_offsets, n_utts, D, K, bound, nullptr,
                            d2, ids, &n_status, true));
    } else {
        CHK(knn_local_batch(h, "snk_sharded_knn_viterbi_batch", Qk, row_offsets, n_utts, D, K, nullptr, nullptr,
                            d2, ids, safe ? nullptr : &n_status));
    }
    // this rank's K-NN status words: kept per step (the next step's K-NN reuses h->res_status)
    t.n_status = n_status;
    if (n_status > 0) {
        CHK(t.status.ensure((size_t)n_status * sizeof(int)));
        HIPCHK(hipMemcpyAsync(t.status.p, h->res_status.p, (size_t)n_status * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    }
    // exchange: the (R, K) matrices are ordered by destination (contiguous utterance blocks)
    std::vector<int64_t> soff((size_t)G), sb((size_t)G), roff((size_t)G), rb((size_t)G);
    for (int p = 0; p < G; ++p) {
        soff[(size_t)p] = t.row0[(size_t)p] * K * 8; sb[(size_t)p] = t.rows_to[(size_t)p] * K * 8;
        roff[(size_t)p] = (int64_t)p * r_own * K * 8; rb[(size_t)p] = r_own * K * 8;
    }
    const double *d2_all = d2;
    const int64_t *id_all = ids;
    if (G > 1 && h->shard_compact) {
        // compacted exchange (knn_kernels.hip shard_*): counts + valid entries per destination.  The block sizes must be on
        // the host for the transfers: an all-gather of every rank's G totals, ONE device -> host copy (the host waits for this
        // step's K-NN here; what was queued for the step before keeps running)
        auto pad16 = [](int64_t v) { return (v + 15) & ~(int64_t)15; };
        CHK(h->sh_cnt.ensure((size_t)R + 64));
        CHK(h->sh_off.ensure((size_t)R * sizeof(int) + 64));
        CHK(h->sh_tot.ensure((size_t)G * sizeof(int64_t)));
        CHK(h->sh_totall.ensure((size_t)G * G * sizeof(int64_t)));
        CHK(h->sh_plan.ensure((size_t)8 * G * sizeof(int64_t)));
        CHK(h->sh_pack.ensure((size_t)R * K * 16 + (size_t)R + (size_t)16 * G + 64));
        CHK(h->sh_rpack.ensure((size_t)G * ((size_t)(r_own > 0 ? r_own : 1) * K * 16 + (size_t)r_own + 32) + 64));
        CHK(h->sh_offq.ensure((size_t)G * (r_own > 0 ? r_own : 1) * sizeof(int) + 64));
        // plan arrays on the device: [0] row0, [1] rows_to, [2] send offsets, [3] totals sent, [4] receive offsets, [5] totals received,
        // [6] offsets of the received counts (= [4]), [7] q * r_own
        std::vector<int64_t> plan((size_t)8 * G, 0);
        for (int p = 0; p < G; ++p) { plan[(size_t)p] = t.row0[(size_t)p]; plan[(size_t)G + p] = t.rows_to[(size_t)p]; plan[(size_t)7 * G + p] = (int64_t)p * r_own; }
        int64_t *pl = h->sh_plan.as<int64_t>();
        CHK(h2d(h, pl, plan.data(), (size_t)2 * G * sizeof(int64_t), h->stream));
        launch_shard_count(ids, R, K, h->sh_cnt.as<unsigned char>(), h->stream);
        launch_shard_scan(h->sh_cnt.as<unsigned char>(), pl, pl + G, h->sh_off.as<int>(), pl, h->sh_tot.as<int64_t>(), G, h->stream);
        CHK(comm_all_gather(h, h->sh_tot.p, h->sh_totall.p, (int64_t)G * 8));
        std::vector<int64_t> totall((size_t)G * G);
        CHK(d2h_sync(h, totall.data(), h->sh_totall.p, (size_t)G * G * sizeof(int64_t), h->stream));
        if ((h->ball_pass_ran || h->coarse_pass_ran || h->probe_ran) && h->cpairctl.p) {
            // (the host is waiting here anyway: what the step's last K-NN call listed decides whether this voice keeps its filter)
            unsigned int ctl[4] = {0u, 0u, 0u, 0u};
            CHK(d2h_sync(h, ctl, h->cpairctl.p, sizeof(ctl), h->stream));
            judge_filter(h, h->ball_pass_ran, h->ball_limit, h->coarse_pass_ran, h->coarse_limit, ctl[0], h->probe_ran, h->probe_limit, ctl[2]);
        }
        // The verdict on the totals is COLLECTIVE: every rank holds the same G x G matrix (it was all-gathered) and the same plan, so
        // every rank checks every (sender, destination) pair and all of them refuse together, here, with the all-to-all not yet
        // queued anywhere (a rank that failed alone between the two collectives left its peers blocked in the second)
        for (int q = 0; q < G; ++q)
            for (int p = 0; p < G; ++p) {
                const int64_t v = totall[(size_t)q * G + p];
                if (v < 0 || v > t.rows_to[(size_t)p] * K) return fail("sharded exchange: inconsistent list totals between ranks (rank %d -> %d: %lld)", q, p, (long long)v);
            }
        int64_t so = 0, ro = 0;
        for (int p = 0; p < G; ++p) {
            const int64_t ts = totall[(size_t)me * G + p], tr = totall[(size_t)p * G + me];
            soff[(size_t)p] = so; sb[(size_t)p] = pad16(t.rows_to[(size_t)p]) + 16 * ts; so += sb[(size_t)p];
            roff[(size_t)p] = ro; rb[(size_t)p] = pad16(r_own) + 16 * tr; ro += rb[(size_t)p];
            plan[(size_t)2 * G + p] = soff[(size_t)p]; plan[(size_t)3 * G + p] = ts;
            plan[(size_t)4 * G + p] = roff[(size_t)p]; plan[(size_t)5 * G + p] = tr; plan[(size_t)6 * G + p] = roff[(size_t)p];
        }
        CHK(h2d(h, pl + 2 * G, plan.data() + (size_t)2 * G, (size_t)6 * G * sizeof(int64_t), h->stream));
        launch_shard_pack(d2, ids, h->sh_cnt.as<unsigned char>(), h->sh_off.as<int>(), pl, pl + G, pl + 2 * G, pl + 3 * G, G, R, K,
                          h->sh_pack.as<unsigned char>(), h->stream);
        CHK(comm_all_to_all_v(h, h->sh_pack.p, soff.data(), sb.data(), h->sh_rpack.p, roff.data(), rb.data()));
        if (r_own > 0) {
            std::vector<int64_t> rows_q((size_t)G, r_own);
            CHK(h2d(h, pl + G, rows_q.data(), (size_t)G * sizeof(int64_t), h->stream));     // (the send side's row counts are no longer needed)
            launch_shard_scan(h->sh_rpack.as<unsigned char>(), pl + 6 * G, pl + G, h->sh_offq.as<int>(), pl + 7 * G, h->sh_tot.as<int64_t>(), G, h->stream);
            launch_shard_unpack(h->sh_rpack.as<unsigned char>(), pl + 4 * G, pl + 5 * G, h->sh_offq.as<int>(), r_own, K, G,
                                h->sh_rd2.as<double>(), h->sh_rid.as<int64_t>(), h->stream);
        }
        HIPCHK(hipGetLastError());
        h->shard_last_sent_mb = (double)(so - sb[(size_t)me]) / 1e6;
        h->shard_last_padded_mb = (double)(R - t.rows_to[(size_t)me]) * K * 16 / 1e6;
        d2_all = h->sh_rd2.as<double>(); id_all = h->sh_rid.as<int64_t>();
    } else if (G > 1) {
        CHK(comm_all_to_all_v(h, d2, soff.data(), sb.data(), h->sh_rd2.p, roff.data(), rb.data()));
        CHK(comm_all_to_all_v(h, ids, soff.data(), sb.data(), h->sh_rid.p, roff.data(), rb.data()));
        h->shard_last_sent_mb = h->shard_last_padded_mb = (double)(R - t.rows_to[(size_t)me]) * K * 16 / 1e6;
        d2_all = h->sh_rd2.as<double>(); id_all = h->sh_rid.as<int64_t>();
    }
    // owner: merge, join bounds / costs, Viterbi of the owned utterances -- queued, not waited for
    t.own_off.assign((size_t)n_own + 1, 0);
    for (int u = 0; u <= n_own; ++u) t.own_off[(size_t)u] = row_offsets[t.ulo[(size_t)me] + u] - t.row0[(size_t)me];
    if (n_own > 0) {
        {
            StageTimer tm(h, h->stream, TM_MERGE);
            launch_merge_topk(d2_all, id_all, G, r_own, K, t.mcand.as<int64_t>(), t.mdist.as<double>(), h->stream);
        }
        const std::vector<int> first = group_utterances(h, t.own_off.data(), n_own, false, K);
        for (int g = 0; g + 1 < (int)first.size(); ++g)
            CHK(viterbi_group(h, g, t.own_off.data(), first[g], first[g + 1], K, t.mcand.as<int64_t>(), t.mdist.as<double>(), true,
                              t.res_path.as<int64_t>(), t.res_plen.as<int64_t>(), t.res_cost.as<double>(), n_own));
    }
    for (int i = 0; i < 2; ++i) HIPCHK(hipEventRecord(t.side_done[i], h->dp_stream[i]));
    HIPCHK(hipEventRecord(t.main_done, h->stream));
    // own results and status words -> pinned memory, on the copy stream behind this step's recursions (no host wait, no
    // default-stream copy: a step submitted next keeps running)
    {
        const size_t sz_path = ((size_t)(r_own > 0 ? r_own : 1) * 8 + 63) & ~(size_t)63, sz_u = ((size_t)(n_own > 0 ? n_own : 1) * 8 + 63) & ~(size_t)63;
        const size_t sz_st = ((size_t)(n_status > 0 ? n_status : 1) * sizeof(int) + 63) & ~(size_t)63;
        CHK(t.stage.ensure(sz_path + 2 * sz_u + sz_st));
        HIPCHK(hipStreamWaitEvent(h->copy_stream, t.main_done, 0));
        for (int i = 0; i < 2; ++i) HIPCHK(hipStreamWaitEvent(h->copy_stream, t.side_done[i], 0));
        char *st = (char *)t.stage.p;
        if (h->results_by_kernel) {
            // a kernel's stores, not DMA copies that would sit at the head of the engine's queue until this step's recursions are
            // done, with the next step's upload behind them (viterbi_kernels.hip results_to_host_kernel)
            void *dst[5] = {st, st + sz_path, st + sz_path + sz_u, st + sz_path + 2 * sz_u, nullptr};
            const void *src[5] = {t.res_path.p, t.res_plen.p, t.res_cost.p, t.status.p, nullptr};
            const size_t nb[5] = {n_own > 0 ? (size_t)r_own * sizeof(int64_t) : 0, n_own > 0 ? (size_t)n_own * sizeof(int64_t) : 0,
                                  n_own > 0 ? (size_t)n_own * sizeof(double) : 0, n_status > 0 ? (size_t)n_status * sizeof(int) : 0, 0};
            launch_results_to_host(dst, src, nb, 4, h->copy_stream);
        } else {
        if (n_own > 0) {
            HIPCHK(hipMemcpyAsync(st, t.res_path.p, (size_t)r_own * sizeof(int64_t), hipMemcpyDeviceToHost, h->copy_stream));
            HIPCHK(hipMemcpyAsync(st + sz_path, t.res_plen.p, (size_t)n_own * sizeof(int64_t), hipMemcpyDeviceToHost, h->copy_stream));
            HIPCHK(hipMemcpyAsync(st + sz_path + sz_u, t.res_cost.p, (size_t)n_own * sizeof(double), hipMemcpyDeviceToHost, h->copy_stream));
        }
        if (n_status > 0)
            HIPCHK(hipMemcpyAsync(st + sz_path + 2 * sz_u, t.status.p, (size_t)n_status * sizeof(int), hipMemcpyDeviceToHost, h->copy_stream));
        }
        HIPCHK(hipEventRecord(t.done, h->copy_stream));
    }
    HIPCHK(hipGetLastError());
    t.busy = true;
    return 0;
}

static int sharded_collect(snk_engine *h, ShardTicket &t, int64_t *path_out, int64_t *path_len_out, double *cost_out, bool *any_bad_out)
{
    const int G = t.G, me = t.me, n_own = t.n_own;
    const int64_t r_own = t.r_own, slots = t.slots, rec = t.rec;
    // the recursions of this step (side streams) and everything of it on the main stream; a step submitted after it
    // may still be running
    HIPCHK(hipEventSynchronize(t.done));
    HIPCHK(hipGetLastError());
    t.busy = false;
    const size_t sz_path = ((size_t)(r_own > 0 ? r_own : 1) * 8 + 63) & ~(size_t)63, sz_u = ((size_t)(n_own > 0 ? n_own : 1) * 8 + 63) & ~(size_t)63;
    const char *stg = (const char *)t.stage.p;
    const int64_t *own_path = reinterpret_cast<const int64_t *>(stg);
    const int64_t *own_len = reinterpret_cast<const int64_t *>(stg + sz_path);
    const double *own_cost = reinterpret_cast<const double *>(stg + sz_path + sz_u);
    int status = 0;
    {
        const int *st = reinterpret_cast<const int *>(stg + sz_path + 2 * sz_u);
        for (int i = 0; i < t.n_status; ++i) status |= st[i];
        if (status != 0 && (status & ~3) == 0 && h->knn_level < 2) { h->knn_level += 1; h->knn_escalations += 1; }     // this rank's later steps: longer lists (snk_engine.h)
    }
    // results of every utterance to every rank: fixed-size records, one all-gather (queued on the main stream: behind
    // the K-NN and the exchange of a step submitted in the meantime)
    CHK(h->sh_res.ensure((size_t)slots * rec * sizeof(int64_t)));
    CHK(h->sh_resall.ensure((size_t)G * slots * rec * sizeof(int64_t)));
    std::vector<int64_t> mine((size_t)(slots * rec), 0);
    for (int j = 0; j < n_own; ++j) {
        int64_t *r = mine.data() + (size_t)j * rec;
        r[0] = own_len[j];
        memcpy(&r[1], &own_cost[j], sizeof(double));
        r[2] = status;
        memcpy(&r[3], own_path + t.own_off[(size_t)j], (size_t)own_len[j] * sizeof(int64_t));
    }
    if (n_own == 0 && slots > 0) mine[2] = status;
    CHK(h2d(h, h->sh_res.p, mine.data(), mine.size() * sizeof(int64_t), h->stream));
    CHK(comm_all_gather(h, h->sh_res.p, h->sh_resall.p, (int64_t)(mine.size() * sizeof(int64_t))));
    std::vector<int64_t> all((size_t)G * mine.size());
    CHK(d2h_sync(h, all.data(), h->sh_resall.p, all.size() * sizeof(int64_t), h->stream));
    bool any_bad = false;
    for (int r = 0; r < G; ++r) {
        const int64_t *blk = all.data() + (size_t)r * mine.size();
        if (slots > 0 && blk[2] != 0) any_bad = true;
        for (int64_t j = 0; j < t.uhi[(size_t)r] - t.ulo[(size_t)r]; ++j) {
            const int64_t *q = blk + (size_t)j * rec;
            const int64_t u = t.ulo[(size_t)r] + j;
            if (q[2] != 0) any_bad = true;
            path_len_out[u] = q[0];
            memcpy(&cost_out[u], &q[1], sizeof(double));
            memcpy(path_out + t.offs[(size_t)u], &q[3], (size_t)q[0] * sizeof(int64_t));
        }
    }
    if (any_bad_out) *any_bad_out = any_bad && !t.safe;
    (void)me;
    collect_timers(h);
    return 0;
}

// Everything that does not depend on the rank is checked here, BEFORE the first collective of the step is queued: all
// ranks then fail together, with nothing in flight.
static int sharded_check(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D, int K)
{
    CHK(check_ready(h, true, true));
    HIPCHK(hipSetDevice(h->device));
    if (h->comm_ranks < 1) return fail("snk_sharded_knn_viterbi_batch: no communicator (snk_comm_init)");
    if (h->comm_dead) return fail("snk_sharded_knn_viterbi_batch: the communicator was aborted after a local error (snk_comm_init again, on every rank)");
    if (!Q || !row_offsets || n_utts < 1) return fail("snk_sharded_knn_viterbi_batch: null/empty argument");
    if (D != h->Dt) return fail("query matrix has %d columns, database has %d", D, h->Dt);
    if (K < 1 || K > 208) return fail("snk_sharded_knn_viterbi_batch: n_candidates=%d outside the supported range 1..208", K);
    if ((int64_t)h->comm_ranks * K > 8192) return fail("snk_sharded_knn_viterbi_batch: G*K=%d exceeds 8192", h->comm_ranks * K);
    if (row_offsets[0] != 0) return fail("snk_sharded_knn_viterbi_batch: row_offsets[0] must be 0");
    for (int u = 0; u < n_utts; ++u)
        if (row_offsets[u + 1] - row_offsets[u] < 1) return fail("snk_sharded_knn_viterbi_batch: utterance %d has no rows", u);
    return 0;
}

// A local failure (allocation, launch, transport) after the step's first collective went out: the peers are, or will be,
// blocked in a collective this rank never joins.  Abort the communicator -- their pending operations then end with an
// error instead of hanging -- and refuse further sharded steps until a new communicator is set up.
static int sharded_fail(snk_engine *h, int rc)
{
    if (rc == 0) return 0;
    const std::string msg = last_error_string();
    if (h->comm_ranks > 1 && !h->have_transport && h->nccl_comm && g_rccl.CommAbort && !h->comm_dead) {
        (void)g_rccl.CommAbort((ncclComm_t)h->nccl_comm);
        h->comm_dead = true;
        for (auto &t : h->sticket) t.busy = false;
        (void)fail("%s [communicator aborted: the other ranks see an error instead of waiting]", msg.c_str());
    }
    return rc;
}

int snk_sharded_knn_viterbi_batch_submit(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D, int K,
                                         int *ticket_out)
{
    CHK(sharded_check(h, Q, row_offsets, n_utts, D, K));
    if (any_batch_busy(h))
        return fail("snk_sharded_knn_viterbi_batch_submit: a submitted batch is still in flight (snk_knn_viterbi_batch_collect it first)");
    if (!ticket_out) return fail("snk_sharded_knn_viterbi_batch_submit: null ticket");
    const int slot = h->sticket[h->snext].busy ? (h->snext ^ 1) : h->snext;
    ShardTicket &t = h->sticket[slot];
    if (t.busy) return fail("snk_sharded_knn_viterbi_batch_submit: two steps are in flight already (collect one first)");
    if (!t.main_done) {
        HIPCHK(hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&t.main_done, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) HIPCHK(hipEventCreateWithFlags(&t.side_done[i], hipEventDisableTiming));
    }
    CHK(sharded_fail(h, sharded_submit(h, t, Q, row_offsets, n_utts, D, K, false)));
    h->snext = slot ^ 1;
    *ticket_out = slot;
    return 0;
}

int snk_sharded_knn_viterbi_batch_collect(snk_handle h, int ticket, int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (ticket < 0 || ticket > 1 || !h->sticket[ticket].busy)
        return fail("snk_sharded_knn_viterbi_batch_collect: no step behind ticket %d", ticket);
    if (!path_out || !path_len_out || !cost_out) return fail("snk_sharded_knn_viterbi_batch_collect: null output");
    ShardTicket &t = h->sticket[ticket];
    bool redo = false;
    CHK(sharded_fail(h, sharded_collect(h, t, path_out, path_len_out, cost_out, &redo)));
    if (redo) {
        // every rank saw the same status words: all of them redo this step in the exact mode, now -- behind whatever
        // a step submitted in the meantime has queued (its recursions must be through with the shared workspaces first)
        h->batch_redos += 1;
        // a step submitted in the meantime is complete on the device after this (its results wait in its own buffers and
        // in pinned memory until it is collected); the shared workspaces are free for the redo
        for (int i = 0; i < 2; ++i) HIPCHK(hipStreamSynchronize(h->dp_stream[i]));
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipStreamSynchronize(h->copy_stream));
        const std::vector<int64_t> offs = t.offs;
        CHK(sharded_fail(h, sharded_submit(h, t, t.Q, offs.data(), t.n_utts, t.D, t.K, true)));
        CHK(sharded_fail(h, sharded_collect(h, t, path_out, path_len_out, cost_out, nullptr)));
    }
    return 0;
}

int snk_sharded_knn_viterbi_batch(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D, int K,
                                  int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    if (!path_out || !path_len_out || !cost_out) return fail("snk_sharded_knn_viterbi_batch: null/empty argument");
    int ticket = -1;
    CHK(snk_sharded_knn_viterbi_batch_submit(h, Q, row_offsets, n_utts, D, K, &ticket));
    return snk_sharded_knn_viterbi_batch_collect(h, ticket, path_out, path_len_out, cost_out);
}

// greedy search with the scan of every step split over the ranks (include/snk.h).  Host-driven: a step is the rank's scan
// launch, one all-gather of 16 bytes per rank, the pick launch -- queued on the engine's stream without a host wait between
// them over RCCL (a transport's callbacks synchronise themselves).
// Collective verdict before the first step: every rank brings (ok, T, D, start_state) -- ok = its own preconditions
// and allocations held -- and all ranks gather all records; unless every rank is ok and the arguments agree, EVERY rank returns
// an error here, with nothing in flight.  (A rank without a communicator cannot tell anybody: that stays a local error.)
static int sharded_greedy_preflight(snk_engine *h, bool ok, int64_t T, int D, int64_t start_state, const std::string &why)
{
    const int G = h->comm_ranks, me = h->comm_rank;
    CHK(h->gshard.ensure((size_t)32 * (G + 1) + (size_t)16 * (G + 1)));
    int64_t rec[4] = {ok ? 1 : 0, T, (int64_t)D, start_state};
    char *base = reinterpret_cast<char *>(h->gshard.p) + (size_t)16 * (G + 1);
    CHK(h2d(h, base, rec, sizeof(rec), h->stream));
    CHK(comm_all_gather(h, base, base + 32, 32));
    std::vector<int64_t> all((size_t)4 * G);
    CHK(d2h_sync(h, all.data(), base + 32, (size_t)32 * G, h->stream));
    for (int r = 0; r < G; ++r)
        if (all[(size_t)4 * r] != 1)
            return r == me ? fail("%s [refused on every rank]", why.c_str())
                           : fail("snk_sharded_greedy: rank %d refused the call (its preconditions or allocations failed): refused on every rank", r);
    for (int r = 0; r < G; ++r)
        if (all[(size_t)4 * r + 1] != T || all[(size_t)4 * r + 2] != D || all[(size_t)4 * r + 3] != start_state)
            return fail("snk_sharded_greedy: the ranks disagree about the call (rank %d: T=%lld D=%lld start_state=%lld, rank %d: T=%lld D=%d start_state=%lld)",
                        r, (long long)all[(size_t)4 * r + 1], (long long)all[(size_t)4 * r + 2], (long long)all[(size_t)4 * r + 3], me, (long long)T, D, (long long)start_state);
    return 0;
}

static int sharded_greedy_steps(snk_engine *h, int64_t nsteps, int64_t start_state, int64_t *path_out, double *dist_out);

int snk_sharded_greedy(snk_handle h, const double *Q, int64_t T, int D, int64_t start_state,
                       int64_t *path_out, double *dist_out, int64_t *nsteps_out)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->comm_ranks < 1) return fail("snk_sharded_greedy: no communicator (snk_comm_init)");
    if (h->comm_dead) return fail("snk_sharded_greedy: the communicator was aborted after a local error (snk_comm_init again, on every rank)");
    // ---- this rank's own preconditions: noted, not returned -- the verdict is collective ----
    std::string why;
    bool ok = true;
    auto refuse = [&](int rc) { if (rc && ok) { ok = false; why = last_error_string(); } return rc; };
    refuse(check_ready(h, true, true));
    if (ok) refuse(no_batch_in_flight(h, "snk_sharded_greedy"));
    if (ok && !h->have_glay) refuse(fail("snk_sharded_greedy: greedy layout not set (snk_set_greedy_layout)"));
    if (ok && (h->shard_offset != 0 || (h->global_N > 0 && h->global_N != h->N)))
        refuse(fail("snk_sharded_greedy: every rank holds the whole database (this engine holds a shard of it)"));
    if (ok && (!path_out || !nsteps_out)) refuse(fail("snk_sharded_greedy: null output"));
    if (ok && start_state >= h->glay.Nwin) refuse(fail("snk_sharded_greedy: start_state %lld out of range", (long long)start_state));
    if (ok && (!Q || D != h->Dt || T < 0)) refuse(fail("snk_sharded_greedy: query matrix has %d columns (database: %d), %lld rows", D, h->Dt, (long long)T));
    int64_t nsteps = 0;
    if (ok) {
        const GreedyLayout &g = h->glay;
        nsteps = T / g.me;
        // everything that can fail locally before the first step: the upload, the tiles, the step buffers
        refuse(upload_queries(h, Q, T, D));
        if (ok && nsteps > 0 && !h->gtiles_ready) {
            if (!refuse(h->gtiles.ensure(greedy_tile_bytes(g, h->Dt)))) {
                launch_greedy_tiles(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->JC_unw.as<float>(), h->Jp, h->gtiles.as<float>(), h->stream);
                if (hipGetLastError() != hipSuccess) refuse(fail("snk_sharded_greedy: greedy tiles launch failed"));
                else h->gtiles_ready = true;
            }
        }
        if (ok && nsteps > 0) {
            const int64_t ntiles = (g.Nwin + 63) / 64;
            int64_t tlo = 0, thi = 0;
            shard_plan(ntiles, h->comm_ranks, h->comm_rank, &tlo, &thi);
            const int nblk = greedy_shard_blocks(g, h->Dt, h->n_cus, thi - tlo > 0 ? thi - tlo : 1);
            if (refuse(h->gprev.ensure(2 * greedy_table_doubles(g, h->Dt) * sizeof(double) + 512)) ||
                refuse(h->gsync.ensure(greedy_counter_bytes())) || refuse(h->gblkmin.ensure((size_t)nblk * sizeof(double))) ||
                refuse(h->gblkarg.ensure((size_t)nblk * sizeof(int64_t))) || refuse(h->gpath.ensure((size_t)nsteps * sizeof(int64_t))) ||
                refuse(h->gdist.ensure((size_t)nsteps * sizeof(double)))) { /* noted */ }
        }
    }
    // (a failure INSIDE the verdict's own collective, or in a step's all-gather, leaves peers in a collective: abort the communicator)
    CHK(sharded_fail(h, sharded_greedy_preflight(h, ok, T, D, start_state, why)));
    *nsteps_out = nsteps;
    if (nsteps == 0) { HIPCHK(hipStreamSynchronize(h->stream)); collect_timers(h); return 0; }
    return sharded_fail(h, sharded_greedy_steps(h, nsteps, start_state, path_out, dist_out));
}

static int sharded_greedy_steps(snk_engine *h, int64_t nsteps, int64_t start_state, int64_t *path_out, double *dist_out)
{
    const GreedyLayout &g = h->glay;
    const int G = h->comm_ranks, me = h->comm_rank;
    const int64_t ntiles = (g.Nwin + 63) / 64;
    int64_t tlo = 0, thi = 0;
    shard_plan(ntiles, G, me, &tlo, &thi);
    const int64_t tile_n = thi - tlo;
    const int nblk = greedy_shard_blocks(g, h->Dt, h->n_cus, tile_n > 0 ? tile_n : 1);
    CHK(h->gprev.ensure(2 * greedy_table_doubles(g, h->Dt) * sizeof(double) + 512));
    CHK(h->gsync.ensure(greedy_counter_bytes()));
    CHK(h->gblkmin.ensure((size_t)nblk * sizeof(double)));
    CHK(h->gblkarg.ensure((size_t)nblk * sizeof(int64_t)));
    CHK(h->gpath.ensure((size_t)nsteps * sizeof(int64_t)));
    CHK(h->gdist.ensure((size_t)nsteps * sizeof(double)));
    double *mine = h->gshard.as<double>(), *all = mine + 2;           // (allocated by the pre-flight: 16 (G + 1) bytes for the steps + its own records)
    {
        // a rank without tiles (more ranks than tiles) contributes "nothing found"
        const double none_d = DBL_MAX;
        const int64_t none_i = INT64_MAX;
        unsigned char init[16];
        memcpy(init, &none_d, 8); memcpy(init + 8, &none_i, 8);
        CHK(h2d(h, mine, init, 16, h->stream));
    }
    {
        StageTimer t(h, h->stream, TM_GREEDY_STEPS);
        launch_greedy_shard_init(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                                 h->wj.as<double>(), h->gtiles.as<float>(), h->Qraw.as<double>(), nsteps, start_state,
                                 h->gprev.as<double>(), h->gsync.as<unsigned int>(), h->stream);
        for (int64_t st = 0; st < nsteps; ++st) {
            if (tile_n > 0)
                launch_greedy_shard_step(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                                         h->wj.as<double>(), h->gtiles.as<float>(), h->Qraw.as<double>(), st, nsteps, tlo, tile_n,
                                         h->gprev.as<double>(), h->gblkmin.as<double>(), h->gblkarg.as<int64_t>(), nblk, h->n_cus,
                                         h->gsync.as<unsigned int>(), mine, h->stream);
            CHK(comm_all_gather(h, mine, all, 16));
            launch_greedy_shard_pick(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                                     h->wj.as<double>(), h->Qraw.as<double>(), st, nsteps, all, G, h->gprev.as<double>(),
                                     h->gpath.as<int64_t>(), h->gdist.as<double>(), h->stream);
        }
    }
    HIPCHK(hipGetLastError());
    {
        StageTimer t(h, h->stream, TM_D2H);
        D2HPart parts[2] = {{path_out, h->gpath.p, (size_t)nsteps * sizeof(int64_t)},
                            {dist_out, h->gdist.p, dist_out ? (size_t)nsteps * sizeof(double) : 0}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    collect_timers(h);
    return 0;
}
