// Internal header of the host side of libsnkhip.so: the engine's state (one device, its streams, every workspace) and the
// helpers the api_*.hip translation units share.  Device memory, streams and events are plain HIP; there is no CPU compute
// fallback.  The C ABI itself is include/snk.h; the kernels' launchers are declared in snk_internal.h.
#pragma once
#include "snk_internal.h"
#include "../../include/snk.h"

#include <dlfcn.h>
#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>

using namespace snk;

int fail(const char *fmt, ...);                 // sets the thread's error message (snk_last_error), returns 1
const std::string &last_error_string();

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define CHK(expr)                    \
    do {                             \
        int rc_ = (expr);            \
        if (rc_) return rc_;         \
    } while (0)

// ---------------------------------------------------------------------------
enum TimerId {
    TM_H2D = 0, TM_PREP, TM_KNN_MINIMA, TM_KNN_THRESHOLD, TM_KNN_FILTER, TM_KNN_BUCKET, TM_KNN_FINALIZE,
    TM_JOIN, TM_VITERBI_DP, TM_D2H, TM_GREEDY_TARGET, TM_GREEDY_STEPS, TM_WEIGHTS, TM_MERGE, TM_JOIN_LB, TM_DP_LB,
    TM_JOIN_SPARSE, TM_DP_SPARSE, TM_KNN_BALLMIN, TM_COUNT
};

extern const char *kTimerNames[TM_COUNT];

// Debug allocator (environment SNK_GUARD=1|2|3, read once): every device buffer gets its own virtual range with an
// unmapped page after it (1: the buffer ends where the mapping ends, an over-read or over-write of even one 16-byte
// element faults at once; 2: it starts where the mapping starts) and exactly the bytes asked for -- no growth slack,
// no reuse, fresh memory filled with 0xFF (a NaN / huge-index pattern).  Every allocation is logged with its call site,
// so the page address in the runtime's "Memory access fault" line names the buffer that was overrun.  Speed is of no
// concern in this mode; the product path never sets it.

inline int guard_mode()
{
    static int mode = -1;
    if (mode < 0) { const char *e = getenv("SNK_GUARD"); mode = e ? atoi(e) : 0; if (mode < 0 || mode > 3) mode = 0; }
    return mode;
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    // guard mode bookkeeping
    void *va = nullptr; size_t va_bytes = 0, map_bytes = 0; hipMemGenericAllocationHandle_t mh{}; bool guarded = false;
    int ensure(size_t need, const char *file = __builtin_FILE(), int line = __builtin_LINE())
    {
        if (guard_mode()) return ensure_guarded(need, file, line);
        if (need <= bytes) return 0;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        size_t want = need + need / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
        bytes = want;
        return 0;
    }
    int ensure_guarded(size_t need, const char *file, int line)
    {
        if (need == 0) need = 1;
        if (p && need == bytes) return 0;                       // same request: keep (contents may be live)
        release();
        if (guard_mode() == 3) {
            // plain allocations of exactly the bytes asked for, filled with 0xFF: separates "relies on fresh memory being
            // zero / on the growth slack" from what the unmapped neighbours of modes 1 and 2 catch
            hipError_t e3 = hipMalloc(&p, need);
            if (e3 != hipSuccess) { p = nullptr; return fail("hipMalloc(%zu) failed: %s", need, hipGetErrorString(e3)); }
            (void)hipMemset(p, 0xFF, need);
            (void)hipDeviceSynchronize();
            bytes = need;
            return 0;
        }
        int dev = 0;
        (void)hipGetDevice(&dev);
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
        if (e != hipSuccess || gran == 0) return fail("SNK_GUARD: hipMemGetAllocationGranularity: %s", hipGetErrorString(e));
        const size_t mapped = ((need + gran - 1) / gran) * gran;
        e = hipMemAddressReserve(&va, mapped + 2 * gran, gran, nullptr, 0);
        if (e != hipSuccess) { va = nullptr; return fail("SNK_GUARD: hipMemAddressReserve(%zu): %s", mapped + 2 * gran, hipGetErrorString(e)); }
        va_bytes = mapped + 2 * gran;
        e = hipMemCreate(&mh, mapped, &prop, 0);
        if (e != hipSuccess) { (void)hipMemAddressFree(va, va_bytes); va = nullptr; return fail("SNK_GUARD: hipMemCreate(%zu): %s", mapped, hipGetErrorString(e)); }
        char *base = (char *)va + gran;
        e = hipMemMap(base, mapped, 0, mh, 0);
        if (e != hipSuccess) return fail("SNK_GUARD: hipMemMap: %s", hipGetErrorString(e));
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(base, mapped, &acc, 1);
        if (e != hipSuccess) return fail("SNK_GUARD: hipMemSetAccess: %s", hipGetErrorString(e));
        map_bytes = mapped;
        guarded = true;
        (void)hipMemset(base, 0xFF, mapped);
        (void)hipDeviceSynchronize();
        // mode 1: right-aligned, to hipMalloc's own 256-byte alignment (kernels may rely on it; SNK_GUARD_ALIGN
        // overrides); mode 2: left-aligned
        static size_t al = 0;
        if (!al) { const char *e2 = getenv("SNK_GUARD_ALIGN"); al = e2 ? (size_t)atoi(e2) : 256; if (al < 16 || (al & (al - 1))) al = 256; }
        const size_t need16 = (need + al - 1) & ~(al - 1);
        p = guard_mode() == 1 ? base + (mapped - need16) : base;
        bytes = need;
        fprintf(stderr, "[snk-guard] %p..%p (%zu B, mapping %p..%p) %s:%d\n", p, (char *)p + need, need, (void *)base,
                (void *)(base + mapped), file, line);
        return 0;
    }
    void release()
    {
        if (guarded) {
            (void)hipDeviceSynchronize();
            (void)hipMemUnmap((char *)va + (va_bytes - map_bytes) / 2, map_bytes);
            (void)hipMemRelease(mh);
            (void)hipMemAddressFree(va, va_bytes);
            va = nullptr; guarded = false; p = nullptr; bytes = 0;
            return;
        }
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct HostBuf {        // pinned host staging (pageable D2H of > ~64 KB pins the user buffer: ms)
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return 0;
        if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
        size_t want = need + need / 4 + 4096;
        if (want < ((size_t)8 << 20)) want = (size_t)8 << 20;     // pinned allocations cost milliseconds
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return fail("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
        bytes = want;
        return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
};

struct EvPair { hipEvent_t a, b; int id; };

struct UttSlot {      // per in-flight utterance workspace (batch pipeline uses two)
    DevBuf cand, tdist, J, bp, path, plen, cost;
    DevBuf Jlo, scale, sets, cex;                   // sparse Viterbi path (joinfast_kernels.hip)
    hipEvent_t knn_done = nullptr, vit_done = nullptr;
    bool vit_recorded = false;          // a recursion was queued on this workspace (its event is valid)
};

#define SNK_BATCH_SLOTS 3
struct BatchSlot {    // one submitted batch (snk_knn_viterbi_batch_submit / _collect): SNK_BATCH_SLOTS may be in flight
    DevBuf Qall, cand, dist, path, plen, cost, status;
    HostBuf stage, qstage;                // results / query rows of this batch (pinned: the copies are queued, not waited for)
    hipEvent_t done = nullptr;            // results of this batch are in `stage`
    bool busy = false;
    int n_utts = 0, n_groups = 0, K = 0, D = 0;
    int64_t q_rows = -1; int q_D = 0;     // query rows resident in Qall (a later submit with Q == NULL searches them again)
    std::vector<int64_t> q_offs;
    double ball_limit = -1.0;             // >= 0: the ball pass listed this batch's tile pairs; beyond this many the voice goes to the coarse sweep
    double coarse_limit = -1.0;           // >= 0: the coarse sweep listed them; beyond this many the voice goes to the one-pass sweep
    std::vector<int> probe_kind;          // per group: the counting probe its K-NN call carried (0 none, 1 ball pass, 2 coarse sweep)
    std::vector<double> probe_limit;
    int64_t seq = -1;                     // number of this batch (timing events of the Viterbi latch)
    int64_t operand_gen = -1;             // generation of the prefilter's operands its K-NN ran on
    bool vit_dense = false, vit_trial = false, vit_judged = false;
    // join_bounds_delay: the Viterbi side of the batch's LAST group and the copy of its results are not queued yet -- the next
    // submit queues them behind a point inside its own first K-NN call, a collect that comes first queues them as they are
    bool tail_pending = false;
    hipEvent_t knn_end = nullptr;         // everything this batch queued on the main stream
    int gbase = 0;                        // added to the group index where it picks a side stream / Viterbi workspace (one-group batches alternate)
    hipEvent_t q_up = nullptr;            // this batch's query rows have arrived in Qall (recorded on the upload stream)
    int64_t total = 0;
    std::vector<int> first;
    std::vector<int64_t> offs;
};

struct ShardTicket {   // one submitted step of the sharded search (snk_sharded_knn_viterbi_batch_submit / _collect)
    bool busy = false, safe = false;
    int G = 1, me = 0, n_utts = 0, K = 0, D = 0, n_own = 0, n_status = 0;
    int64_t R = 0, r_own = 0, slots = 0, rec = 0;
    const double *Q = nullptr;                                   // the caller's query rows (kept valid until collect: a redo reads them)
    std::vector<int64_t> offs, ulo, uhi, rows_to, row0, own_off;
    DevBuf mcand, mdist, res_path, res_plen, res_cost, status;
    HostBuf stage;                                               // own results + status words, pinned: filled by the copy stream
    HostBuf qstage;                                              // the query rows this rank uploads (pinned)
    hipEvent_t main_done = nullptr, side_done[2] = {nullptr, nullptr}, done = nullptr;
};

struct snk_engine {
    int device = 0;
    hipStream_t stream = nullptr, stream2 = nullptr, copy_stream = nullptr;
    hipStream_t up_stream = nullptr;       // query rows of a submitted batch: host -> HBM beside the batch before it (created at the first submit)
    // Optimistic thresholds (api_knn.hip): the filter threshold of the matrix prefilter from the j-th smallest sample minimum,
    // j < K -- an ESTIMATE of the K-th nearest key that makes the lists 3-4 x shorter; the re-rank proves every row's list
    // complete or flags the call, which is then redone with the guaranteed thresholds (the K-th smallest minimum)
    int tau_optimism = 1;                  // option
    int tau_rank_override = 0;             // option tau_optimism_rank (tests): j, 0 = chosen from K and the sample's stride
    bool opt_suppress = false;             // a redo in progress: guaranteed thresholds
    bool opt_off = false;                  // this voice (set of weights) failed too often: guaranteed thresholds until the weights change
    int64_t opt_calls = 0, opt_fails = 0;  // calls that ran optimistic since the weights were set / that had a row flagged
    int64_t opt_fails_total = 0;
    int opt_last_rank = 0;                 // j of the most recent call (0: it ran with guaranteed thresholds)
    int roofline_counters = 0;             // option: the re-rank and pass 3 count what their rooflines are priced on (atomics on one address: off in production)
    int tail_defer = 1;                    // option (api_viterbi.hip snk_knn_viterbi_batch_submit): 0 never / 1 always / 2 while the host keeps up (measured: 2 flaps between the two and loses to both, profiles/r06f_ab.log)
    int64_t submits_seen = 0, submits_starved = 0;      // pipelined submits / those that found the K-NN stream idle
    double starved_ema = 0.0;
    int results_by_kernel = 1;             // option: a batch's results reach page-locked host memory by a kernel's stores, not DMA copies
    // option: 1 = the rows of a submitted batch travel on a stream of their own.  Off: measured on the B* step (profiles/r06d_ab.log) the
    // copy on its own stream costs the HOST 0.31 ms more per submit (0.59 against 0.27 ms) and the step is sensitive to exactly
    // that -- the next batch is submitted when the one before the last is collected, 0.3 ms before the K-NN stream runs dry --:
    // 4.75 M frames/s with it, 4.98 M with the 0.3 ms copy at the head of the main stream
    int upload_stream = 0;
    // Three workspaces: a caller that submits batch i + 2 before it collects batch i never lets the K-NN stream run dry while it
    // waits for a batch's last recursions (with two, the next submit could only follow the collect of the batch before the last --
    // 0.3 ms before the stream ran dry on a fast host, after it on a slow one: 4.1 .. 5.0 M frames/s by the box, profiles/r06f_ab.log)
    BatchSlot bslot[SNK_BATCH_SLOTS];
    int bnext = 0;
    int blast = -1;                        // the slot submitted most recently (its tail may be pending)
    hipEvent_t knn_all_done = nullptr;
    // database
    int64_t N = 0, Njc = 0, Nalloc = 0;
    int Dt = 0, Dj = 0, Dpad = 0, Djpad = 0;
    int Fp = 0, Jp = 0;           // row pitch (floats, multiple of 4) of the unweighted device copies
    DevBuf F_unw, JC_unw, Fw, fnorm, JCw, wt, wj, unit_class;
    bool have_db = false, have_join = false, have_weights = false, have_classes = false;
    int64_t shard_offset = 0, global_N = -1;
    // in-library collectives (snk_comm_init / snk_comm_init_transport)
    int comm_ranks = 0, comm_rank = 0;
    void *nccl_comm = nullptr;            // ncclComm_t
    bool comm_dead = false;               // a local error struck after a collective of a step was queued: the communicator was aborted
    snk_transport transport{};            // caller-provided collectives (functional tests)
    bool have_transport = false;
    DevBuf sh_d2, sh_id, sh_bound, sh_rd2, sh_rid, sh_res, sh_resall;
    ShardTicket sticket[2];
    int snext = 0;
    // replicated global sample (snk_upload_global_sample): stage A of a rank's own rows runs against it
    DevBuf gs_unw, gs_w, gs_norm, gs_tiles, gs_fmax2;
    int64_t gs_rows = 0, gs_slabs = 0;
    bool gs_ready = false;
    // k-nn workspace
    DevBuf Qraw, Qp, Qf, qnorm, thr, gmin, cnt, lkey, lidx, status, qclass, d2tmp, slabctr, pool, poolctl, chunkfill;
    UttSlot slot[8];
    hipStream_t dp_stream[2] = {nullptr, nullptr};
    DevBuf res_path, res_plen, res_cost, Qall, res_status, mcand, mdist, rowflag, exact_rows, exact_scratch;
    DevBuf frames_spec, frames_fzv, cc_in, cc_out;      // waveform-side gather
    int64_t frames_rows = 0; int frames_W = 0;
    int exact_row_fallbacks = 0;
    int pool_overflows = 0;               // K-NN calls whose retry still exhausted the entry pool (all rows served exactly)
    // f16-split prefilter state
    DevBuf a16h, a16l, s16h, s16l, b16h, b16l, eps16, thr32, gmin32, fmax2;
    bool f16_ready = false, cls16_ready = false;
    bool wide16_tried = false;    // ... were asked for since the last snk_set_weights (built lazily: api_core.hip ensure_wide_operands)
    bool wide16_ready = false;    // rows of 257 .. 512 columns: bf16-split operands for the blocked product (knn_wide16b)
    int64_t wide_launches = 0;    // K-NN calls served by it
    DevBuf cls16_full, cls16_samp;      // class id per tile row of the two f32 operands
    int precision = 1;            // 1: f32 prefilter + exact f64 re-rank (default), 0: f64 sweep only
    int nt16 = 4, nt16_eff = 4;
    int64_t n_slabs16 = 0, n_slabs16_a = 0, stride16 = 16;
    double eps_c = 8e-6;          // 2x the analytical f32 bound (knn16_kernels.hip)
    int join_bounds_stream = 1;   // batches: pass 1 of the sparse Viterbi path on 1: the group's side stream, 0: the main (K-NN) stream
    // batches: the Viterbi side of group g starts 0: as soon as its candidates are there (beside stage A of group g + 1),
    // 1: behind the thresholds of group g + 1, 2: behind its bucket pass (beside its re-rank) -- groups that have a successor in the
    // SAME batch only (the side stream cannot wait for what is not queued yet) -- where the shape makes it pay (api_viterbi.hip);
    // 3 / 4: 1 / 2 whatever the shape.  knn_mid: where group g + 1 stands (api_knn.hip)
    int join_bounds_delay = 1;
    int wide_one_group = 0;    // K > 128: a batch that fits one K-NN call is ONE group, batches alternate between the side streams (api_viterbi.hip; measured slower: off)
    int split_one_group = 1;   // a long batch that fits one K-NN call is cut into two groups (K-NN batch entry points only)
    hipEvent_t knn_mid = nullptr;
    bool knn_mid_recorded = false;
    int prefilter = 1;            // 1: bf16-split operands on the bf16 matrix pipe where the shape has a variant, 0: float32 operands
    bool bf16_ready = false;      // a16l / s16l (and gs_tiles_b) hold the bf16-split operands of the current weights
    double eps_c_bf = 4e-6;       // accumulation part of the bound of the bf16-split keys (knn16_kernels.hip: c_acc)
    DevBuf kth16;                 // sharded search: per-row second bound (K-th key of the local list, all-reduced)
    DevBuf ball_c, ball_cn, ball_rad, ball_c16, ball_tq, ball_nq;   // pass 0: tile centres (float64, norms, radii, bf16-split operand), per-row terms
    DevBuf ball_c2, ball_cn2, ball_rad2, ball_s16, ball_mask;       // the balls of 32 consecutive tiles (centres, norms, radii, bf16-split operand), (super ball, query tile) bits
    int64_t ball_supers = 0;      // super balls of the operand (0: not built)
    int prefilter_super_balls = 1;   // 1: the ball pass tests the balls of 32 tiles first and visits the blocks they mark
    DevBuf ball_aq, ball_nql, ball_gmin, ball_bound;                // stage A' (scout): tile list per query tile, keys of their units per row, centre-key minima, the row's bound
    int prefilter_ball_bound = 0; // 1: the thresholds also take the K-th smallest key of the units of the nearest tiles (stage A'; where the ball pass
                                  // runs).  Off by default: at B* it shortens the lists 1785 -> 568 entries per row and costs more (0.48 ms per 9 600 rows)
                                  // than bucket + refine save (0.17 ms); HISTORY.md 4.1c
    int prefilter_balls = 1;      // 1: the tiles' balls list the pairs first; the coarse sweep runs only where they list too many
    double coarse_gate_fraction = 0.10;
    int64_t ball_tiles = 0;       // valid tiles of the ball operand (0: not built)
    double ball_limit = 0.0;      // pairs beyond which the ball pass of the most recent call listed too many
    bool ball_pass_ran = false;
    bool filter_coarse = false;   // this voice's tiles are not compact: the ball pass listed too many pairs once, the coarse sweep lists them since
    // ... and where the coarse sweep lists most pairs too (units in no order at all: a tile holds 32 unrelated frames, nearly every
    // (tile, query tile) pair has SOME unit under SOME row's threshold) the voice goes on to the one-pass three-term sweep: nothing
    // to list, nothing to overflow (an overflowing pair list sent every group of every step through the exact float64 redo)
    bool filter_onepass = false;
    double onepass_gate_fraction = 0.5, coarse_limit = 0.0;
    // The two latches are re-armable (option latch_rearm, default 1).  A voice that sits on a slower filter is probed now and then
    // with the pass it left, in a form that only COUNTS the tile pairs it would list (pair_cap 0: no list, no refine pass): the ball
    // pass for a voice on the coarse sweep (a thirtieth of the database), the coarse sweep for a voice on the one-pass sweep.  A count
    // under half the limit that made the voice leave takes it back; a count above doubles the probe period (16 .. 256 calls).
    // ... and a voice whose tiles are not compact is given an ORDER of its own (option reorder, default 1; kmeans_kernels.hip): when
    // the ball pass lists too many pairs the units are clustered once (per set of weights) and the prefilter's operands rebuilt cluster
    // by cluster; perm[position] = unit, applied by the operand builders and undone by the bucket kernel -- no result changes
    int reorder = 1, reorder_iters = 4;
    DevBuf perm, perm2, km_ws;
    double reorder_radius_before = 0.0, reorder_radius_after = 0.0;      // mean radius of the tiles' balls around the last clustering
    bool perm_ready = false, reorder_pending = false, reorder_done = false, reorder_useless = false;
    int64_t reorders = 0;
    int64_t reorder_failures = 0;     // attempts that failed (allocation, launch, an order that was no permutation): the voice kept its order, the call went on
    int64_t operand_gen = 0;              // generation of the prefilter's operands: what a batch listed is judged only against the operands it ran on
    int latch_rearm = 1;
    int64_t filter_calls = 0, probe_next = 16;
    int probe_period = 16, probe_ran = 0;
    double probe_limit = 0.0;
    int64_t filter_rearms = 0;
    bool coarse_pass_ran = false;
    int onepass_switches = 0;
    int64_t ball_switches = 0;
    DevBuf e1_16, thr1_32, cpairs, cpairctl;   // two-pass filter: per-row coarse margin and threshold, (tile, query tile) pair list
    int prefilter_two_pass = 1;   // 1: bf16-split filter as hi.hi sweep + three-term keys of the tile pairs it lets through (default)
    DevBuf margin_stat;           // tripwire of the prefilter's key bound: [0] rows with room < 2 eps, [1] smallest room / eps (float bits)
    int shard_gather_queries = 1; // sharded steps: 1: every rank uploads the rows of its own utterances and the ranks exchange them, 0: every rank uploads all rows
    int shard_refine = 1;         // 1: snk_sharded_knn_viterbi_batch prunes the shards' lists to that bound before the re-rank
    int shard_compact = 1;        // 1: the lists travel compacted (counts + valid entries; one device -> host copy of the block sizes per step)
    DevBuf sh_cnt, sh_off, sh_tot, sh_totall, sh_plan, sh_pack, sh_rpack, sh_offq;
    double shard_last_sent_mb = 0.0, shard_last_padded_mb = 0.0;   // exchange payload of the most recent sharded step: sent / what the padded lists would have been
    DevBuf gs_tiles_b, cq16, rho16, gs_rho16;   // per-row split coefficient; dropped-piece ratios of the operands
    int f16_fallbacks = 0;
    int last_f16_status = 0;
    // A voice whose candidate lists overflow under the prefilter's thresholds (status bit 1) climbs a ladder instead of going to
    // the float64 sweep call after call: 0 = the shape's default (bf16-split operands, lists of 40 K entries, 2 048 near ties in the exact
    // re-rank), 1 = lists of 8 192 entries and a re-rank tier that takes 8 192 near ties, 2 = the float32 operands (their key error is a third of the bf16-split one: fewer units inside the margin) with
    // lists of 8 192.  Seen at N = 12 M units of SURVEY 8d's generator: consecutive units are 3e-6 apart in d^2 there, the lists
    // grow with the units inside the key margin (2 082 entries at most at 1 M, 3 614 at 8 M, > 4 000 at 12 M).  Per set of weights.
    int knn_level = 0;
    int64_t knn_escalations = 0;
    HostBuf hstage;
    HostBuf up;                   // upload staging (h2d): pinned, bump-allocated, wraps behind a stream wait
    size_t up_used = 0;
    // greedy
    GreedyLayout glay{};
    bool have_glay = false, gtiles_ready = false;
    int64_t qall_rows = -1;               // rows of the batch resident in Qall
    std::vector<double> tsel, jsel;       // snk_set_column_selection: 1 = column takes part (empty: all do)
    DevBuf tmask;                         // tsel on the device (query rows are masked after upload)
    DevBuf Dm, gprev, gblkmin, gblkarg, gpath, gdist, gsync, gtiles;
    DevBuf gshard;            // snk_sharded_greedy: this rank's winner of a step (16 bytes) and the ranks' winners (16 bytes each)
    DevBuf g32_blk, g32_ctl;          // float32 persistent scan: block records + candidate lists, {gen, status}
    DevBuf g32_res;                   // resident scan (greedy_res_kernels.hip): one 16-byte record per workgroup
    int64_t greedy_last_status[16] = {0};   // status words of the most recent one-launch scan (undecided step + 1, rounds, exact windows, watchdog, ...)
    int greedy_last_kernel = 0;             // which scan wrote them: 1 streamed (greedy32_kernel: [4] speculated, [5] several holders), 2 resident (greedy_res_kernel: [4..7] why)
    // tripwire of the float32 scans' bound (greedy32_device.h g32_trip; cleared by snk_reset_timers): exact totals that came out further
    // below the float32 minimum than the bound allows, and the largest share of the bound any weighed window consumed
    int64_t greedy_bound_violations = 0;
    double greedy_bound_max_used = 0.0;
    int greedy_fenced = 0;            // 1: cross-check mode of the one-launch scans: agent-scope fences around every hand-off
    int greedy_resident = 1;          // 1: one utterance against a database that fits the chip's LDS takes the resident scan
    int64_t greedy_resident_launches = 0;
    // hoisted target term of the float32 scan (greedy_hoist_kernels.hip): window norms (per database, layout and
    // weights), left operands and products of the utterances in work
    DevBuf gh_nw, gh_max, gh_aq, gh_qn2, gh_W;
    DevBuf gtiles16;                      // float16 copy of the join tiles (the hoisted scan of streamed databases)
    bool gt16_ready = false, gt16_ok = false, gj_ready = false;
    double g16_delta = 0.0;               // the float16 bound's term: 2^-11 max ||w o S'|| + 2^-25 ||w||
    int greedy_f16 = 1;                   // 1: float16 join tiles where the database is streamed; 2: always (tests); 0: never
    bool gh_ready = false;
    double gh_fwmax2 = 0.0;
    int greedy_test_stall = 0;            // test hook (option greedy_test_stall): one workgroup of the float32 scan never arrives at step 1
    int64_t greedy_stalls = 0;            // launches of the float32 scan ended by their watchdog (a workgroup was not running)
    int64_t greedy_f16_launches = 0;      // ... of them from the float16 join tiles
    int64_t greedy_hoist_launches = 0;    // scans that read the hoisted target term
    int64_t greedy_second_rounds = 0, greedy_exact_windows = 0;     // statistics of the float32 scan's exact decisions
    int greedy_hoist = 1;                 // 1: the float32 scan reads one precomputed target value per window (default)
    int greedy_speculate = 1;             // 1: in float16 scans the workgroup whose minimum is the smallest published so far decides before the gather (default)
    int greedy_hoist_fast = 1;            // 1: scans of float16 join tiles take the target values from the bf16 matrix pipe (default)
    int64_t greedy_hoist16_launches = 0;
    double greedy_hoist_max_gb = 48.0;    // products of one scan group beyond this many GB: the scan computes the target term itself
    int greedy_mode = 2;                  // 2: auto (batches: float32 scan; one utterance: exact scan); 1: float32 prefilter scan in one
                                          // persistent launch (exact decision); 0: exact float64 scan, a launch per step
    int greedy_fallbacks = 0;             // utterance groups the float32 scan could not decide (mass ties) and the exact scan finished
    // options
    int cap = 4096;
    // stage A's sample: every 24th unit.  (1/16 until round 6: with the optimistic thresholds the lists no longer pay for a sparser
    // sample the way they did -- ~650 entries at 1/16, ~780 at 1/24, ~900 at 1/32 for K = 100 -- while the sample sweep costs its
    // share of the database: the B* step 3.84 -> 3.73 ms at 1/24, 3.80 at 1/32, profiles/r06f_ab.log, r06g_ab.log)
    double sample_frac = 1.0 / 24.0;
    int min_sample_slabs = 256;           // small databases / shards: the sample stride shrinks to keep this many sampled slabs
    int nt_override = 0;
    int timers_on = 1;
    unsigned int timers_mask = 0;          // timers 3 (experiments): the stages whose bit is set (option timers_mask)
    int n_cus = 256;
    int reserved_cus = 2;
    int batch_rows = 12288;    // rows per K-NN call of the batch entry points (utterances are grouped)
    int viterbi_weights = 0;   // 0: float64 recursion (default); 1: OpenFST's float32 weight chain (fst_functions_wrapped.py:47,201,368,389), dense kernels
    int viterbi_mode = 2;      // 2: auto; 1: f32 lower bounds on the matrix pipe + sparse exact recursion; 0: dense exact join + recursion
    // viterbi_mode 2, batches: which of the two exact paths a voice's batches take is JUDGED (option viterbi_latch, default 1).  Where
    // the bounds do not prune -- more than vit_refine_gate of a batch's cells refined in pass 4: join rows with no natural successors --
    // the dense kernels are tried for three batches and kept if the batch period (stream time between the completions of consecutive
    // batches) is 5 % shorter; the path not in use is tried again after 32, 64 .. 1 024 batches.  Same results either way.
    int viterbi_latch = 1;
    double vit_refine_gate = 0.01;
    // viterbi_weights 1 on the sparse path: pass 2's predecessor sets are wider by this fraction of the (estimated) absolute
    // total -- two float32 roundings of it sit between an excluded predecessor and the proof (joinfast_kernels.hip).  Speed
    // only; measured at B*: 24.4 ms per batch at 0, 6.3 at 3e-7, 13.8 at 6e-7 (beyond it the four-member sets overflow)
    double fst32_slack = 3e-7;
    struct VitLatch {
        int mode = 0, trial_mode = -1, trial_left = 0;      // 0 sparse, 1 dense
        double ms_row[2] = {0.0, 0.0}, trial_best = 0.0;
        int64_t batches = 0, next_probe = 4, switches = 0, trials = 0;
        int period = 32;
    } vit;
    bool vit_now_dense = false;                             // the batch being submitted takes the dense kernels
    hipEvent_t vit_t0[4] = {nullptr, nullptr, nullptr, nullptr}, vit_t1[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t vit_seq = 0;                                    // batches submitted (ring index of the timing events)
    int64_t vit_last_collected = -1;
    unsigned long long vit_cells_prev = 0;
    double join_beta = 5e-4;   // pass-2 margin in units of the step's largest centred norm (speed only, never the result)
    // pass 2 (approximate recursion) in chunks of viterbi_lb_chunk steps side by side (0: one chain per utterance), each
    // started viterbi_lb_warm steps early; launches of up to viterbi_lb_chunk_max_utts utterances (24 = all).  A single
    // utterance (T = 600, K = 100) 0.93 -> 0.10 ms; a B* step 5.98 -> 5.60 ms.  Up to four utterances: chunks of 32 at most.
    int lb_chunk = 48;
    int lb_chunk_max_utts = 24;
    int lb_warm = 16;
    // the warm-up this VOICE runs with: a voice whose totals remember more than 16 steps (join costs that hardly differ between
    // candidates: speech-like data) fails pass 4's proofs by the tens of thousands with chunks started 16 steps early (profiles/
    // r06b_sweep.log: 61 000 refined cells per 16 utterances, 17 ms; 3 000 and 2 ms with 48) -- a batch that refines more than
    // viterbi_refine_gate of its cells takes the voice to lb_warm_long; back to lb_warm when the weights change
    int lb_warm_eff = 0;       // 0: lb_warm
    int lb_warm_long = 48;
    int64_t lb_warm_raises = 0;
    DevBuf vstats;             // [0] cells refined, [1] steps with a refinement, [2] exact costs computed there
    // pass 1 of the sparse path, second form (joinfast_kernels.hip: join_lb2_kernel): float32 copy of the weighted join rows,
    // built at the first sparse recursion after snk_set_weights; [0] of jw_umax: bits of the largest row norm
    DevBuf JW32, jw_umax;
    bool jw32_ready = false;
    double join_lb_test_scale = 1.0;   // test hook: pass 1's bounds times this factor (> 1: no bounds any more -- the tripwire must fire)
    int join_lb_variant = 1;   // 1: bf16 matrix pipe over the weighted float32 copy (default); 0: float32 matrix pipe, weights applied per gather
    int pool_chunk_limit = 0;  // test hook: cap of the entry pool (chunks) in every attempt; 0 = none
    int pool_chunks = 4096;    // entry pool: 4096 chunks x 2048 entries x 16 B = 128 MiB      // left free by the persistent K-NN sweep for Viterbi DP blocks
    int last_retries = 0;
    int tie_overflow = 0;
    int batch_redos = 0;
    int64_t last_T = 0;
    // timers
    std::vector<EvPair> pending;
    std::vector<hipEvent_t> ev_pool;
    double tm_ms[TM_COUNT] = {0};
    int64_t tm_n[TM_COUNT] = {0};
};

hipEvent_t ev_get(snk_engine *h);

struct StageTimer {     // records an event pair around a stage on a stream
    snk_engine *h; hipStream_t s; EvPair ep; bool on;
    // timers_on 1: every stage; 2: the bounds pass of the Viterbi side only (the bench line's roofline kernel) -- a timed stage is
    // two timestamp events on its stream, ~60 of them per B* step: 4-7 % of the step (profiles/r06h_ab.log: 5.0 -> 5.2-5.4 M frames/s
    // without); 0: none
    StageTimer(snk_engine *h_, hipStream_t s_, int id) : h(h_), s(s_), on((h_->timers_on == 1 || (h_->timers_on == 2 && id == TM_JOIN_LB) || (h_->timers_on == 3 && id >= 0 && ((h_->timers_mask >> id) & 1u))) && id >= 0)
    {
        if (!on) return;
        ep.id = id; ep.a = ev_get(h); ep.b = ev_get(h);
        (void)hipEventRecord(ep.a, s);
    }
    ~StageTimer()
    {
        if (!on) return;
        (void)hipEventRecord(ep.b, s);
        h->pending.push_back(ep);
    }
};

void collect_timers(snk_engine *h);   // call after the streams were synchronised

// device -> pinned staging -> user memory; `parts` are (dst, src, bytes) triples

struct D2HPart { void *dst; const void *src; size_t bytes; };

int staged_d2h(snk_engine *h, hipStream_t st, const D2HPart *parts, int n);

// host <-> device copies through page-locked memory the library owns (api_core.hip)
bool host_memory_is_pinned(const void *p);
int h2d(snk_engine *h, void *dst_dev, const void *src_host, size_t bytes, hipStream_t st);
int h2d_rows(snk_engine *h, void *dst_dev, size_t dst_pitch, const void *src_host, size_t src_pitch, size_t row_bytes,
             size_t n_rows, hipStream_t st);
int h2d_sync(snk_engine *h, void *dst_dev, const void *src_host, size_t bytes);
int d2h_sync(snk_engine *h, void *dst_host, const void *src_dev, size_t bytes, hipStream_t st);
int h2d_via(HostBuf &stage, void *dst_dev, const void *src_host, size_t bytes, hipStream_t st);

// the prefilter's operands for the current weights / an order for a voice whose tiles are not compact (api_core.hip)
int build_prefilter_operands(snk_engine *h);
int ensure_wide_operands(snk_engine *h);
int reorder_units(snk_engine *h);

// state checks (api_core.hip)
int check_ready(snk_engine *h, bool need_target, bool need_join);
int no_batch_in_flight(snk_engine *h, const char *who);
inline int roundup(int64_t v, int64_t m) { return (int)(((v + m - 1) / m) * m); }

// error of ONE v_mfma_f32_32x32x16_bf16, as a fraction of the sum of its |products| and |C|, that the bf16-split bound
// assumes: 2^-20.  Probed (snk_probe_mfma_bf16, tests/test_gpu_prefilter.py): the unit aligns the sixteen products to
// the largest exponent and cuts them two bits below its float32 unit -- up to 0.55 x 2^-20 on patterns built for it.
#define SNK_BF16_MFMA_UNIT 9.5367431640625e-07
#define SNK_KNN_MAX_ROWS 32768      // rows of one K-NN call (batch_rows is capped to it)

// K-NN pipeline on the device (api_knn.hip)
KnnPlan make_plan(snk_engine *h, int K);
void note_ball_pairs(snk_engine *h, unsigned int listed);
// what a call's filter listed (listed) and what its counting probe would have (probe_listed; kind 0: none), judged at a host wait
void judge_filter(snk_engine *h, bool ran_balls, double ball_limit, bool ran_coarse, double coarse_limit, unsigned int listed,
                  int probe_kind, double probe_limit, unsigned int probe_listed);
int upload_queries(snk_engine *h, const double *Q, int64_t T, int D);
int knn_device(snk_engine *h, const double *Qdev, int64_t T, int K, const int32_t *qclass_dev,
               int64_t *cand_dev, double *dist_dev, double *d2_dev, int *deferred_status = nullptr,
               const double *bound_in = nullptr, double *bound_out = nullptr, bool gs = false, bool refine = false,
               unsigned int *pairs_listed_dev = nullptr,      // with deferred_status: receives the tile pairs the first pass listed
               unsigned int *probe_listed_dev = nullptr);     // ... and what the call's counting probe would have listed (0xffffffff: no probe)

// Viterbi side of a group of utterances (api_viterbi.hip)
bool use_sparse_viterbi(const snk_engine *h, int K, int n_utts = 1);
inline bool any_batch_busy(const snk_engine *h) { for (const auto &b : h->bslot) if (b.busy) return true; return false; }
void note_optimism_failure(snk_engine *h);
std::vector<int> group_utterances(const snk_engine *h, const int64_t *row_offsets, int n_utts, bool knn_beside, int K);
int viterbi_group(snk_engine *h, int g, const int64_t *row_offsets, int u0, int u1, int K,
                  const int64_t *cand_all, const double *tdist_all, bool side_stream,
                  int64_t *res_path = nullptr, int64_t *res_plen = nullptr, double *res_cost = nullptr,
                  int n_batch_utts = 2,
                  bool knn_done_recorded = false,        // the slot's knn_done event already marks the end of the group's K-NN
                  hipEvent_t also_behind = nullptr);     // the side stream waits for this one too (join_bounds_delay)

// collectives of the sharded search, on the engine's stream (api_shard.hip)
int comm_all_reduce_min(snk_engine *h, double *buf, int64_t n);
