// C ABI of libsnkhip.so (include/snk.h): host-side engine around the gfx950 kernels.
// Device memory, streams and events are plain HIP; there is no CPU compute fallback.
#include "snk_internal.h"
#include "../../include/snk.h"

#include <dlfcn.h>
#include <float.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>

using namespace snk;

static thread_local std::string g_err;

static int fail(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

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
static const char *kTimerNames[TM_COUNT] = {
    "h2d_queries", "prepare_queries", "knn_minima", "knn_threshold", "knn_filter", "knn_bucket", "knn_finalize",
    "join_costs", "viterbi_dp", "d2h_results", "greedy_target_gemm", "greedy_steps", "set_weights",
    "merge_topk", "join_lower_bounds", "viterbi_lower_bound", "join_exact_sparse", "viterbi_sparse", "knn_ball_bound"};

// Debug allocator (environment SNK_GUARD=1|2|3, read once): every device buffer gets its own virtual range with an
// unmapped page after it (1: the buffer ends where the mapping ends, an over-read or over-write of even one 16-byte
// element faults at once; 2: it starts where the mapping starts) and exactly the bytes asked for -- no growth slack,
// no reuse, fresh memory filled with 0xFF (a NaN / huge-index pattern).  Every allocation is logged with its call site,
// so the page address in the runtime's "Memory access fault" line names the buffer that was overrun.  Speed is of no
// concern in this mode; the product path never sets it.
static int guard_mode()
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

struct BatchSlot {    // one submitted batch (snk_knn_viterbi_batch_submit / _collect): two may be in flight
    DevBuf Qall, cand, dist, path, plen, cost, status;
    HostBuf stage, qstage;                // results / query rows of this batch (pinned: the copies are queued, not waited for)
    hipEvent_t done = nullptr;            // results of this batch are in `stage`
    bool busy = false;
    int n_utts = 0, n_groups = 0, K = 0, D = 0;
    int64_t q_rows = -1; int q_D = 0;     // query rows resident in Qall (a later submit with Q == NULL searches them again)
    std::vector<int64_t> q_offs;
    double ball_limit = -1.0;             // >= 0: the ball pass listed this batch's tile pairs; beyond this many the voice goes to the coarse sweep
    double coarse_limit = -1.0;           // >= 0: the coarse sweep listed them; beyond this many the voice goes to the one-pass sweep
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
    BatchSlot bslot[2];
    int bnext = 0;
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
    bool wide16_ready = false;    // rows of 257 .. 512 columns: bf16-split operands for the blocked product (knn_wide16b)
    int64_t wide_launches = 0;    // K-NN calls served by it
    DevBuf cls16_full, cls16_samp;      // class id per tile row of the two f32 operands
    int precision = 1;            // 1: f32 prefilter + exact f64 re-rank (default), 0: f64 sweep only
    int nt16 = 4, nt16_eff = 4;
    int64_t n_slabs16 = 0, n_slabs16_a = 0, stride16 = 16;
    double eps_c = 8e-6;          // 2x the analytical f32 bound (knn16_kernels.hip)
    int join_bounds_stream = 1;   // batches: pass 1 of the sparse Viterbi path on 1: the group's side stream, 0: the main (K-NN) stream
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
                                  // than bucket + refine save (0.17 ms); DESIGN.md 4.1c
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
    DevBuf g32_blk, g32_ctl;          // float32 persistent scan: block records + candidate lists, {gen, status}
    DevBuf g32_res;                   // resident scan (greedy_res_kernels.hip): one 16-byte record per workgroup
    int64_t greedy_last_status[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // status words of the most recent one-launch scan (undecided step + 1, rounds, exact windows, watchdog)
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
    double sample_frac = 1.0 / 16.0;
    int min_sample_slabs = 256;           // small databases / shards: the sample stride shrinks to keep this many sampled slabs
    int nt_override = 0;
    int timers_on = 1;
    int n_cus = 256;
    int reserved_cus = 2;
    int batch_rows = 12288;    // rows per K-NN call of the batch entry points (utterances are grouped)
    int viterbi_weights = 0;   // 0: float64 recursion (default); 1: OpenFST's float32 weight chain (fst_functions_wrapped.py:47,201,368,389), dense kernels
    int viterbi_mode = 2;      // 2: auto; 1: f32 lower bounds on the matrix pipe + sparse exact recursion; 0: dense exact join + recursion
    double join_beta = 5e-4;   // pass-2 margin in units of the step's largest centred norm (speed only, never the result)
    // pass 2 (approximate recursion) in chunks of viterbi_lb_chunk steps side by side (0: one chain per utterance), each
    // started viterbi_lb_warm steps early; launches of up to viterbi_lb_chunk_max_utts utterances (24 = all).  A single
    // utterance (T = 600, K = 100) 0.93 -> 0.10 ms; a B* step 5.98 -> 5.60 ms.  Up to four utterances: chunks of 32 at most.
    int lb_chunk = 48;
    int lb_chunk_max_utts = 24;
    int lb_warm = 16;
    DevBuf vstats;             // [0] cells refined, [1] steps with a refinement, [2] exact costs computed there
    // pass 1 of the sparse path, second form (joinfast_kernels.hip: join_lb2_kernel): float32 copy of the weighted join rows,
    // built at the first sparse recursion after snk_set_weights; [0] of jw_umax: bits of the largest row norm
    DevBuf JW32, jw_umax;
    bool jw32_ready = false;
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

static hipEvent_t ev_get(snk_engine *h)
{
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

struct StageTimer {     // records an event pair around a stage on a stream
    snk_engine *h; hipStream_t s; EvPair ep; bool on;
    StageTimer(snk_engine *h_, hipStream_t s_, int id) : h(h_), s(s_), on(h_->timers_on != 0 && id >= 0)
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

static void collect_timers(snk_engine *h)   // call after the streams were synchronised
{
    std::vector<EvPair> later;                // stages of a batch that is still in flight (submit / collect)
    for (auto &ep : h->pending) {
        float ms = 0.f;
        const hipError_t e = hipEventElapsedTime(&ms, ep.a, ep.b);
        if (e == hipErrorNotReady) { later.push_back(ep); continue; }
        if (e == hipSuccess) { h->tm_ms[ep.id] += ms; h->tm_n[ep.id] += 1; }
        h->ev_pool.push_back(ep.a);
        h->ev_pool.push_back(ep.b);
    }
    (void)hipGetLastError();
    h->pending.swap(later);
}

// device -> pinned staging -> user memory; `parts` are (dst, src, bytes) triples
struct D2HPart { void *dst; const void *src; size_t bytes; };
static int staged_d2h(snk_engine *h, hipStream_t st, const D2HPart *parts, int n)
{
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (parts[i].bytes + 63) & ~(size_t)63;
    CHK(h->hstage.ensure(total));
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (parts[i].bytes)
            HIPCHK(hipMemcpyAsync((char *)h->hstage.p + off, parts[i].src, parts[i].bytes, hipMemcpyDeviceToHost, st));
        off += (parts[i].bytes + 63) & ~(size_t)63;
    }
    HIPCHK(hipStreamSynchronize(st));
    off = 0;
    for (int i = 0; i < n; ++i) {
        if (parts[i].bytes) memcpy(parts[i].dst, (char *)h->hstage.p + off, parts[i].bytes);
        off += (parts[i].bytes + 63) & ~(size_t)63;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Host <-> device copies.  The GPU (its DMA engines, the runtime's copy kernels) only ever touches page-locked memory
// that THIS library allocated (hipHostMalloc) or that the caller registered (snk_host_register): caller buffers are
// ordinary pageable memory (numpy arrays, stack variables), and handing those to hipMemcpyAsync makes the runtime pin
// and map them on the fly -- a process that frees and reuses such memory all the time (a Python test session, a tuning
// loop) was seen to die with "Memory access fault by GPU node ... on address <an address of the host heap>" inside an
// unrelated call.  Uploads go caller -> pinned staging (memcpy) -> device, results device ->
// pinned staging -> caller (staged_d2h).
// ---------------------------------------------------------------------------
static bool host_memory_is_pinned(const void *p)
{
    static int bypass = -1;                                // developer switch: SNK_NO_STAGING=1 hands caller memory to the runtime as before
    if (bypass < 0) { const char *e = getenv("SNK_NO_STAGING"); bypass = (e && *e == '1') ? 1 : 0; }
    if (bypass) return true;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// the engine's upload staging: a bump allocator over one pinned buffer; it wraps (after waiting for the stream: the
// transfers queued so far read it) when full.  Transfers are queued on `st` (always the engine's main stream).
static int h2d(snk_engine *h, void *dst_dev, const void *src_host, size_t bytes, hipStream_t st)
{
    if (!bytes) return 0;
    if (host_memory_is_pinned(src_host)) {                 // registered by the caller / pinned by us: a plain queued copy
        HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
        return 0;
    }
    const size_t chunk_max = (size_t)64 << 20;
    const char *src = static_cast<const char *>(src_host);
    char *dst = static_cast<char *>(dst_dev);
    while (bytes) {
        const size_t n = bytes < chunk_max ? bytes : chunk_max;
        const size_t need = (n + 255) & ~(size_t)255;
        if (h->up_used + need > h->up.bytes) {
            HIPCHK(hipStreamSynchronize(st));              // everything queued out of the buffer has been read
            h->up_used = 0;
            if (need > h->up.bytes) CHK(h->up.ensure(need));
        }
        char *stage = static_cast<char *>(h->up.p) + h->up_used;
        memcpy(stage, src, n);
        HIPCHK(hipMemcpyAsync(dst, stage, n, hipMemcpyHostToDevice, st));
        h->up_used += need;
        src += n; dst += n; bytes -= n;
    }
    return 0;
}

// rows of row_bytes at src_pitch -> rows at dst_pitch (zero-filled padding); dst rows are contiguous at dst_pitch
static int h2d_rows(snk_engine *h, void *dst_dev, size_t dst_pitch, const void *src_host, size_t src_pitch, size_t row_bytes,
                    size_t n_rows, hipStream_t st)
{
    if (!n_rows || !row_bytes) return 0;
    if (dst_pitch == src_pitch && dst_pitch == row_bytes) return h2d(h, dst_dev, src_host, row_bytes * n_rows, st);
    size_t per = ((size_t)32 << 20) / dst_pitch;
    if (per < 1) per = 1;
    const char *src = static_cast<const char *>(src_host);
    char *dst = static_cast<char *>(dst_dev);
    for (size_t r0 = 0; r0 < n_rows; r0 += per) {
        const size_t n = n_rows - r0 < per ? n_rows - r0 : per;
        const size_t need = (n * dst_pitch + 255) & ~(size_t)255;
        if (h->up_used + need > h->up.bytes) {
            HIPCHK(hipStreamSynchronize(st));
            h->up_used = 0;
            if (need > h->up.bytes) CHK(h->up.ensure(need));
        }
        char *stage = static_cast<char *>(h->up.p) + h->up_used;
        for (size_t r = 0; r < n; ++r) {
            memcpy(stage + r * dst_pitch, src + (r0 + r) * src_pitch, row_bytes);
            if (dst_pitch > row_bytes) memset(stage + r * dst_pitch + row_bytes, 0, dst_pitch - row_bytes);
        }
        HIPCHK(hipMemcpyAsync(dst + r0 * dst_pitch, stage, n * dst_pitch, hipMemcpyHostToDevice, st));
        h->up_used += need;
    }
    return 0;
}

// small synchronous upload (weights, masks, counters): staged, waited for
static int h2d_sync(snk_engine *h, void *dst_dev, const void *src_host, size_t bytes)
{
    CHK(h2d(h, dst_dev, src_host, bytes, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// one device -> host transfer through the pinned staging, waited for
static int d2h_sync(snk_engine *h, void *dst_host, const void *src_dev, size_t bytes, hipStream_t st)
{
    const D2HPart part = {dst_host, src_dev, bytes};
    return staged_d2h(h, st, &part, 1);
}

// an upload whose staging must outlive the call (submit / collect): the caller's own pinned buffer takes the copy
static int h2d_via(HostBuf &stage, void *dst_dev, const void *src_host, size_t bytes, hipStream_t st)
{
    if (!bytes) return 0;
    if (host_memory_is_pinned(src_host)) {
        HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
        return 0;
    }
    CHK(stage.ensure(bytes));
    memcpy(stage.p, src_host, bytes);
    HIPCHK(hipMemcpyAsync(dst_dev, stage.p, bytes, hipMemcpyHostToDevice, st));
    return 0;
}

static int no_batch_in_flight(snk_engine *h, const char *who);
static int create_streams(snk_engine *h);
static bool use_sparse_viterbi(const snk_engine *h, int K, int n_utts);
static int roundup(int64_t v, int64_t m) { return (int)(((v + m - 1) / m) * m); }
// error of ONE v_mfma_f32_32x32x16_bf16, as a fraction of the sum of its |products| and |C|, that the bf16-split bound
// assumes: 2^-20.  Probed (snk_probe_mfma_bf16, tests/test_gpu_prefilter.py): the unit aligns the sixteen products to
// the largest exponent and cuts them two bits below its float32 unit -- up to 0.55 x 2^-20 on patterns built for it.
#define SNK_BF16_MFMA_UNIT 9.5367431640625e-07
#define SNK_KNN_MAX_ROWS 32768      // rows of one K-NN call (batch_rows is capped to it)

// ---------------------------------------------------------------------------
extern "C" {

int snk_abi_version(void) { return 1; }
const char *snk_last_error(void) { return g_err.c_str(); }

int snk_device_count(int *count_out)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count_out = 0; return fail("hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count_out = n;
    return 0;
}

int snk_create(int device_id, snk_handle *out)
{
    if (!out) return fail("snk_create: null handle_out");
    *out = nullptr;
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (n <= 0) return fail("snk_create: no HIP device available (this engine has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail("snk_create: device %d out of range (0..%d)", device_id, n - 1);
    HIPCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("snk_create: device %d is %s; this library is built for gfx950 (MI355X) only",
                    device_id, prop.gcnArchName);
    snk_engine *h = new snk_engine();
    h->device = device_id;
    h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (h->slabctr.ensure(64)) { delete h; return 1; }
    if (h->margin_stat.ensure(2 * sizeof(unsigned int))) { delete h; return 1; }
    int rc = create_streams(h);
    if (!rc) { const unsigned int init[2] = {0u, 0x7f800000u}; rc = h2d_sync(h, h->margin_stat.p, init, sizeof(init)); }
    if (rc) { (void)snk_destroy(h); return rc; }
    *out = h;
    return 0;
}

static int create_streams(snk_engine *h)
{
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    // Two side streams take the T-step recursions of alternate utterance groups.  (ROCm multiplexes
    // the streams of a process onto 4 hardware queues: with more side streams one of them shares a
    // queue with the main stream and a recursion stalls the K-NN sweep queued behind it.)
    HIPCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    for (int i = 0; i < 8; ++i) {
        HIPCHK(hipEventCreateWithFlags(&h->slot[i].knn_done, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->slot[i].vit_done, hipEventDisableTiming));
    }
    h->dp_stream[0] = h->stream2;
    HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->knn_all_done, hipEventDisableTiming));
    for (auto &b : h->bslot) HIPCHK(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
    HIPCHK(hipStreamCreateWithFlags(&h->dp_stream[1], hipStreamNonBlocking));
    return 0;
}

int snk_destroy(snk_handle h)
{
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    collect_timers(h);
    h->tmask.release(); h->mcand.release(); h->mdist.release(); h->vstats.release();
    (void)snk_comm_destroy(h);
    { DevBuf *cb[] = {&h->sh_d2, &h->sh_id, &h->sh_bound, &h->sh_rd2, &h->sh_rid, &h->sh_res, &h->sh_resall,
                      &h->gs_unw, &h->gs_w, &h->gs_norm, &h->gs_tiles, &h->gs_fmax2};
      for (auto *b : cb) b->release(); }
    DevBuf *bufs[] = {&h->F_unw, &h->JC_unw, &h->Fw, &h->fnorm, &h->JCw, &h->wt, &h->wj, &h->unit_class, &h->JW32, &h->jw_umax,
                      &h->Qraw, &h->Qp, &h->Qf, &h->qnorm, &h->thr, &h->gmin, &h->cnt, &h->lkey, &h->lidx,
                      &h->status, &h->qclass, &h->d2tmp, &h->slabctr, &h->pool, &h->poolctl, &h->chunkfill, &h->Dm, &h->gprev, &h->gblkmin, &h->gblkarg,
                      &h->gpath, &h->gdist, &h->gsync, &h->gtiles, &h->cls16_full, &h->cls16_samp,
                      &h->g32_blk, &h->g32_ctl, &h->g32_res,
                      &h->gh_nw, &h->gh_max, &h->gh_aq, &h->gh_qn2, &h->gh_W, &h->gtiles16};
    for (auto *b : bufs) b->release();
    if (h->dp_stream[1]) (void)hipStreamDestroy(h->dp_stream[1]);
    h->res_path.release(); h->res_plen.release(); h->res_cost.release(); h->Qall.release();
    h->rowflag.release(); h->exact_rows.release(); h->exact_scratch.release();
    h->frames_spec.release(); h->frames_fzv.release(); h->cc_in.release(); h->cc_out.release();
    h->res_status.release(); h->hstage.release(); h->up.release();
    { DevBuf *fb[] = {&h->a16h, &h->a16l, &h->s16h, &h->s16l, &h->b16h, &h->b16l, &h->eps16, &h->thr32, &h->gmin32, &h->fmax2, &h->gs_tiles_b, &h->cq16, &h->rho16, &h->gs_rho16, &h->kth16, &h->margin_stat, &h->e1_16, &h->thr1_32, &h->cpairs, &h->cpairctl,
                      &h->ball_c, &h->ball_cn, &h->ball_rad, &h->ball_c16, &h->ball_tq, &h->ball_nq,
                      &h->ball_aq, &h->ball_nql, &h->ball_gmin, &h->ball_bound,
                      &h->ball_c2, &h->ball_cn2, &h->ball_rad2, &h->ball_s16, &h->ball_mask,
                      &h->sh_cnt, &h->sh_off, &h->sh_tot, &h->sh_totall, &h->sh_plan, &h->sh_pack, &h->sh_rpack, &h->sh_offq};
      for (auto *b : fb) b->release(); }
    for (int i = 0; i < 8; ++i) {
        UttSlot &s = h->slot[i];
        DevBuf *sb[] = {&s.cand, &s.tdist, &s.J, &s.bp, &s.path, &s.plen, &s.cost, &s.Jlo, &s.scale, &s.sets, &s.cex};
        for (auto *b : sb) b->release();
        if (s.knn_done) (void)hipEventDestroy(s.knn_done);
        if (s.vit_done) (void)hipEventDestroy(s.vit_done);
    }
    for (auto e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->knn_all_done) (void)hipEventDestroy(h->knn_all_done);
    for (auto &t : h->sticket) {
        t.mcand.release(); t.mdist.release(); t.res_path.release(); t.res_plen.release(); t.res_cost.release(); t.status.release();
        t.stage.release(); t.qstage.release();
        if (t.done) (void)hipEventDestroy(t.done);
        if (t.main_done) (void)hipEventDestroy(t.main_done);
        for (int i = 0; i < 2; ++i) if (t.side_done[i]) (void)hipEventDestroy(t.side_done[i]);
    }
    for (auto &b : h->bslot) { b.Qall.release(); b.cand.release(); b.dist.release(); b.path.release(); b.plen.release(); b.cost.release(); b.status.release(); b.stage.release(); b.qstage.release(); if (b.done) (void)hipEventDestroy(b.done); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    delete h;
    return 0;
}

static int upload_join(snk_engine *h, const float *JC_unw, int64_t Njc, int Dj)
{
    if (!JC_unw || Njc < 2 || Dj < 1) return fail("upload: bad join matrix (Njc=%lld Dj=%d)", (long long)Njc, Dj);
    h->Njc = Njc; h->Dj = Dj; h->Djpad = roundup(Dj, 32);
    h->Jp = roundup(Dj, 4);     // 16-byte aligned rows, zero-filled padding (greedy scan: 128-bit loads)
    CHK(h->JC_unw.ensure((size_t)Njc * h->Jp * sizeof(float)));
    if (h->Jp != Dj) HIPCHK(hipMemsetAsync(h->JC_unw.p, 0, (size_t)Njc * h->Jp * sizeof(float), h->stream));
    CHK(h->JCw.ensure((size_t)Njc * h->Djpad * sizeof(double)));
    CHK(h2d_rows(h, h->JC_unw.p, (size_t)h->Jp * sizeof(float), JC_unw, (size_t)Dj * sizeof(float), (size_t)Dj * sizeof(float),
                 (size_t)Njc, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_join = true;
    h->have_weights = false;
    // everything derived from the join matrix (greedy layout, float32 / float16 join tiles, their norms and range check)
    h->have_glay = false;
    h->gtiles_ready = false; h->gt16_ready = false; h->gt16_ok = false; h->gj_ready = false;
    h->jw32_ready = false;
    return 0;
}

int snk_upload_db(snk_handle h, const float *F_unw, int64_t N, int Dt, const float *JC_unw,
                  int64_t Njc, int Dj)
{
    CHK(no_batch_in_flight(h, "snk_upload_db"));
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!F_unw || N < 1 || Dt < 1) return fail("snk_upload_db: bad target matrix (N=%lld Dt=%d)", (long long)N, Dt);
    // up to 256 columns the matrix sweeps serve the K-NN; wider rows (the doubled join rows of an epoch voice from
    // train_halfphone as a K-NN database: 2 x 151 columns, Synthesiser.join_knn) go through the canonical-distance
    // selection, a workgroup per query row (knn_device)
    if (Dt > 512) return fail("snk_upload_db: Dt=%d > 512 columns is not supported", Dt);
    if (N >= (1LL << 31) - 4096) return fail("snk_upload_db: N=%lld exceeds the 31-bit unit id range", (long long)N);
    if (JC_unw && Njc != N + 1) return fail("snk_upload_db: join_contexts must have N+1 rows (got %lld, N=%lld)", (long long)Njc, (long long)N);
    h->N = N; h->Dt = Dt; h->Dpad = roundup(Dt, SNK_DPAD);
    h->Nalloc = roundup(N, 16) + 16 * SNK_NT_MAX;
    h->Fp = roundup(Dt, 4);
    CHK(h->F_unw.ensure((size_t)N * h->Fp * sizeof(float)));
    if (h->Fp != Dt) HIPCHK(hipMemsetAsync(h->F_unw.p, 0, (size_t)N * h->Fp * sizeof(float), h->stream));
    CHK(h->Fw.ensure((size_t)h->Nalloc * h->Dpad * sizeof(double)));
    CHK(h->fnorm.ensure((size_t)h->Nalloc * sizeof(double)));
    CHK(h2d_rows(h, h->F_unw.p, (size_t)h->Fp * sizeof(float), F_unw, (size_t)Dt * sizeof(float), (size_t)Dt * sizeof(float),
                 (size_t)N, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_db = true;
    h->have_weights = false;
    h->have_classes = false;
    h->bslot[0].q_rows = -1; h->bslot[1].q_rows = -1;      // a new voice: the next batch submit carries its query rows
    h->have_glay = false;
    h->gtiles_ready = false; h->gt16_ready = false;
    h->gh_ready = false; h->gj_ready = false;
    h->gs_rows = 0; h->gs_ready = false;
    if (h->global_N < 0) { h->shard_offset = 0; }
    if (JC_unw) CHK(upload_join(h, JC_unw, Njc, Dj));
    return 0;
}

int snk_upload_join_only(snk_handle h, const float *JC_unw, int64_t Njc, int Dj)
{
    if (!h) return fail("null handle");
    CHK(no_batch_in_flight(h, "snk_upload_join_only"));
    HIPCHK(hipSetDevice(h->device));
    return upload_join(h, JC_unw, Njc, Dj);
}

int snk_set_shard(snk_handle h, int64_t global_row_offset, int64_t global_N)
{
    if (!h) return fail("null handle");
    if (global_row_offset < 0 || global_N < 1) return fail("snk_set_shard: bad arguments");
    // merged lists carry unit ids through 32-bit sort keys (merge_topk_kernel)
    if (global_N >= (1LL << 31)) return fail("snk_set_shard: global_N=%lld exceeds the 31-bit unit id range", (long long)global_N);
    h->shard_offset = global_row_offset;
    h->global_N = global_N;
    return 0;
}

// Stream truncation (truncate_target_streams / truncate_join_streams, synth_simple.py:982-992; the reference
// drops the columns from its weighted copies and from the query rows).  Here the columns stay in place:
// the next snk_set_weights gives them weight 0 and uploaded query rows get them zeroed, so each adds
// exactly +0.0 to every squared distance -- same candidates, distances and paths as dropping them.
// cols: ascending indices of the columns that take part; n < 0: all columns.
int snk_set_column_selection(snk_handle h, const int *tcols, int nt, const int *jcols, int nj)
{
    CHK(no_batch_in_flight(h, "snk_set_column_selection"));
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    auto build = [&](const int *cols, int n, int width, std::vector<double> &sel, const char *what) -> int {
        sel.clear();
        if (n < 0) return 0;
        if (width <= 0) return fail("snk_set_column_selection: no %s matrix uploaded", what);
        if (n > 0 && !cols) return fail("snk_set_column_selection: null %s column list", what);
        sel.assign((size_t)width, 0.0);
        for (int i = 0; i < n; ++i) {
            if (cols[i] < 0 || cols[i] >= width) return fail("snk_set_column_selection: %s column %d outside 0..%d", what, cols[i], width - 1);
            sel[(size_t)cols[i]] = 1.0;
        }
        return 0;
    };
    CHK(build(tcols, nt, h->have_db ? h->Dt : 0, h->tsel, "target"));
    CHK(build(jcols, nj, h->have_join ? h->Dj : 0, h->jsel, "join"));
    if (!h->tsel.empty()) {
        CHK(h->tmask.ensure(h->tsel.size() * sizeof(double)));
        CHK(h2d_sync(h, h->tmask.p, h->tsel.data(), h->tsel.size() * sizeof(double)));
    }
    h->have_weights = false;                 // takes effect with the next snk_set_weights
    // rows resident in the batch workspaces were masked with the selection of their upload: the next submit must carry Q
    h->bslot[0].q_rows = -1; h->bslot[1].q_rows = -1;
    return 0;
}

int snk_set_weights(snk_handle h, const double *wt, int n_wt, const double *wj, int n_wj)
{
    CHK(no_batch_in_flight(h, "snk_set_weights"));
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!h->have_db && !h->have_join) return fail("snk_set_weights: no database uploaded");
    if (h->have_db) {
        if (!wt || n_wt != h->Dt)
            return fail("snk_set_weights: target weight vector has %d entries, database has %d columns", n_wt, h->Dt);
        CHK(h->wt.ensure((size_t)n_wt * sizeof(double)));
        std::vector<double> eff(wt, wt + n_wt);
        if (!h->tsel.empty()) {
            if ((int)h->tsel.size() != n_wt) return fail("snk_set_weights: the target column selection was made for %d columns", (int)h->tsel.size());
            for (int c = 0; c < n_wt; ++c) eff[(size_t)c] *= h->tsel[(size_t)c];
        }
        CHK(h2d_sync(h, h->wt.p, eff.data(), (size_t)n_wt * sizeof(double)));
    }
    if (h->have_join) {
        if (!wj || n_wj != h->Dj)
            return fail("snk_set_weights: join weight vector has %d entries, join_contexts has %d columns", n_wj, h->Dj);
        CHK(h->wj.ensure((size_t)n_wj * sizeof(double)));
        std::vector<double> eff(wj, wj + n_wj);
        if (!h->jsel.empty()) {
            if ((int)h->jsel.size() != n_wj) return fail("snk_set_weights: the join column selection was made for %d columns", (int)h->jsel.size());
            for (int c = 0; c < n_wj; ++c) eff[(size_t)c] *= h->jsel[(size_t)c];
        }
        CHK(h2d_sync(h, h->wj.p, eff.data(), (size_t)n_wj * sizeof(double)));
    }
    {
        StageTimer t(h, h->stream, TM_WEIGHTS);
        if (h->have_db)
            launch_weight_target(h->F_unw.as<float>(), h->Fp, h->N, h->Dt, h->wt.as<double>(), h->Fw.as<double>(),
                                 h->fnorm.as<double>(), h->Nalloc, h->Dpad, nullptr, h->stream);
        if (h->have_join)
            launch_weight_join(h->JC_unw.as<float>(), h->Jp, h->Njc, h->Dj, h->wj.as<double>(), h->JCw.as<double>(),
                               h->Djpad, h->stream);
    }
    HIPCHK(hipGetLastError());
    h->gh_ready = false; h->gj_ready = false;                  // window norms of the hoisted greedy target term follow the target weights
    h->jw32_ready = false;                                     // ... and the float32 copy of the weighted join rows the join weights
    // float32 operands of the prefilter (knn16_kernels.hip): ||f||^2 rides in ONE spare padding column
    h->f16_ready = false;
    h->cls16_ready = false;
    if (h->have_db && h->Dpad <= 256 && h->Dpad - h->Dt >= 1) {
        // tiles per wavefront: the database fragments of a slab stay in registers (32 * Dpad / 64
        // floats per tile and lane), so wider rows leave room for fewer tiles
        const int dch16 = h->Dpad / 64;
        const int nt = (dch16 == 1) ? h->nt16 : (dch16 == 2) ? 2 : 1;
        h->nt16_eff = nt;
        // |key~ - key| <= c (2 |q| Fmax + Fmax^2): operand rounding 2^-24 each and an f32 FMA chain of
        // Dpad + 1 terms; c = 2 x that (8e-6 at Dpad = 64)
        h->eps_c = 2.0 * (double)(h->Dpad + 3) * 5.9604644775390625e-08;
        CHK(h->fmax2.ensure(sizeof(double)));
        launch_fmax(h->fnorm.as<double>(), h->N, h->fmax2.as<double>(), h->stream);
        double fmax2 = 0.0;
        CHK(d2h_sync(h, &fmax2, h->fmax2.p, sizeof(double), h->stream));
        const int64_t slab_rows = 32 * nt;
        h->n_slabs16 = (h->N + slab_rows - 1) / slab_rows;
        int64_t stride = (int64_t)floor(1.0 / h->sample_frac + 0.5);
        if (stride < 1) stride = 1;
        // small databases (one rank's shard of a row-sharded one): keep >= 512 sample groups so the
        // K-th smallest group minimum stays close to the K-th nearest sampled unit
        while (stride > 1 && (h->N / stride) / slab_rows < h->min_sample_slabs) --stride;
        h->stride16 = stride;
        h->n_slabs16_a = (h->N / stride) / slab_rows;
        if (fmax2 < 1.0e30 && h->n_slabs16_a >= 1) {
            const int64_t tiles_b = h->n_slabs16 * nt, tiles_a = h->n_slabs16_a * nt;
            const size_t per_tile = (size_t)8 * 64 * 16 * dch16;
            CHK(h->a16h.ensure(tiles_b * per_tile));
            CHK(h->s16h.ensure(tiles_a * per_tile));
            launch_build_db16(h->Fw.as<double>(), h->fnorm.as<double>(), h->N, h->Dt, h->Dpad, tiles_b, 0, 0, nt,
                              h->a16h.p, h->stream);
            launch_build_db16(h->Fw.as<double>(), h->fnorm.as<double>(), h->N, h->Dt, h->Dpad, tiles_a, stride,
                              2 * h->n_slabs16_a, nt, h->s16h.p, h->stream);
            HIPCHK(hipGetLastError());
            h->f16_ready = true;
            h->bf16_ready = false;
            if (h->prefilter >= 1 && knn_sweep16b_supported(nt, dch16, h->Dt, h->Dpad, false)) {
                // key bound = cq ||f|| (what the split drops: measured, prepare_queries16b_kernel) + c_acc (...):
                // 2^-20 per MFMA over the 4 `terms` MFMAs of a chunk's chain and the norm pieces' 2^-24 (knn16_kernels.hip)
                const int terms = h->prefilter == 2 ? 4 : 3;
                // (chains of one 64-column chunk: 4 `terms` MFMAs; the chunks' sums are added in float32)
                h->eps_c_bf = 1.02 * (SNK_BF16_MFMA_UNIT * (double)(terms * 4 + 1) + 6e-8 * (double)(2 * (h->Dpad / 64) + 1));
                CHK(h->rho16.ensure(2 * sizeof(double)));
                launch_db16b_ratios(h->Fw.as<double>(), h->N, h->Dt, h->Dpad, h->rho16.as<double>(), h->stream);
                CHK(h->a16l.ensure(tiles_b * per_tile));
                CHK(h->s16l.ensure(tiles_a * per_tile));
                launch_build_db16b(h->Fw.as<double>(), h->fnorm.as<double>(), h->N, h->Dt, h->Dpad, tiles_b, 0, 0, nt,
                                   h->a16l.p, h->stream);
                launch_build_db16b(h->Fw.as<double>(), h->fnorm.as<double>(), h->N, h->Dt, h->Dpad, tiles_a, stride,
                                   2 * h->n_slabs16_a, nt, h->s16l.p, h->stream);
                HIPCHK(hipGetLastError());
                h->bf16_ready = true;
                // pass 0 of the two-pass filter: centre and radius of every 32-unit tile, the centres as one more bf16-split
                // operand (its dropped-piece ratios join the database's: one key bound serves both)
                h->ball_tiles = 0;
                h->filter_coarse = false; h->filter_onepass = false;
                if (h->prefilter_balls) {
                    const int64_t vt = (h->N + 31) / 32, ct = (vt + 31) / 32;
                    CHK(h->ball_c.ensure((size_t)vt * h->Dpad * sizeof(double)));
                    CHK(h->ball_cn.ensure((size_t)vt * sizeof(double)));
                    CHK(h->ball_rad.ensure((size_t)vt * sizeof(float)));
                    CHK(h->ball_c16.ensure((size_t)ct * per_tile));
                    launch_build_tile_balls(h->Fw.as<double>(), h->N, h->Dt, h->Dpad, vt, h->ball_c.as<double>(), h->ball_cn.as<double>(),
                                            h->ball_rad.as<float>(), h->stream);
                    launch_db16b_ratios(h->ball_c.as<double>(), vt, h->Dt, h->Dpad, h->rho16.as<double>(), h->stream, true);
                    launch_build_db16b(h->ball_c.as<double>(), h->ball_cn.as<double>(), vt, h->Dt, h->Dpad, ct, 0, 0, nt, h->ball_c16.p, h->stream);
                    HIPCHK(hipGetLastError());
                    h->ball_tiles = vt;
                    // one level up: the balls of 32 consecutive tiles, as one more operand in the same format
                    h->ball_supers = 0;
                    if (ct >= 64) {
                        const int64_t ct2 = (ct + 31) / 32;
                        CHK(h->ball_c2.ensure((size_t)ct * h->Dpad * sizeof(double)));
                        CHK(h->ball_cn2.ensure((size_t)ct * sizeof(double)));
                        CHK(h->ball_rad2.ensure((size_t)ct * sizeof(float)));
                        CHK(h->ball_s16.ensure((size_t)ct2 * per_tile));
                        launch_build_super_balls(h->ball_c.as<double>(), h->ball_rad.as<float>(), h->N, vt, h->Dt, h->Dpad, ct, h->ball_c2.as<double>(),
                                                 h->ball_cn2.as<double>(), h->ball_rad2.as<float>(), h->stream);
                        launch_db16b_ratios(h->ball_c2.as<double>(), ct, h->Dt, h->Dpad, h->rho16.as<double>(), h->stream, true);
                        launch_build_db16b(h->ball_c2.as<double>(), h->ball_cn2.as<double>(), ct, h->Dt, h->Dpad, ct2, 0, 0, nt, h->ball_s16.p, h->stream);
                        HIPCHK(hipGetLastError());
                        h->ball_supers = ct;
                    }
                }
            }
        }
    }
    h->wide16_ready = false;
    if (h->have_db && h->prefilter >= 1 && knn_wide16b_supported(h->Dt, h->Dpad)) {
        // rows of 257 .. 512 columns (Synthesiser.join_knn on the doubled join rows of an epoch voice): bf16-split operands
        // of the whole database and of the stage-A sample, one tile per slab; the blocked product of knn_wide16b serves
        // both stages, the exact float64 re-rank is the one of every other width
        const int terms = h->prefilter == 2 ? 4 : 3;
        h->nt16_eff = 1;
        CHK(h->fmax2.ensure(sizeof(double)));
        launch_fmax(h->fnorm.as<double>(), h->N, h->fmax2.as<double>(), h->stream);
        double fmax2 = 0.0;
        CHK(d2h_sync(h, &fmax2, h->fmax2.p, sizeof(double), h->stream));
        h->n_slabs16 = (h->N + 31) / 32;
        int64_t stride = (int64_t)floor(1.0 / h->sample_frac + 0.5);
        if (stride < 1) stride = 1;
        while (stride > 1 && (h->N / stride) / 32 < h->min_sample_slabs) --stride;
        h->stride16 = stride;
        h->n_slabs16_a = (h->N / stride) / 32;
        if (fmax2 < 1.0e30 && h->n_slabs16_a >= 1) {
            const size_t per_tile = (size_t)8 * 64 * 16 * (h->Dpad / 64);
            h->eps_c_bf = 1.02 * (SNK_BF16_MFMA_UNIT * (double)(terms * 4 + 1) + 6e-8 * (double)(2 * (h->Dpad / 64) + 1));
            CHK(h->rho16.ensure(2 * sizeof(double)));
            launch_db16b_ratios(h->Fw.as<double>(), h->N, h->Dt, h->Dpad, h->rho16.as<double>(), h->stream);
            CHK(h->a16l.ensure(h->n_slabs16 * per_tile));
            CHK(h->s16l.ensure(h->n_slabs16_a * per_tile));
            launch_build_db16b(h->Fw.as<double>(), h->fnorm.as<double>(), h->N, h->Dt, h->Dpad, h->n_slabs16, 0, 0, 1, h->a16l.p, h->stream);
            launch_build_db16b(h->Fw.as<double>(), h->fnorm.as<double>(), h->N, h->Dt, h->Dpad, h->n_slabs16_a, stride,
                               2 * h->n_slabs16_a, 1, h->s16l.p, h->stream);
            HIPCHK(hipGetLastError());
            h->wide16_ready = true;
        }
    }
    h->gs_ready = false;
    if (h->gs_rows > 0 && h->f16_ready) {
        // the replicated global sample in the operand layout of stage A (groups scattered over the sample)
        const int nt = h->nt16_eff;
        h->gs_slabs = h->gs_rows / (32 * nt);
        if (h->gs_slabs >= 1) {
            const int64_t rows_alloc = roundup(h->gs_rows, 16) + 16 * SNK_NT_MAX;
            CHK(h->gs_w.ensure((size_t)rows_alloc * h->Dpad * sizeof(double)));
            CHK(h->gs_norm.ensure((size_t)rows_alloc * sizeof(double)));
            CHK(h->gs_fmax2.ensure(sizeof(double)));
            launch_weight_target(h->gs_unw.as<float>(), h->Fp, h->gs_rows, h->Dt, h->wt.as<double>(), h->gs_w.as<double>(),
                                 h->gs_norm.as<double>(), rows_alloc, h->Dpad, nullptr, h->stream);
            launch_fmax(h->gs_norm.as<double>(), h->gs_rows, h->gs_fmax2.as<double>(), h->stream);
            const int64_t tiles = h->gs_slabs * nt;
            CHK(h->gs_tiles.ensure((size_t)tiles * 8 * 64 * 16 * (h->Dpad / 64)));
            launch_build_db16(h->gs_w.as<double>(), h->gs_norm.as<double>(), h->gs_rows, h->Dt, h->Dpad, tiles, 1,
                              2 * h->gs_slabs, nt, h->gs_tiles.p, h->stream);
            if (h->bf16_ready) {
                CHK(h->gs_rho16.ensure(2 * sizeof(double)));
                launch_db16b_ratios(h->gs_w.as<double>(), h->gs_rows, h->Dt, h->Dpad, h->gs_rho16.as<double>(), h->stream);
                CHK(h->gs_tiles_b.ensure((size_t)tiles * 8 * 64 * 16 * (h->Dpad / 64)));
                launch_build_db16b(h->gs_w.as<double>(), h->gs_norm.as<double>(), h->gs_rows, h->Dt, h->Dpad, tiles, 1,
                                   2 * h->gs_slabs, nt, h->gs_tiles_b.p, h->stream);
            }
            HIPCHK(hipGetLastError());
            h->gs_ready = true;
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    collect_timers(h);
    h->have_weights = true;
    return 0;
}

// Row-sharded databases: every rank also keeps a SAMPLE of the whole database (every s-th unit, chosen by the
// caller: 1/16 of 1 M units x 61 columns is 17 MB) and bounds the K-th nearest key of ITS share of the query
// rows against it -- the bound a single GPU would compute -- instead of every rank bounding every row against
// its own shard's sample.  Takes effect with the next snk_set_weights; rows in the database's column layout.
int snk_upload_global_sample(snk_handle h, const float *F_sample_unw, int64_t n_rows, int Dt)
{
    if (!h) return fail("null handle");
    CHK(no_batch_in_flight(h, "snk_upload_global_sample"));
    HIPCHK(hipSetDevice(h->device));
    if (!h->have_db) return fail("snk_upload_global_sample: upload the database shard first");
    if (!F_sample_unw || n_rows < 1 || Dt != h->Dt) return fail("snk_upload_global_sample: bad sample matrix (rows=%lld Dt=%d, database Dt=%d)", (long long)n_rows, Dt, h->Dt);
    CHK(h->gs_unw.ensure((size_t)n_rows * h->Fp * sizeof(float)));
    if (h->Fp != Dt) HIPCHK(hipMemsetAsync(h->gs_unw.p, 0, (size_t)n_rows * h->Fp * sizeof(float), h->stream));
    CHK(h2d_rows(h, h->gs_unw.p, (size_t)h->Fp * sizeof(float), F_sample_unw, (size_t)Dt * sizeof(float), (size_t)Dt * sizeof(float),
                 (size_t)n_rows, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->gs_rows = n_rows;
    h->gs_ready = false;
    h->have_weights = false;
    return 0;
}

int snk_set_unit_classes(snk_handle h, const int32_t *unit_class, int64_t N)
{
    CHK(no_batch_in_flight(h, "snk_set_unit_classes"));
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!h->have_db || N != h->N) return fail("snk_set_unit_classes: N=%lld does not match the database (%lld)", (long long)N, (long long)h->N);
    CHK(h->unit_class.ensure((size_t)h->Nalloc * sizeof(int32_t)));
    HIPCHK(hipMemsetAsync(h->unit_class.p, 0xff, (size_t)h->Nalloc * sizeof(int32_t), h->stream));
    CHK(h2d(h, h->unit_class.p, unit_class, (size_t)N * sizeof(int32_t), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_classes = true;
    h->cls16_ready = false;
    return 0;
}

// ---------------------------------------------------------------------------
// K-NN pipeline on device.  Q must already be on the device (Qraw).  Results go to the
// given device buffers.  Synchronises the stream once per attempt to read the status word.
// ---------------------------------------------------------------------------
// SNK_TRACE: the entry pool the filter sweep left, checked on the host before the bucket kernel scatters it
static int debug_check_pool(snk_engine *h, int max_chunks, int64_t Tpad, int64_t idx_limit, hipStream_t s)
{
    HIPCHK(hipStreamSynchronize(s));
    unsigned int ctl[2] = {0, 0};
    CHK(d2h_sync(h, ctl, h->poolctl.p, sizeof(ctl), s));
    int used = (int)ctl[0];
    if (used > max_chunks) used = max_chunks;
    std::vector<int> fill((size_t)max_chunks);
    CHK(d2h_sync(h, fill.data(), h->chunkfill.p, fill.size() * sizeof(int), s));
    struct E { double key; int idx; int row; };
    const int chunk = knn_pool_chunk_entries();
    std::vector<E> en((size_t)chunk);
    int64_t total = 0, bad = 0;
    for (int c = 0; c < used; ++c) {
        const int n = fill[(size_t)c];
        if (n < 0 || n > chunk) { fprintf(stderr, "[snk-trace] pool: chunk %d of %d has fill %d (chunk size %d)\n", c, used, n, chunk); ++bad; continue; }
        if (n == 0) continue;
        CHK(d2h_sync(h, en.data(), (const char *)h->pool.p + (size_t)c * chunk * sizeof(E), (size_t)n * sizeof(E), s));
        for (int e = 0; e < n; ++e) {
            ++total;
            if (en[(size_t)e].row < 0 || en[(size_t)e].row >= Tpad || en[(size_t)e].idx < 0 || en[(size_t)e].idx >= idx_limit) {
                if (bad < 8) fprintf(stderr, "[snk-trace] pool: chunk %d entry %d of %d: row %d idx %d key %g (Tpad %lld, idx limit %lld)\n", c, e, n,
                                     en[(size_t)e].row, en[(size_t)e].idx, en[(size_t)e].key, (long long)Tpad, (long long)idx_limit);
                ++bad;
            }
        }
    }
    fprintf(stderr, "[snk-trace] pool: %u chunks handed out (max %d, overflow %u), %lld entries, %lld bad\n", ctl[0], max_chunks, ctl[1], (long long)total, (long long)bad);
    return 0;
}

// what the ball pass of a call listed, seen at the call's next host synchronisation: beyond the limit this voice's
// filter goes back to the coarse sweep (until the weights change)
static void note_ball_pairs(snk_engine *h, unsigned int listed)
{
    if (h->ball_pass_ran && !h->filter_coarse && (double)listed > h->ball_limit) { h->filter_coarse = true; h->ball_switches += 1; }
    if (h->coarse_pass_ran && !h->filter_onepass && (double)listed > h->coarse_limit) { h->filter_onepass = true; h->onepass_switches += 1; }
}

static KnnPlan make_plan(snk_engine *h, int K)
{
    KnnPlan p{};
    p.dch = h->Dpad / 64;
    int nt = (p.dch == 1) ? 4 : (p.dch == 2) ? 4 : (p.dch == 3) ? 2 : 1;
    if (h->nt_override > 0 && p.dch <= 2) {
        if (h->nt_override == 2 || h->nt_override == 4 || (h->nt_override == 8 && p.dch == 1)) nt = h->nt_override;
    }
    p.nt = nt;
    const int64_t slab_rows = 16 * nt;
    p.n_slabs = (h->N + slab_rows - 1) / slab_rows;
    p.row_limit = h->N;
    p.grid_cus = h->n_cus - h->reserved_cus;
    if (p.grid_cus < 1) p.grid_cus = 1;
    p.slab_counter = h->slabctr.as<unsigned int>();
    // stage-A sample: every a_stride-th database row (uniform at single-unit granularity),
    // at least 4K groups of sampled-slab-lane minima (16 groups per slab of 16*nt sampled rows)
    int64_t stride = (int64_t)floor(1.0 / h->sample_frac + 0.5);
    if (stride < 1) stride = 1;
    const int64_t min_rows = ((4 * (int64_t)K + 15) / 16) * slab_rows;
    while (stride > 1 && (h->N + stride - 1) / stride < min_rows) --stride;
    p.a_stride = stride;
    const int64_t sample_rows = (h->N + stride - 1) / stride;
    p.a_count = (sample_rows + slab_rows - 1) / slab_rows;
    return p;
}

// deferred_status != nullptr: enqueue the first attempt only, leave its status word in that device
// int and do NOT synchronise (batch pipeline: the caller checks all words at the end of the batch
// and redoes the rare overflowed utterance synchronously).
//
// Row-sharded databases (snk_knn_local_batch_bounds_dev / _bounded_dev): `bound_out` != nullptr runs stage
// A only and leaves, per row, an upper bound of the K-th nearest key of THIS shard (DBL_MAX where the
// f32 path cannot give one); `bound_in` != nullptr skips stage A and filters against bound_in + eps --
// the caller passes the minimum of the bounds of all shards, which still bounds the K-th nearest key
// of the whole database.  Lists may then hold fewer than K entries (padded with id -1).
// gs (with bound_out): stage A runs against the replicated GLOBAL sample (snk_upload_global_sample) instead of
// this shard's own one -- the bound is then that of the whole database, as on a single GPU.
namespace { int comm_all_reduce_min(snk_engine *h, double *buf, int64_t n); }

// refine (with bound_in, inside snk_sharded_knn_viterbi_batch): between bucket and re-rank the shards agree on a
// second, tighter bound -- the smallest of their lists' K-th keys (knn_kernels.hip knn_local_kth_kernel) -- with one
// more all-reduce per call, and prune their lists to it.
static int knn_device(snk_engine *h, const double *Qdev, int64_t T, int K, const int32_t *qclass_dev,
                      int64_t *cand_dev, double *dist_dev, double *d2_dev, int *deferred_status = nullptr,
                      const double *bound_in = nullptr, double *bound_out = nullptr, bool gs = false, bool refine = false,
                      unsigned int *pairs_listed_dev = nullptr)     // with deferred_status: receives the tile pairs the ball pass listed
{
    if (K < 1 || K > 208) return fail("K-NN: n_candidates=%d outside the supported range 1..208", K);
    if (T > SNK_KNN_MAX_ROWS) {
        // very long query matrices: the per-row workspaces (and the bucket kernel's LDS histogram)
        // are sized for SNK_KNN_MAX_ROWS rows; the search is per row, so it is cut into calls
        for (int64_t r0 = 0; r0 < T; r0 += SNK_KNN_MAX_ROWS) {
            const int64_t rows = (T - r0 < SNK_KNN_MAX_ROWS) ? T - r0 : SNK_KNN_MAX_ROWS;
            CHK(knn_device(h, Qdev + r0 * h->Dt, rows, K, qclass_dev ? qclass_dev + r0 : nullptr,
                           cand_dev ? cand_dev + r0 * K : nullptr, dist_dev ? dist_dev + r0 * K : nullptr,
                           d2_dev ? d2_dev + r0 * K : nullptr, nullptr, bound_in ? bound_in + r0 : nullptr,
                           bound_out ? bound_out + r0 : nullptr, gs, refine));
        }
        if (deferred_status) HIPCHK(hipMemsetAsync(deferred_status, 0, sizeof(int), h->stream));
        return 0;
    }
    if (h->Dpad > 256 && h->wide16_ready && h->precision == 1 && !qclass_dev && !bound_out && !bound_in && !deferred_status &&
        2 * h->n_slabs16_a >= K) {
        // 257 .. 512 columns: stage A and the filter as a blocked bf16-split product (knn_wide16b), bucket and the exact
        // float64 re-rank as for every other width; a list or pool overflow sends the call to the exact selection below
        const int64_t Tpad = roundup(T, 32);
        const KnnPlan p0 = make_plan(h, K);
        int cap = h->cap;
        if (cap < 40 * K) cap = 40 * K < 8192 ? 40 * K : 8192;
        const int64_t G16 = 2 * h->n_slabs16_a;
        const int terms = h->prefilter == 2 ? 4 : 3;
        CHK(h->Qp.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
        CHK(h->qnorm.ensure((size_t)Tpad * sizeof(double)));
        CHK(h->thr.ensure((size_t)Tpad * sizeof(double)));
        CHK(h->cnt.ensure((size_t)Tpad * sizeof(int)));
        CHK(h->lkey.ensure((size_t)Tpad * cap * sizeof(double)));
        CHK(h->lidx.ensure((size_t)Tpad * cap * sizeof(int)));
        CHK(h->status.ensure(sizeof(int)));
        int max_chunks = h->pool_chunks;
        {
            const int64_t want_row = 20 * (int64_t)K > 3072 ? 20 * (int64_t)K : 3072;
            const int64_t need = (Tpad * (cap < want_row ? cap : want_row)) / knn_pool_chunk_entries() + 2048;
            if (need > max_chunks) max_chunks = (int)need;
        }
        if (h->pool_chunk_limit > 0 && max_chunks > h->pool_chunk_limit) max_chunks = h->pool_chunk_limit;
        CHK(h->pool.ensure(knn_pool_bytes(max_chunks)));
        CHK(h->poolctl.ensure(2 * sizeof(unsigned int)));
        CHK(h->chunkfill.ensure((size_t)max_chunks * sizeof(int)));
        CHK(h->b16l.ensure((size_t)(Tpad / 32) * 8 * 64 * 16 * (h->Dpad / 64)));
        CHK(h->eps16.ensure((size_t)Tpad * sizeof(double)));
        CHK(h->cq16.ensure((size_t)Tpad * sizeof(double)));
        CHK(h->thr32.ensure((size_t)Tpad * sizeof(float)));
        CHK(h->gmin32.ensure((size_t)Tpad * G16 * sizeof(float)));
        hipStream_t s = h->stream;
        {
            StageTimer t(h, s, TM_PREP);
            launch_prepare_queries(Qdev, T, h->Dt, h->Qp.as<double>(), nullptr, h->qnorm.as<double>(), Tpad, h->Dpad, s);
            launch_knn_reset(h->cnt.as<int>(), Tpad, h->status.as<int>(), h->poolctl.as<unsigned int>(), h->slabctr.as<unsigned int>(),
                             h->chunkfill.as<int>(), max_chunks, s);
            launch_prepare_queries16b(h->Qp.as<double>(), h->qnorm.as<double>(), T, h->Dt, h->Dpad, h->fmax2.as<double>(),
                                      h->rho16.as<double>(), h->eps_c_bf, h->b16l.p, h->eps16.as<double>(), h->cq16.as<double>(), s);
        }
        {
            StageTimer t(h, s, TM_KNN_MINIMA);
            launch_knn_wide16b(0, terms, p0.grid_cus, h->s16l.p, h->b16l.p, h->Dpad, nullptr, Tpad, h->n_slabs16_a, h->gmin32.as<float>(), G16,
                               nullptr, nullptr, nullptr, 0, knn_pool_chunk_entries(), s);
        }
        {
            StageTimer t(h, s, TM_KNN_THRESHOLD);
            launch_knn_threshold16(h->gmin32.as<float>(), G16, T, Tpad, K, h->eps16.as<double>(), h->thr.as<double>(), h->thr32.as<float>(),
                                   nullptr, nullptr, s);
        }
        {
            StageTimer t(h, s, TM_KNN_FILTER);
            launch_knn_wide16b(1, terms, p0.grid_cus, h->a16l.p, h->b16l.p, h->Dpad, h->thr32.as<float>(), Tpad, h->n_slabs16, nullptr, 0,
                               h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks, knn_pool_chunk_entries(), s);
        }
        {
            StageTimer t(h, s, TM_KNN_BUCKET);
            launch_knn_bucket(h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks, Tpad, h->N, h->cnt.as<int>(),
                              h->lkey.as<double>(), h->lidx.as<int>(), cap, h->status.as<int>(), s);
        }
        {
            StageTimer t(h, s, TM_KNN_FINALIZE);
            launch_knn_finalize(h->Fw.as<double>(), h->F_unw.as<float>(), h->Fp, h->wt.as<double>(), h->Dpad, h->Dt, h->Qp.as<double>(),
                                h->qnorm.as<double>(), T, K, h->cnt.as<int>(), h->lkey.as<double>(), h->lidx.as<int>(), cap, h->shard_offset,
                                h->eps16.as<double>(), h->fnorm.as<double>(), h->eps_c_bf, h->cq16.as<double>(), cand_dev, dist_dev, d2_dev,
                                h->status.as<int>(), nullptr, s, false, h->thr.as<double>(), h->margin_stat.as<unsigned int>());
        }
        int status = 0;
        CHK(d2h_sync(h, &status, h->status.p, sizeof(int), s));
        HIPCHK(hipGetLastError());
        h->last_retries = 0; h->last_T = T;
        h->last_f16_status = status;
        if (status == 0) { h->wide_launches += 1; return 0; }
        h->f16_fallbacks += 1;               // overflow or too many near ties: the exact selection below serves the call
    }
    if (h->Dpad > 256) {
        // Rows wider than a database row's fragments fit a wavefront's registers: every row through the exact selection
        // (knn_exact_rows_kernel: canonical distances to every unit by one workgroup per query row, radix select, ties by
        // lowest id) -- the all-pairs join K-NN of active_learning_join.py:184-212 on 302-column join rows.  No bounds
        // for a sharded caller (nothing is pruned), nothing deferred.
        if (bound_out) { launch_fill_threshold(bound_out, T, T, DBL_MAX, h->stream); return 0; }
        const int64_t Tp = roundup(T, 16);
        CHK(h->Qp.ensure((size_t)Tp * h->Dpad * sizeof(double)));
        CHK(h->qnorm.ensure((size_t)Tp * sizeof(double)));
        launch_prepare_queries(Qdev, T, h->Dt, h->Qp.as<double>(), nullptr, h->qnorm.as<double>(), Tp, h->Dpad, h->stream);
        // rows per launch: as many workgroups as the scratch (one float64 per unit and row) allows within 2 GB
        int64_t per = ((int64_t)2 << 30) / ((int64_t)h->Nalloc * 8);
        per = per < 1 ? 1 : per > 512 ? 512 : per;
        CHK(h->exact_rows.ensure((size_t)per * sizeof(int)));
        CHK(h->exact_scratch.ensure((size_t)per * h->Nalloc * sizeof(double)));
        std::vector<int> rows((size_t)per);
        for (int64_t r0 = 0; r0 < T; r0 += per) {
            const int n = (int)(T - r0 < per ? T - r0 : per);
            for (int i = 0; i < n; ++i) rows[(size_t)i] = (int)(r0 + i);
            CHK(h2d(h, h->exact_rows.p, rows.data(), (size_t)n * sizeof(int), h->stream));
            launch_knn_exact_rows(h->Fw.as<double>(), h->Dpad, h->Dt, h->N, h->Qp.as<double>(), h->exact_rows.as<int>(), n, K,
                                  h->exact_scratch.as<double>(), h->Nalloc, qclass_dev ? h->unit_class.as<int32_t>() : nullptr,
                                  qclass_dev, h->shard_offset, cand_dev, dist_dev, d2_dev, h->stream);
            HIPCHK(hipStreamSynchronize(h->stream));          // (the host array of row numbers is reused)
        }
        HIPCHK(hipGetLastError());
        if (deferred_status) HIPCHK(hipMemsetAsync(deferred_status, 0, sizeof(int), h->stream));
        h->last_retries = 0; h->last_T = T;
        return 0;
    }
    const int64_t Tpad = roundup(T, 32);
    const KnnPlan p0 = make_plan(h, K);
    const bool cls = qclass_dev != nullptr;
    const int32_t *uc = cls ? h->unit_class.as<int32_t>() : nullptr;
    int cap = h->cap;
    if (cap < 2 * K) cap = 2 * K;
    // the sampled thresholds let ~17 K candidates per row through (mean; 3200 at K = 200): large K needs
    // longer lists and a bigger pool share than the defaults sized for K <= 128
    // (40 K: the bf16-split prefilter's wider key margin lengthens the lists by a fifth)
    if (cap < 40 * K) cap = 40 * K < 8192 ? 40 * K : 8192;
    if (K > 4096) return fail("K-NN: K too large");
    KnnPlan p = p0;
    int64_t G = p.a_count * 16;
    CHK(h->Qp.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
    CHK(h->Qf.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
    CHK(h->qnorm.ensure((size_t)Tpad * sizeof(double)));
    CHK(h->thr.ensure((size_t)Tpad * sizeof(double)));
    CHK(h->gmin.ensure((size_t)Tpad * G * sizeof(double)));
    CHK(h->cnt.ensure((size_t)Tpad * sizeof(int)));
    CHK(h->lkey.ensure((size_t)Tpad * cap * sizeof(double)));
    CHK(h->lidx.ensure((size_t)Tpad * cap * sizeof(int)));
    CHK(h->status.ensure(sizeof(int)));
    CHK(h->rowflag.ensure((size_t)Tpad * sizeof(int)));
    // entry pool: room for ~3K survivors per row plus one partly filled chunk per resident wave
    int max_chunks = h->pool_chunks;
    {
        const int64_t want_row = 20 * (int64_t)K > 3072 ? 20 * (int64_t)K : 3072;
        const int64_t per_row = cap < want_row ? cap : want_row;
        const int64_t need = (Tpad * per_row) / knn_pool_chunk_entries() + 2048;
        if (need > max_chunks) max_chunks = (int)need;
    }
    if (h->pool_chunk_limit > 0 && max_chunks > h->pool_chunk_limit) max_chunks = h->pool_chunk_limit;
    CHK(h->pool.ensure(knn_pool_bytes(max_chunks)));
    CHK(h->poolctl.ensure(2 * sizeof(unsigned int)));
    CHK(h->chunkfill.ensure((size_t)max_chunks * sizeof(int)));
    hipStream_t s = h->stream;
    {
        StageTimer t(h, s, TM_PREP);
        launch_prepare_queries(Qdev, T, h->Dt, h->Qp.as<double>(), h->Qf.as<double>(), h->qnorm.as<double>(),
                               Tpad, h->Dpad, s);
    }
    h->last_retries = 0;
    h->last_T = T;
    int *status_dev = deferred_status ? deferred_status : h->status.as<int>();

    // ---- fast path: f16-split prefilter (exact results through the float64 re-rank) ----
    // class-restricted searches run the 2-tile (one chunk) / 2- / 1-tile variants of the f32 sweep
    const int dch16 = h->Dpad / 64;
    const int nt_run = (cls && dch16 == 1) ? 2 : h->nt16_eff;
    const int slab_factor = h->nt16_eff / (nt_run > 0 ? nt_run : 1);
    const bool use_gs = gs && bound_out && !cls && h->gs_ready && 2 * h->gs_slabs >= K;
    if (h->precision == 1 && h->f16_ready && (!cls || (h->nt16_eff % nt_run) == 0) &&
        (use_gs || 2 * h->n_slabs16_a * slab_factor >= K)) {
        const int64_t n_slabs_a = use_gs ? h->gs_slabs : h->n_slabs16_a * slab_factor, n_slabs_b = h->n_slabs16 * slab_factor;
        const int64_t G16 = 2 * n_slabs_a;
        const int32_t *cls_full = nullptr, *cls_samp = nullptr;
        if (cls) {
            if (!h->cls16_ready) {
                const int64_t tiles_b = h->n_slabs16 * h->nt16_eff, tiles_a = h->n_slabs16_a * h->nt16_eff;
                CHK(h->cls16_full.ensure((size_t)tiles_b * 32 * sizeof(int32_t)));
                CHK(h->cls16_samp.ensure((size_t)tiles_a * 32 * sizeof(int32_t)));
                launch_build_class16(h->unit_class.as<int32_t>(), h->N, tiles_b, 0, 0, h->nt16_eff,
                                     h->cls16_full.as<int32_t>(), s);
                launch_build_class16(h->unit_class.as<int32_t>(), h->N, tiles_a, h->stride16, 2 * h->n_slabs16_a,
                                     h->nt16_eff, h->cls16_samp.as<int32_t>(), s);
                h->cls16_ready = true;
            }
            cls_full = h->cls16_full.as<int32_t>();
            cls_samp = h->cls16_samp.as<int32_t>();
        }
        const bool bf = h->bf16_ready && h->prefilter >= 1 && !cls && nt_run == h->nt16_eff;
        const double eps_c_run = bf ? h->eps_c_bf : h->eps_c;
        // two-pass filter (knn16_kernels.hip): not for stage-A-only calls
        const bool coarse = bf && h->prefilter_two_pass && !bound_out && knn_coarse16b_supported(nt_run, dch16) && !h->filter_onepass;
        const int64_t n_tiles_b = n_slabs_b * nt_run;
        unsigned int pair_cap = 0;
        if (!coarse) { h->ball_pass_ran = false; h->coarse_pass_ran = false; }      // (nothing listed by this call: nothing to judge the voice by)
        if (coarse) {
            const int64_t all = (Tpad / 32) * n_tiles_b;
            int64_t capp = all / 4 > ((int64_t)4 << 20) ? all / 4 : ((int64_t)4 << 20);
            if (capp > all) capp = all;
            if (capp > ((int64_t)1 << 31) - 1) capp = ((int64_t)1 << 31) - 1;
            pair_cap = (unsigned int)capp;
            CHK(h->cpairs.ensure((size_t)pair_cap * knn_coarse_pair_bytes()));
            CHK(h->cpairctl.ensure(4 * sizeof(unsigned int)));
            CHK(h->e1_16.ensure((size_t)Tpad * sizeof(double)));
            CHK(h->thr1_32.ensure((size_t)Tpad * sizeof(float)));
            HIPCHK(hipMemsetAsync(h->cpairctl.p, 0, 4 * sizeof(unsigned int), s));
        }
        // accumulation of the coarse pass's own chain (one MFMA per k-block through C) on top of the three-term chain's
        const double c_coarse = eps_c_run + 1.02 * SNK_BF16_MFMA_UNIT * (double)(h->Dpad / 16 + 1);
        CHK((bf ? h->b16l : h->b16h).ensure((size_t)(Tpad / 32) * 8 * 64 * 16 * (h->Dpad / 64)));
        CHK(h->eps16.ensure((size_t)Tpad * sizeof(double)));
        if (bf) CHK(h->cq16.ensure((size_t)Tpad * sizeof(double)));
        CHK(h->thr32.ensure((size_t)Tpad * sizeof(float)));
        CHK(h->gmin32.ensure((size_t)Tpad * G16 * sizeof(float)));
        launch_knn_reset(h->cnt.as<int>(), Tpad, status_dev, h->poolctl.as<unsigned int>(),
                         h->slabctr.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks, s);
        {
            StageTimer t(h, s, TM_PREP);
            if (bf)
                launch_prepare_queries16b(h->Qp.as<double>(), h->qnorm.as<double>(), T, h->Dt, h->Dpad,
                                          use_gs ? h->gs_fmax2.as<double>() : h->fmax2.as<double>(),
                                          use_gs ? h->gs_rho16.as<double>() : h->rho16.as<double>(), eps_c_run, h->b16l.p,
                                          h->eps16.as<double>(), h->cq16.as<double>(), s, c_coarse, coarse ? h->e1_16.as<double>() : nullptr);
            else
            launch_prepare_queries16(h->Qp.as<double>(), h->qnorm.as<double>(), T, h->Dt, h->Dpad,
                                     use_gs ? h->gs_fmax2.as<double>() : h->fmax2.as<double>(), h->eps_c, h->b16h.p,
                                     h->eps16.as<double>(), s);
        }
        if (!bound_in) {
            StageTimer t(h, s, TM_KNN_MINIMA);
            if (bf)
                launch_knn_sweep16b(0, h->prefilter == 2 ? 4 : 3, nt_run, dch16, p0.grid_cus, use_gs ? h->gs_tiles_b.p : h->s16l.p, h->b16l.p, nullptr, Tpad,
                                    n_slabs_a, h->slabctr.as<unsigned int>(), h->gmin32.as<float>(), G16, nullptr, nullptr,
                                    nullptr, 0, knn_pool_chunk_entries(), s);
            else
            launch_knn_sweep16(0, nt_run, dch16, (h->Dt + 2) / 2, p0.grid_cus, use_gs ? h->gs_tiles.p : h->s16h.p, h->b16h.p, cls_samp, qclass_dev,
                               nullptr, Tpad, n_slabs_a, h->slabctr.as<unsigned int>(), h->gmin32.as<float>(),
                               G16, nullptr, nullptr, nullptr, 0, knn_pool_chunk_entries(), s);
        }
        // stage A': where the ball pass is going to list the tile pairs (compact tiles), the K-th smallest key among the units
        // of the tiles nearest to a row is a second, much tighter bound of its K-th nearest key
        const bool balls = coarse && h->prefilter_balls && h->ball_tiles > 0 && slab_factor == 1 && !h->filter_coarse;
        const bool ball_bound = balls && h->prefilter_ball_bound && !bound_in && knn_scout_keys_per_row() >= K;
        if (ball_bound) {
            StageTimer t(h, s, TM_KNN_BALLMIN);
            const int Gb = knn_scout_groups(Tpad, h->ball_tiles), G2 = knn_scout_keys_per_row();
            CHK(h->ball_gmin.ensure((size_t)Tpad * Gb * sizeof(float)));
            CHK(h->ball_aq.ensure(knn_scout_list_bytes(Tpad)));
            CHK(h->ball_nql.ensure((size_t)Tpad * G2 * sizeof(float)));
            CHK(h->ball_bound.ensure((size_t)Tpad * sizeof(double)));
            launch_knn_scout16b(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->ball_c16.p, h->a16l.p, h->b16l.p, T, Tpad, h->ball_tiles,
                                h->ball_gmin.as<float>(), h->ball_aq.as<unsigned int>(), h->ball_nql.as<float>(), s);
            // K-th smallest of the row's keys + eps -> ball_bound (thr / thr32 are written again below)
            launch_knn_threshold16(h->ball_nql.as<float>(), G2, T, Tpad, K, h->eps16.as<double>(), h->thr.as<double>(),
                                   h->thr32.as<float>(), nullptr, h->ball_bound.as<double>(), s);
        }
        {
            StageTimer t(h, s, TM_KNN_THRESHOLD);
            launch_knn_threshold16(h->gmin32.as<float>(), G16, T, Tpad, K, h->eps16.as<double>(), h->thr.as<double>(),
                                   h->thr32.as<float>(), bound_in, bound_out, s, coarse ? h->e1_16.as<double>() : nullptr,
                                   coarse ? h->thr1_32.as<float>() : nullptr, ball_bound ? h->ball_bound.as<double>() : nullptr);
        }
        if (bound_out) return 0;             // stage A only
        {
            StageTimer t(h, s, TM_KNN_FILTER);
            if (coarse) {
                if (balls) {
                    CHK(h->ball_tq.ensure((size_t)Tpad * sizeof(float)));
                    CHK(h->ball_nq.ensure((size_t)Tpad * sizeof(float)));
                    launch_ball_query_terms(h->thr32.as<float>(), h->eps16.as<double>(), h->qnorm.as<double>(), T, Tpad, h->ball_tq.as<float>(),
                                            h->ball_nq.as<float>(), s);
                    const unsigned int *visit = nullptr;
                    if (h->prefilter_super_balls && h->ball_supers > 0) {
                        // the balls of 32 tiles first: a bit per (super ball, query tile); the tile pass visits the marked blocks
                        const size_t words = (size_t)h->ball_supers * ((Tpad / 32 + 31) / 32);
                        CHK(h->ball_mask.ensure(words * sizeof(unsigned int)));
                        HIPCHK(hipMemsetAsync(h->ball_mask.p, 0, words * sizeof(unsigned int), s));
                        launch_knn_balls16b(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->ball_s16.p, h->b16l.p, h->ball_rad2.as<float>(),
                                            h->ball_tq.as<float>(), h->ball_nq.as<float>(), Tpad, h->ball_supers, nullptr, nullptr, 0u, s,
                                            h->ball_mask.as<unsigned int>(), nullptr);
                        visit = h->ball_mask.as<unsigned int>();
                    }
                    launch_knn_balls16b(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->ball_c16.p, h->b16l.p, h->ball_rad.as<float>(),
                                        h->ball_tq.as<float>(), h->ball_nq.as<float>(), Tpad, h->ball_tiles, h->cpairs.p,
                                        h->cpairctl.as<unsigned int>(), pair_cap, s, nullptr, visit);
                    h->ball_limit = h->coarse_gate_fraction * (double)(Tpad / 32) * (double)h->ball_tiles;
                }
                h->ball_pass_ran = balls;
                h->coarse_pass_ran = !balls;
                if (!balls) {
                    // beyond half of all pairs the one-pass sweep is the cheaper filter; the list must not overflow either
                    const double all = (double)(Tpad / 32) * (double)n_tiles_b;
                    h->coarse_limit = h->onepass_gate_fraction * all < 0.9 * (double)pair_cap ? h->onepass_gate_fraction * all : 0.9 * (double)pair_cap;
                }
                launch_knn_filter16c(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->a16l.p, h->b16l.p, h->thr32.as<float>(), h->thr1_32.as<float>(),
                                     Tpad, n_tiles_b, h->slabctr.as<unsigned int>() + 1, h->cpairs.p, h->cpairctl.as<unsigned int>(), pair_cap,
                                     h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks, knn_pool_chunk_entries(), s,
                                     !balls);
            }
            else if (bf)
                launch_knn_sweep16b(1, h->prefilter == 2 ? 4 : 3, nt_run, dch16, p0.grid_cus, h->a16l.p, h->b16l.p, h->thr32.as<float>(), Tpad, n_slabs_b,
                                    h->slabctr.as<unsigned int>() + 1, nullptr, 0, h->pool.p, h->poolctl.as<unsigned int>(),
                                    h->chunkfill.as<int>(), max_chunks, knn_pool_chunk_entries(), s);
            else
            launch_knn_sweep16(1, nt_run, dch16, (h->Dt + 2) / 2, p0.grid_cus, h->a16h.p, h->b16h.p, cls_full, qclass_dev,
                               h->thr32.as<float>(), Tpad, n_slabs_b, h->slabctr.as<unsigned int>() + 1, nullptr, 0,
                               h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks,
                               knn_pool_chunk_entries(), s);
        }
        if (trace_on()) CHK(debug_check_pool(h, max_chunks, Tpad, n_slabs_b * 32 * nt_run, s));
        {
            StageTimer t(h, s, TM_KNN_BUCKET);
            launch_knn_bucket(h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks,
                              Tpad, h->N, h->cnt.as<int>(), h->lkey.as<double>(), h->lidx.as<int>(), cap, status_dev, s);
        }
        if (refine && bound_in && h->comm_ranks > 1 && h->shard_refine) {
            StageTimer t(h, s, TM_KNN_BUCKET);
            CHK(h->kth16.ensure((size_t)Tpad * sizeof(double)));
            launch_knn_local_kth(h->cnt.as<int>(), h->lkey.as<double>(), cap, K, h->eps16.as<double>(), T, h->kth16.as<double>(), s);
            CHK(comm_all_reduce_min(h, h->kth16.as<double>(), T));
            launch_knn_list_prune(h->cnt.as<int>(), h->lkey.as<double>(), h->lidx.as<int>(), cap, h->kth16.as<double>(),
                                  h->eps16.as<double>(), T, s);
        }
        {
            StageTimer t(h, s, TM_KNN_FINALIZE);
            launch_knn_finalize(h->Fw.as<double>(), h->F_unw.as<float>(), h->Fp, h->wt.as<double>(), h->Dpad, h->Dt, h->Qp.as<double>(), h->qnorm.as<double>(), T, K,
                                h->cnt.as<int>(), h->lkey.as<double>(), h->lidx.as<int>(), cap,
                                h->shard_offset, h->eps16.as<double>(), h->fnorm.as<double>(), eps_c_run, bf ? h->cq16.as<double>() : nullptr, cand_dev, dist_dev, d2_dev, status_dev, nullptr, s,
                                bound_in != nullptr,         // a shard's lists under the shared bound are short
                                bound_in ? nullptr : h->thr.as<double>(), h->margin_stat.as<unsigned int>(), h->rowflag.as<int>());
        }
        if (deferred_status) {               // the batch caller redoes failures with precision 0
            // (and learns how many tile pairs the ball pass listed)
            if (pairs_listed_dev) {
                if (coarse) HIPCHK(hipMemcpyAsync(pairs_listed_dev, h->cpairctl.p, sizeof(unsigned int), hipMemcpyDeviceToDevice, s));
                else HIPCHK(hipMemsetAsync(pairs_listed_dev, 0, sizeof(unsigned int), s));
            }
            return 0;
        }
        int status = 0;
        {
            unsigned int listed = 0;
            D2HPart parts[2] = {{&status, h->status.p, sizeof(int)}, {&listed, h->cpairctl.p, coarse ? sizeof(unsigned int) : 0}};
            CHK(staged_d2h(h, s, parts, 2));
            note_ball_pairs(h, listed);
        }
        HIPCHK(hipGetLastError());
        h->last_f16_status = status;
        if (status == 0) return 0;
        h->f16_fallbacks += 1;               // overflow or too many near ties: exact f64 sweep below
    }

    if (bound_out) {                          // no f32 path for this shape: no bound, nothing is pruned
        launch_fill_threshold(bound_out, T, T, DBL_MAX, s);
        return 0;
    }
    if (refine && bound_in && h->comm_ranks > 1 && h->shard_refine) {
        // this rank's shard has no prefilter lists (shape without a variant): it still takes part in the other
        // ranks' all-reduce of the second bound, contributing nothing
        CHK(h->kth16.ensure((size_t)Tpad * sizeof(double)));
        launch_fill_threshold(h->kth16.as<double>(), T, T, DBL_MAX, s);
        CHK(comm_all_reduce_min(h, h->kth16.as<double>(), T));
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        // attempt 0: thresholds from a strided sample of slabs (stage A).
        // attempt 1 (a candidate list overflowed): stage A over EVERY slab -- at most
        //   nt*K database rows then lie under each threshold, which the lists always hold.
        launch_knn_reset(h->cnt.as<int>(), Tpad, status_dev, h->poolctl.as<unsigned int>(),
                         h->slabctr.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks, s);
        if (attempt == 1) {
            p.a_stride = 1; p.a_count = p.n_slabs;
            G = p.a_count * 16;
            CHK(h->gmin.ensure((size_t)Tpad * G * sizeof(double)));
            h->last_retries = 1;
            // the first attempt may have exhausted the entry pool (mass ties at the thresholds): give the
            // retry room for a full list per row, so that only the lists themselves can still overflow
            const int64_t full = (Tpad * (int64_t)cap) / knn_pool_chunk_entries() + 4096;
            if (full > max_chunks && h->pool_chunk_limit <= 0) {
                max_chunks = (int)full;
                CHK(h->pool.ensure(knn_pool_bytes(max_chunks)));
                CHK(h->chunkfill.ensure((size_t)max_chunks * sizeof(int)));
            }
        }
        if (G >= K) {       // tiny databases: fewer than K groups cannot bound the K-th neighbour
            {
                StageTimer t(h, s, TM_KNN_MINIMA);
                launch_knn_minima(p, h->Fw.as<double>(), h->fnorm.as<double>(), h->Qf.as<double>(), Tpad,
                                  h->gmin.as<double>(), G, uc, qclass_dev, s);
            }
            {
                StageTimer t(h, s, TM_KNN_THRESHOLD);
                launch_knn_threshold(h->gmin.as<double>(), G, T, Tpad, K, h->thr.as<double>(), attempt, s);
            }
        } else {
            launch_fill_threshold(h->thr.as<double>(), T, Tpad, DBL_MAX, s);
        }
        {
            StageTimer t(h, s, TM_KNN_FILTER);
            launch_knn_filter(p, h->Fw.as<double>(), h->fnorm.as<double>(), h->Qf.as<double>(),
                              h->thr.as<double>(), Tpad, h->pool.p, h->poolctl.as<unsigned int>(),
                              h->chunkfill.as<int>(), max_chunks, uc, qclass_dev, s);
        }
        {
            StageTimer t(h, s, TM_KNN_BUCKET);
            launch_knn_bucket(h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks,
                              Tpad, h->N, h->cnt.as<int>(), h->lkey.as<double>(), h->lidx.as<int>(), cap,
                              status_dev, s);
        }
        {
            StageTimer t(h, s, TM_KNN_FINALIZE);
            launch_knn_finalize(h->Fw.as<double>(), h->F_unw.as<float>(), h->Fp, h->wt.as<double>(), h->Dpad, h->Dt, h->Qp.as<double>(), h->qnorm.as<double>(), T, K,
                                h->cnt.as<int>(), h->lkey.as<double>(), h->lidx.as<int>(), cap,
                                h->shard_offset, nullptr, nullptr, 0.0, nullptr, cand_dev, dist_dev, d2_dev, status_dev, h->rowflag.as<int>(), s);
        }
        if (deferred_status) return 0;
        int status = 0;
        CHK(d2h_sync(h, &status, h->status.p, sizeof(int), s));
        HIPCHK(hipGetLastError());
        if (status == 0) return 0;
        if ((status & 5) && attempt == 0) continue;          // a list or the pool overflowed: exact thresholds next
        // Rows the list pipeline cannot serve: more units tied with (or within rounding of) the K-th
        // neighbour than a list or the exact re-rank holds -- mass duplicates.  They get the
        // one-workgroup-per-row exact selection (slow, exact, ties by lowest id).  If even the enlarged
        // pool overflowed, the sweep dropped entries of rows that cannot be told apart: every row goes.
        std::vector<int> flags((size_t)T);
        CHK(d2h_sync(h, flags.data(), h->rowflag.p, (size_t)T * sizeof(int), s));
        std::vector<int> rows;
        for (int64_t t = 0; t < T; ++t) if (flags[(size_t)t] || (status & 4)) rows.push_back((int)t);
        if (status & 4) h->pool_overflows += 1;
        if (status & 2) h->tie_overflow = 1;
        for (size_t r0 = 0; r0 < rows.size(); r0 += 64) {    // 64 rows (x Nalloc doubles of scratch) at a time
            const int n = (int)((rows.size() - r0 < 64) ? rows.size() - r0 : 64);
            CHK(h->exact_rows.ensure((size_t)64 * sizeof(int)));
            CHK(h->exact_scratch.ensure((size_t)64 * h->Nalloc * sizeof(double)));
            CHK(h2d(h, h->exact_rows.p, rows.data() + r0, (size_t)n * sizeof(int), s));
            launch_knn_exact_rows(h->Fw.as<double>(), h->Dpad, h->Dt, h->N, h->Qp.as<double>(), h->exact_rows.as<int>(), n, K,
                                  h->exact_scratch.as<double>(), h->Nalloc, uc, qclass_dev, h->shard_offset,
                                  cand_dev, dist_dev, d2_dev, s);
            HIPCHK(hipStreamSynchronize(s));
        }
        HIPCHK(hipGetLastError());
        h->exact_row_fallbacks += (int)rows.size();
        return 0;
    }
    return fail("K-NN: internal error (attempt loop fell through)");
}

static int check_ready(snk_engine *h, bool need_target, bool need_join)
{
    if (!h) return fail("null handle");
    if (need_target && !h->have_db) return fail("no unit database uploaded (snk_upload_db)");
    if (need_join && !h->have_join) return fail("no join_contexts uploaded");
    if (!h->have_weights) return fail("weights not set (snk_set_weights)");
    return 0;
}

// state changes (database, weights, classes) while a submitted batch is still in flight would be seen by it
static int no_batch_in_flight(snk_engine *h, const char *who)
{
    if (h && (h->bslot[0].busy || h->bslot[1].busy))
        return fail("%s: a submitted batch is still in flight (snk_knn_viterbi_batch_collect it first)", who);
    if (h && (h->sticket[0].busy || h->sticket[1].busy))
        return fail("%s: a submitted sharded step is still in flight (snk_sharded_knn_viterbi_batch_collect it first)", who);
    return 0;
}

static int upload_queries(snk_engine *h, const double *Q, int64_t T, int D)
{
    if (!Q) return fail("null query matrix");
    if (D != h->Dt) return fail("query matrix has %d columns, database has %d", D, h->Dt);
    if (T < 1) return fail("query matrix has no rows");
    CHK(h->Qraw.ensure((size_t)T * D * sizeof(double)));
    StageTimer t(h, h->stream, TM_H2D);
    CHK(h2d(h, h->Qraw.p, Q, (size_t)T * D * sizeof(double), h->stream));
    if (!h->tsel.empty()) launch_mask_columns(h->Qraw.as<double>(), T, D, h->tmask.as<double>(), h->stream);
    return 0;
}

static int slot_ensure(snk_engine *h, UttSlot &s, int64_t T, int K)
{
    CHK(s.cand.ensure((size_t)T * K * sizeof(int64_t)));
    CHK(s.tdist.ensure((size_t)T * K * sizeof(double)));
    if (!use_sparse_viterbi(h, K, 1)) CHK(s.J.ensure((size_t)(T > 1 ? T - 1 : 1) * K * K * sizeof(double)));
    CHK(s.bp.ensure((size_t)T * K));
    CHK(s.path.ensure((size_t)T * sizeof(int64_t)));
    CHK(s.plen.ensure(sizeof(int64_t)));
    CHK(s.cost.ensure(sizeof(double)));
    return 0;
}

int snk_knn(snk_handle h, const double *Q, int64_t T, int D, int K, int64_t *cand_out, double *dist_out)
{
    CHK(check_ready(h, true, false));
    CHK(no_batch_in_flight(h, "snk_knn"));
    HIPCHK(hipSetDevice(h->device));
    if (!cand_out || !dist_out) return fail("snk_knn: null output");
    CHK(upload_queries(h, Q, T, D));
    UttSlot &s = h->slot[0];
    CHK(s.cand.ensure((size_t)T * K * sizeof(int64_t)));
    CHK(s.tdist.ensure((size_t)T * K * sizeof(double)));
    {
        const int rc = knn_device(h, h->Qraw.as<double>(), T, K, nullptr, s.cand.as<int64_t>(), s.tdist.as<double>(), nullptr);
        if (rc) { (void)hipStreamSynchronize(h->stream); collect_timers(h); return rc; }
    }
    {
        StageTimer t(h, h->stream, TM_D2H);
        D2HPart parts[2] = {{cand_out, s.cand.p, (size_t)T * K * sizeof(int64_t)},
                            {dist_out, s.tdist.p, (size_t)T * K * sizeof(double)}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    collect_timers(h);
    return 0;
}

// Diagnostic of the prefilter's error bound (include/snk.h): the prefilter's minimum key of every (query row, slab of
// rows_per_slab consecutive units) pair and the bound eps[t] it is trusted to; the caller compares with float64 keys.
int snk_prefilter_minima(snk_handle h, const double *Q, int64_t T, int D, float *slab_min, int64_t slab_min_len,
                         double *eps_out, int64_t *n_slabs_out, int *rows_per_slab_out)
{
    CHK(check_ready(h, true, false));
    CHK(no_batch_in_flight(h, "snk_prefilter_minima"));
    HIPCHK(hipSetDevice(h->device));
    if (!h->f16_ready) return fail("snk_prefilter_minima: this database shape has no float32 / bf16 prefilter");
    if (T < 1 || T > SNK_KNN_MAX_ROWS) return fail("snk_prefilter_minima: T outside 1..%d", (int)SNK_KNN_MAX_ROWS);
    const int64_t n_slabs = h->n_slabs16;
    if (n_slabs_out) *n_slabs_out = n_slabs;
    if (rows_per_slab_out) *rows_per_slab_out = 32 * h->nt16_eff;
    if (!slab_min) return 0;                   // size query
    if (slab_min_len < T * n_slabs || !eps_out) return fail("snk_prefilter_minima: output too small");
    CHK(upload_queries(h, Q, T, D));
    const int64_t Tpad = roundup(T, 32), G16 = 2 * n_slabs;
    const bool bf = h->bf16_ready && h->prefilter >= 1;
    const int dch16 = h->Dpad / 64;
    hipStream_t s = h->stream;
    CHK(h->Qp.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
    CHK(h->Qf.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
    CHK(h->qnorm.ensure((size_t)Tpad * sizeof(double)));
    CHK((bf ? h->b16l : h->b16h).ensure((size_t)(Tpad / 32) * 8 * 64 * 16 * dch16));
    CHK(h->eps16.ensure((size_t)Tpad * sizeof(double)));
    CHK(h->gmin32.ensure((size_t)Tpad * G16 * sizeof(float)));
    CHK(h->slabctr.ensure(4 * sizeof(unsigned int)));
    HIPCHK(hipMemsetAsync(h->slabctr.p, 0, 4 * sizeof(unsigned int), s));
    launch_prepare_queries(h->Qraw.as<double>(), T, h->Dt, h->Qp.as<double>(), h->Qf.as<double>(), h->qnorm.as<double>(),
                           Tpad, h->Dpad, s);
    const KnnPlan p0 = make_plan(h, 1);
    if (bf) {
        CHK(h->cq16.ensure((size_t)Tpad * sizeof(double)));
        launch_prepare_queries16b(h->Qp.as<double>(), h->qnorm.as<double>(), T, h->Dt, h->Dpad, h->fmax2.as<double>(),
                                  h->rho16.as<double>(), h->eps_c_bf, h->b16l.p, h->eps16.as<double>(), h->cq16.as<double>(), s);
        launch_knn_sweep16b(0, h->prefilter == 2 ? 4 : 3, h->nt16_eff, dch16, p0.grid_cus, h->a16l.p, h->b16l.p, nullptr, Tpad, n_slabs,
                            h->slabctr.as<unsigned int>(), h->gmin32.as<float>(), G16, nullptr, nullptr, nullptr, 0,
                            knn_pool_chunk_entries(), s);
    } else {
        launch_prepare_queries16(h->Qp.as<double>(), h->qnorm.as<double>(), T, h->Dt, h->Dpad, h->fmax2.as<double>(),
                                 h->eps_c, h->b16h.p, h->eps16.as<double>(), s);
        launch_knn_sweep16(0, h->nt16_eff, dch16, (h->Dt + 2) / 2, p0.grid_cus, h->a16h.p, h->b16h.p, nullptr, nullptr,
                           nullptr, Tpad, n_slabs, h->slabctr.as<unsigned int>(), h->gmin32.as<float>(), G16, nullptr,
                           nullptr, nullptr, 0, knn_pool_chunk_entries(), s);
    }
    HIPCHK(hipGetLastError());
    std::vector<float> g((size_t)T * G16);
    {
        D2HPart parts[2] = {{g.data(), h->gmin32.p, g.size() * sizeof(float)}, {eps_out, h->eps16.p, (size_t)T * sizeof(double)}};
        CHK(staged_d2h(h, s, parts, 2));
    }
    for (int64_t t = 0; t < T; ++t)
        for (int64_t w = 0; w < n_slabs; ++w) {
            const float a = g[t * G16 + 2 * w], b = g[t * G16 + 2 * w + 1];
            slab_min[t * n_slabs + w] = a < b ? a : b;
        }
    return 0;
}

int snk_knn_by_class(snk_handle h, const double *Q, int64_t T, int D, int K, const int32_t *query_class,
                     int64_t *cand_out, double *dist_out)
{
    CHK(check_ready(h, true, false));
    CHK(no_batch_in_flight(h, "snk_knn_by_class"));
    HIPCHK(hipSetDevice(h->device));
    if (!h->have_classes) return fail("snk_knn_by_class: unit classes not set (snk_set_unit_classes)");
    if (!query_class || !cand_out || !dist_out) return fail("snk_knn_by_class: null argument");
    CHK(upload_queries(h, Q, T, D));
    const int64_t Tpad = roundup(T, 32);
    CHK(h->qclass.ensure((size_t)Tpad * sizeof(int32_t)));
    HIPCHK(hipMemsetAsync(h->qclass.p, 0xfe, (size_t)Tpad * sizeof(int32_t), h->stream));
    CHK(h2d(h, h->qclass.p, query_class, (size_t)T * sizeof(int32_t), h->stream));
    UttSlot &s = h->slot[0];
    CHK(s.cand.ensure((size_t)T * K * sizeof(int64_t)));
    CHK(s.tdist.ensure((size_t)T * K * sizeof(double)));
    CHK(knn_device(h, h->Qraw.as<double>(), T, K, h->qclass.as<int32_t>(), s.cand.as<int64_t>(), s.tdist.as<double>(), nullptr));
    {
        D2HPart parts[2] = {{cand_out, s.cand.p, (size_t)T * K * sizeof(int64_t)}, {dist_out, s.tdist.p, (size_t)T * K * sizeof(double)}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    collect_timers(h);
    return 0;
}

int snk_candidate_distances(snk_handle h, const double *Q, int64_t T, int D, const int64_t *cand, int K,
                            double *dist_out)
{
    CHK(check_ready(h, true, false));
    HIPCHK(hipSetDevice(h->device));
    if (!cand || !dist_out || K < 1) return fail("snk_candidate_distances: null/empty argument");
    CHK(upload_queries(h, Q, T, D));
    const int64_t Tpad = roundup(T, 16);
    CHK(h->Qp.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
    CHK(h->Qf.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
    CHK(h->qnorm.ensure((size_t)Tpad * sizeof(double)));
    UttSlot &s = h->slot[0];
    CHK(s.cand.ensure((size_t)T * K * sizeof(int64_t)));
    CHK(s.tdist.ensure((size_t)T * K * sizeof(double)));
    launch_prepare_queries(h->Qraw.as<double>(), T, h->Dt, h->Qp.as<double>(), h->Qf.as<double>(), h->qnorm.as<double>(),
                           Tpad, h->Dpad, h->stream);
    CHK(h2d(h, s.cand.p, cand, (size_t)T * K * sizeof(int64_t), h->stream));
    launch_candidate_dist(h->Fw.as<double>(), h->Dpad, h->Dt, h->N, h->Qp.as<double>(), s.cand.as<int64_t>(), T, K,
                          s.tdist.as<double>(), h->stream);
    HIPCHK(hipGetLastError());
    D2HPart parts[1] = {{dist_out, s.tdist.p, (size_t)T * K * sizeof(double)}};
    CHK(staged_d2h(h, h->stream, parts, 1));
    collect_timers(h);
    return 0;
}

static int64_t join_units(snk_engine *h)
{
    // data_frames of unit_end_data = rows of join_contexts - 1 (synth_halfphone.py:3227)
    return h->Njc - 1;
}

// viterbi_mode 2 (default): the sparse path wherever it is supported -- batches (its first pass runs over the whole
// chip while the per-utterance passes hide beside the next group's K-NN) and, since pass 2 runs in chunks side by
// side and pass 4 on one wavefront, a single utterance too (T = 600, K = 100: 0.75 ms against 1.65 ms through the
// dense kernels).  1 forces it, 0 forces the dense exact path.  Same results.
static bool use_sparse_viterbi(const snk_engine *h, int K, int n_utts = 1)
{
    (void)n_utts;
    if (h->viterbi_weights == 1) return false;          // the float32 weight chain runs on the dense kernels
    if (!join_lb_supported(h->Dj, K)) return false;
    return h->viterbi_mode == 1 || h->viterbi_mode == 2;
}

static int sparse_ensure(snk_engine *h, UttSlot &s, int64_t rows, int K)
{
    CHK(s.Jlo.ensure((size_t)(rows > 1 ? rows - 1 : 1) * K * K * sizeof(float)));
    CHK(s.scale.ensure((size_t)rows * sizeof(float)));
    CHK(s.sets.ensure((size_t)rows * K * 16));
    CHK(s.cex.ensure((size_t)rows * K * join_record_bytes() + 4096));
    CHK(s.bp.ensure((size_t)rows * K));
    if (!h->vstats.p) {
        CHK(h->vstats.ensure((128 + 16 * 1024) * sizeof(unsigned long long)));    // + the stamps of a -DSNK_JF_TRACE build
        HIPCHK(hipMemset(h->vstats.p, 0, 128 * sizeof(unsigned long long)));
    }
    return 0;
}

static int ensure_jw32(snk_engine *h, hipStream_t st)
{
    if (h->jw32_ready) return 0;
    // once per set of weights; waited for: the groups of a batch launch pass 1 on different streams
    const int Jq = join_lb2_pitch(h->Dj);
    CHK(h->JW32.ensure((size_t)h->Njc * Jq * sizeof(float)));
    CHK(h->jw_umax.ensure(64));
    launch_join_weight32(h->JC_unw.as<float>(), h->Jp, h->Njc, h->Dj, h->wj.as<double>(), h->JW32.as<float>(), Jq,
                         h->jw_umax.as<unsigned int>(), st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    h->jw32_ready = true;
    return 0;
}

static void join_bounds_launch(snk_engine *h, const int64_t *cand, int64_t rows, int K, float *Jlo, float *scale, hipStream_t st)
{
    if (h->join_lb_variant == 1)
        launch_join_lb2(h->JW32.as<float>(), h->Dj, h->jw_umax.as<unsigned int>(), join_units(h), cand, rows, K, Jlo, scale, st);
    else
        launch_join_lb(h->JC_unw.as<float>(), h->Jp, h->Dj, h->wj.as<double>(), join_units(h), cand, rows, K, Jlo, scale, st);
}

// Passes 1..4 of joinfast_kernels.hip over `rows` candidate rows holding n_utts utterances (off: n_utts + 1
// row offsets).  Pass 1 runs on `main` (the whole chip, in parallel over the rows); the three per-utterance
// passes on `side` behind `knn_done` when the two streams differ.
static int viterbi_sparse_rows(snk_engine *h, UttSlot &s, const int64_t *cand, const double *tdist, int64_t rows,
                               const int64_t *off, int n_utts, int first_utt, int K, hipStream_t main, hipStream_t side,
                               int64_t *path, int64_t *plen, double *cost)
{
    const float *JC = h->JC_unw.as<float>();
    const double *wj = h->wj.as<double>();
    const bool lb_side = h->join_bounds_stream == 1 && side != main;
    if (h->join_lb_variant == 1) CHK(ensure_jw32(h, main));
    if (lb_side) {
        HIPCHK(hipEventRecord(s.knn_done, main));
        HIPCHK(hipStreamWaitEvent(side, s.knn_done, 0));
    }
    {
        StageTimer t(h, lb_side ? side : main, TM_JOIN_LB);
        join_bounds_launch(h, cand, rows, K, s.Jlo.as<float>(), s.scale.as<float>(), lb_side ? side : main);
    }
    if (side != main && !lb_side) {
        HIPCHK(hipEventRecord(s.knn_done, main));
        HIPCHK(hipStreamWaitEvent(side, s.knn_done, 0));
    }
    {
        StageTimer t(h, side, TM_DP_LB);
        launch_viterbi_lb(cand, tdist, s.Jlo.as<float>(), s.scale.as<float>(), off, n_utts, K, join_units(h),
                          (float)h->join_beta, s.sets.p, side,
                          n_utts <= h->lb_chunk_max_utts ? (n_utts <= 4 && h->lb_chunk > 32 ? 32 : h->lb_chunk) : 0, h->lb_warm);
    }
    {
        StageTimer t(h, side, TM_JOIN_SPARSE);
        launch_join_exact_sparse(JC, h->Jp, h->Dj, wj, join_units(h), cand, tdist, rows, K, s.sets.p, s.cex.p, side);
    }
    {
        StageTimer t(h, side, TM_DP_SPARSE);
        launch_viterbi_sparse(cand, s.cex.p, s.Jlo.as<float>(), JC, h->Jp, h->Dj, wj, off, n_utts, first_utt, K,
                              join_units(h), s.bp.as<unsigned char>(), path, plen, cost,
                              h->vstats.as<unsigned long long>(), side);
    }
    return 0;
}

static int viterbi_device(snk_engine *h, UttSlot &s, int64_t T, int K, hipStream_t st)
{
    if (K > 208) return fail("viterbi: n_candidates=%d > 208 not supported", K);
    if (use_sparse_viterbi(h, K, 1)) {
        CHK(sparse_ensure(h, s, T, K));
        const int64_t off[2] = {0, T};
        return viterbi_sparse_rows(h, s, s.cand.as<int64_t>(), s.tdist.as<double>(), T, off, 1, 0, K, st, st,
                                   s.path.as<int64_t>(), s.plen.as<int64_t>(), s.cost.as<double>());
    }
    {
        StageTimer t(h, st, TM_JOIN);
        launch_join_costs(h->JCw.as<double>(), h->Djpad, h->Dj, join_units(h), s.cand.as<int64_t>(), T, K,
                          s.J.as<double>(), st);
    }
    {
        StageTimer t(h, st, TM_VITERBI_DP);
        launch_viterbi_dp(s.cand.as<int64_t>(), s.tdist.as<double>(), s.J.as<double>(), T, K, join_units(h),
                          s.bp.as<unsigned char>(), s.path.as<int64_t>(), s.plen.as<int64_t>(),
                          s.cost.as<double>(), st, h->viterbi_weights == 1);
    }
    return 0;
}

int snk_join_costs(snk_handle h, const int64_t *cand, int64_t T, int K, double *J_out)
{
    CHK(check_ready(h, false, true));
    CHK(no_batch_in_flight(h, "snk_join_costs"));
    HIPCHK(hipSetDevice(h->device));
    if (!cand || !J_out) return fail("snk_join_costs: null argument");
    if (T < 2) return fail("snk_join_costs: need at least 2 columns");
    if (K < 1 || K > 208) return fail("snk_join_costs: K=%d outside 1..208", K);
    UttSlot &s = h->slot[0];
    CHK(slot_ensure(h, s, T, K));
    CHK(s.J.ensure((size_t)(T - 1) * K * K * sizeof(double)));
    CHK(h2d(h, s.cand.p, cand, (size_t)T * K * sizeof(int64_t), h->stream));
    {
        StageTimer t(h, h->stream, TM_JOIN);
        launch_join_costs(h->JCw.as<double>(), h->Djpad, h->Dj, join_units(h), s.cand.as<int64_t>(), T, K,
                          s.J.as<double>(), h->stream);
    }
    HIPCHK(hipGetLastError());
    CHK(d2h_sync(h, J_out, s.J.p, (size_t)(T - 1) * K * K * sizeof(double), h->stream));
    collect_timers(h);
    return 0;
}

int snk_join_bounds(snk_handle h, const int64_t *cand, int64_t T, int K, float *lo_out, float *scale_out)
{
    CHK(check_ready(h, false, true));
    CHK(no_batch_in_flight(h, "snk_join_bounds"));
    HIPCHK(hipSetDevice(h->device));
    if (!cand || !lo_out || !scale_out) return fail("snk_join_bounds: null argument");
    if (T < 2) return fail("snk_join_bounds: need at least 2 columns");
    if (!join_lb_supported(h->Dj, K)) return fail("snk_join_bounds: no bounds variant for %d join columns, K=%d", h->Dj, K);
    UttSlot &s = h->slot[0];
    CHK(slot_ensure(h, s, T, K));
    CHK(sparse_ensure(h, s, T, K));
    CHK(h2d(h, s.cand.p, cand, (size_t)T * K * sizeof(int64_t), h->stream));
    if (h->join_lb_variant == 1) CHK(ensure_jw32(h, h->stream));
    join_bounds_launch(h, s.cand.as<int64_t>(), T, K, s.Jlo.as<float>(), s.scale.as<float>(), h->stream);
    HIPCHK(hipGetLastError());
    D2HPart parts[2] = {{lo_out, s.Jlo.p, (size_t)(T - 1) * K * K * sizeof(float)}, {scale_out, s.scale.p, (size_t)(T - 1) * sizeof(float)}};
    CHK(staged_d2h(h, h->stream, parts, 2));
    collect_timers(h);
    return 0;
}

int snk_viterbi(snk_handle h, const int64_t *cand, const double *tdist, int64_t T, int K,
                int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    CHK(check_ready(h, false, true));
    CHK(no_batch_in_flight(h, "snk_viterbi"));
    HIPCHK(hipSetDevice(h->device));
    if (!cand || !tdist || !path_out || !path_len_out) return fail("snk_viterbi: null argument");
    if (T < 1 || K < 1) return fail("snk_viterbi: empty trellis");
    UttSlot &s = h->slot[0];
    CHK(slot_ensure(h, s, T, K));
    CHK(h2d(h, s.cand.p, cand, (size_t)T * K * sizeof(int64_t), h->stream));
    CHK(h2d(h, s.tdist.p, tdist, (size_t)T * K * sizeof(double), h->stream));
    CHK(viterbi_device(h, s, T, K, h->stream));
    HIPCHK(hipGetLastError());
    double cost = 0;
    {
        D2HPart parts[3] = {{path_len_out, s.plen.p, sizeof(int64_t)}, {&cost, s.cost.p, sizeof(double)},
                            {path_out, s.path.p, (size_t)T * sizeof(int64_t)}};
        CHK(staged_d2h(h, h->stream, parts, 3));
    }
    if (cost_out) *cost_out = cost;
    collect_timers(h);
    return 0;
}

int snk_knn_viterbi(snk_handle h, const double *Q, int64_t T, int D, int K, int64_t *cand_out,
                    double *dist_out, int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    CHK(check_ready(h, true, true));
    CHK(no_batch_in_flight(h, "snk_knn_viterbi"));
    HIPCHK(hipSetDevice(h->device));
    if (!path_out || !path_len_out) return fail("snk_knn_viterbi: null output");
    if (h->Njc != h->N + 1) return fail("snk_knn_viterbi: join_contexts rows (%lld) != N+1", (long long)h->Njc);
    CHK(upload_queries(h, Q, T, D));
    UttSlot &s = h->slot[0];
    CHK(slot_ensure(h, s, T, K));
    CHK(knn_device(h, h->Qraw.as<double>(), T, K, nullptr, s.cand.as<int64_t>(), s.tdist.as<double>(), nullptr));
    CHK(viterbi_device(h, s, T, K, h->stream));
    HIPCHK(hipGetLastError());
    double cost = 0;
    {
        StageTimer t(h, h->stream, TM_D2H);
        D2HPart parts[5] = {{cand_out, s.cand.p, cand_out ? (size_t)T * K * sizeof(int64_t) : 0},
                            {dist_out, s.tdist.p, dist_out ? (size_t)T * K * sizeof(double) : 0},
                            {path_len_out, s.plen.p, sizeof(int64_t)},
                            {&cost, s.cost.p, sizeof(double)},
                            {path_out, s.path.p, (size_t)T * sizeof(int64_t)}};
        CHK(staged_d2h(h, h->stream, parts, 5));
    }
    if (cost_out) *cost_out = cost;
    collect_timers(h);
    return 0;
}

// Groups consecutive utterances into K-NN calls of about h->batch_rows rows: the search is per row,
// so one sweep over the database serves every utterance of the group and the per-call stages
// (sample minima, thresholds, bucket, re-rank) amortise.  first[g] .. first[g+1] are the utterances
// of group g.
static std::vector<int> group_utterances(const snk_engine *h, const int64_t *row_offsets, int n_utts)
{
    std::vector<int> first(1, 0);
    // as few groups as batch_rows allows, of equal size and an even number of them: two groups of 16 utterances take a
    // B* step 13 % less time than 13 + 13 + 6 (11.9 against 13.7 ms; three of 11 / 11 / 10: 13.2, four of 8: 12.0,
    // one of 32: 18.2 -- nothing of its own step to run beside)
    const int64_t total = row_offsets[n_utts] - row_offsets[0];
    int64_t target = h->batch_rows;
    if (h->batch_rows > 0) {
        int64_t n_groups = (total + h->batch_rows - 1) / h->batch_rows;
        if (n_groups > 1 && (n_groups & 1)) ++n_groups;       // groups alternate between two workspaces and side streams
        target = (total + n_groups - 1) / n_groups;
    }
    int64_t rows = 0;
    for (int u = 0; u < n_utts; ++u) {
        const int64_t T = row_offsets[u + 1] - row_offsets[u];
        // close the group when adding this utterance would overshoot the even share by more than it undershoots
        if (u > first.back() && (h->batch_rows <= 0 || rows + T > h->batch_rows || rows + T - target > target - rows)) {
            first.push_back(u);
            rows = 0;
        }
        rows += T;
    }
    first.push_back(n_utts);
    return first;
}

// Join costs and recursions of the utterances [u0, u1) whose candidate rows are resident in
// cand_all / tdist_all (batch row numbering).  The join costs of the whole group are ONE launch on
// the main stream (its rows form one long sequence; the slab between two utterances is never read),
// the recursions ONE launch (a workgroup per utterance) on the side stream of the group's parity,
// where they land on the compute units the persistent K-NN sweep of the next group leaves free.
static int viterbi_group(snk_engine *h, int g, const int64_t *row_offsets, int u0, int u1, int K,
                         const int64_t *cand_all, const double *tdist_all, bool side_stream,
                         int64_t *res_path = nullptr, int64_t *res_plen = nullptr, double *res_cost = nullptr,
                         int n_batch_utts = 2)
{
    if (!res_path) { res_path = h->res_path.as<int64_t>(); res_plen = h->res_plen.as<int64_t>(); res_cost = h->res_cost.as<double>(); }
    const int64_t r0 = row_offsets[u0], rows = row_offsets[u1] - r0;
    UttSlot &s = h->slot[g & 1];
    hipStream_t dps = side_stream ? h->dp_stream[g & 1] : h->stream;
    // workspace reuse: the join costs of this group overwrite what the last recursion queued on this
    // workspace reads (an earlier group of this batch, or the tail of the batch submitted before)
    const bool sparse = use_sparse_viterbi(h, K, n_batch_utts);
    // (the sparse path with its bounds on the side stream touches the workspace on that stream only: in order)
    if (s.vit_recorded && !(sparse && side_stream && h->join_bounds_stream == 1)) HIPCHK(hipStreamWaitEvent(h->stream, s.vit_done, 0));
    if (sparse) {
        CHK(sparse_ensure(h, s, rows, K));
        std::vector<int64_t> off((size_t)(u1 - u0) + 1);
        for (int u = u0; u <= u1; ++u) off[(size_t)(u - u0)] = row_offsets[u] - r0;
        CHK(viterbi_sparse_rows(h, s, cand_all + r0 * K, tdist_all + r0 * K, rows, off.data(), u1 - u0, u0, K, h->stream, dps,
                                res_path + r0, res_plen, res_cost));
        if (side_stream) { HIPCHK(hipEventRecord(s.vit_done, dps)); s.vit_recorded = true; }
        else s.vit_recorded = false;
        return 0;
    }
    CHK(s.J.ensure((size_t)(rows > 1 ? rows - 1 : 1) * K * K * sizeof(double)));
    CHK(s.bp.ensure((size_t)rows * K));
    {
        StageTimer t(h, h->stream, TM_JOIN);
        launch_join_costs(h->JCw.as<double>(), h->Djpad, h->Dj, join_units(h), cand_all + r0 * K, rows, K,
                          s.J.as<double>(), h->stream);
    }
    if (side_stream) {
        HIPCHK(hipEventRecord(s.knn_done, h->stream));
        HIPCHK(hipStreamWaitEvent(dps, s.knn_done, 0));
    }
    std::vector<int64_t> off((size_t)(u1 - u0) + 1);
    for (int u = u0; u <= u1; ++u) off[(size_t)(u - u0)] = row_offsets[u] - r0;
    {
        StageTimer t(h, dps, TM_VITERBI_DP);
        launch_viterbi_dp_batch(cand_all + r0 * K, tdist_all + r0 * K, s.J.as<double>(), off.data(), u1 - u0, u0, K,
                                join_units(h), s.bp.as<unsigned char>(), res_path + r0, res_plen, res_cost, dps,
                                h->viterbi_weights == 1);
    }
    if (side_stream) { HIPCHK(hipEventRecord(s.vit_done, dps)); s.vit_recorded = true; }
    else s.vit_recorded = false;             // ran on the main stream: ordered with everything that follows
    return 0;
}

// Page-lock a caller buffer that is uploaded again and again (the query rows of a tune set): a copy
// from pageable memory makes the host wait for the stream, a copy from registered memory is queued.
int snk_host_register(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return fail("snk_host_register: null/empty buffer");
    HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return 0;
}

int snk_host_unregister(void *ptr)
{
    if (!ptr) return fail("snk_host_unregister: null buffer");
    HIPCHK(hipHostUnregister(ptr));
    return 0;
}

// Batch pipeline.  The main stream runs the K-NN of a group of utterances and their join costs; the
// T-step recursions run on a side stream, overlapping the K-NN of the next group.  All results stay
// on the device until the end of the batch; a copy stream moves them to pinned host memory behind
// the last recursions.  submit() only queues work (two batches may be in flight, each with its own
// query / candidate / result buffers), collect() waits for one batch: a caller that submits batch
// i+1 before collecting batch i hides the tail of batch i (its last group's recursions, the copy
// and the host-side hand-over) behind the K-NN of batch i+1.
int snk_knn_viterbi_batch_submit(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D,
                                 int K, int *ticket_out)
{
    CHK(check_ready(h, true, true));
    if (h->sticket[0].busy || h->sticket[1].busy)
        return fail("snk_knn_viterbi_batch_submit: a submitted sharded step is still in flight (snk_sharded_knn_viterbi_batch_collect it first)");
    HIPCHK(hipSetDevice(h->device));
    if (!row_offsets || n_utts < 1 || !ticket_out)
        return fail("snk_knn_viterbi_batch_submit: null/empty argument");
    if (D != h->Dt) return fail("query matrix has %d columns, database has %d", D, h->Dt);
    if (h->Njc != h->N + 1) return fail("snk_knn_viterbi_batch: join_contexts rows != N+1");
    if (K > 208) return fail("viterbi: n_candidates=%d > 208 not supported", K);
    if (row_offsets[0] != 0) return fail("snk_knn_viterbi_batch: row_offsets[0] must be 0 (got %lld)", (long long)row_offsets[0]);
    const int64_t total = row_offsets[n_utts];
    for (int u = 0; u < n_utts; ++u)
        if (row_offsets[u + 1] - row_offsets[u] < 1) return fail("snk_knn_viterbi_batch: utterance %d has no rows", u);
    const int slot = h->bslot[h->bnext].busy ? (h->bnext ^ 1) : h->bnext;
    BatchSlot &b = h->bslot[slot];
    if (b.busy) return fail("snk_knn_viterbi_batch_submit: two batches are in flight already (collect one first)");
    b.first = group_utterances(h, row_offsets, n_utts);
    b.n_groups = (int)b.first.size() - 1;
    b.n_utts = n_utts; b.K = K; b.D = D; b.total = total;
    b.offs.assign(row_offsets, row_offsets + n_utts + 1);
    { void *before = b.Qall.p; CHK(b.Qall.ensure((size_t)total * D * sizeof(double))); if (b.Qall.p != before) b.q_rows = -1; }
    CHK(b.cand.ensure((size_t)total * K * sizeof(int64_t)));
    CHK(b.dist.ensure((size_t)total * K * sizeof(double)));
    CHK(b.path.ensure((size_t)total * sizeof(int64_t)));
    CHK(b.plen.ensure((size_t)n_utts * sizeof(int64_t)));
    CHK(b.cost.ensure((size_t)n_utts * sizeof(double)));
    CHK(b.status.ensure((size_t)2 * b.n_groups * sizeof(int)));          // per group: K-NN status word | tile pairs the ball pass listed
    const size_t sz_path = ((size_t)total * sizeof(int64_t) + 63) & ~(size_t)63;
    const size_t sz_u = ((size_t)n_utts * 8 + 63) & ~(size_t)63, sz_st = ((size_t)2 * b.n_groups * sizeof(int) + 63) & ~(size_t)63;
    CHK(b.stage.ensure(sz_path + 2 * sz_u + sz_st));
    if (Q) {
        StageTimer t(h, h->stream, TM_H2D);
        CHK(h2d_via(b.qstage, b.Qall.p, Q, (size_t)total * D * sizeof(double), h->stream));
        if (!h->tsel.empty()) launch_mask_columns(b.Qall.as<double>(), total, D, h->tmask.as<double>(), h->stream);
        b.q_rows = total; b.q_D = D;
        b.q_offs.assign(row_offsets, row_offsets + n_utts + 1);
    } else if (b.q_rows != total || b.q_D != D || b.q_offs.size() != (size_t)n_utts + 1 ||
               !std::equal(b.q_offs.begin(), b.q_offs.end(), row_offsets)) {
        return fail("snk_knn_viterbi_batch_submit: no query matrix given and this workspace holds no rows of that shape "
                    "(the first submit on each of the two workspaces must carry Q)");
    }
    for (int g = 0; g < b.n_groups; ++g) {
        const int64_t r0 = row_offsets[b.first[g]], rows = row_offsets[b.first[g + 1]] - r0;
        CHK(knn_device(h, b.Qall.as<double>() + r0 * D, rows, K, nullptr, b.cand.as<int64_t>() + r0 * K,
                       b.dist.as<double>() + r0 * K, nullptr, b.status.as<int>() + g, nullptr, nullptr, false, false,
                       reinterpret_cast<unsigned int *>(b.status.as<int>() + b.n_groups + g)));
        b.ball_limit = h->ball_pass_ran ? h->ball_limit : -1.0;
        b.coarse_limit = h->coarse_pass_ran ? h->coarse_limit : -1.0;
        CHK(viterbi_group(h, g, row_offsets, b.first[g], b.first[g + 1], K, b.cand.as<int64_t>(), b.dist.as<double>(), true,
                          b.path.as<int64_t>(), b.plen.as<int64_t>(), b.cost.as<double>(), n_utts));
    }
    // results -> pinned memory, behind the K-NN status words (main stream) and the last recursions
    HIPCHK(hipEventRecord(h->knn_all_done, h->stream));
    HIPCHK(hipStreamWaitEvent(h->copy_stream, h->knn_all_done, 0));
    for (int i = 0; i < 2; ++i)
        if (h->slot[i].vit_recorded) HIPCHK(hipStreamWaitEvent(h->copy_stream, h->slot[i].vit_done, 0));
    {
        StageTimer t(h, h->copy_stream, TM_D2H);
        char *st = (char *)b.stage.p;
        HIPCHK(hipMemcpyAsync(st, b.path.p, (size_t)total * sizeof(int64_t), hipMemcpyDeviceToHost, h->copy_stream));
        HIPCHK(hipMemcpyAsync(st + sz_path, b.plen.p, (size_t)n_utts * sizeof(int64_t), hipMemcpyDeviceToHost, h->copy_stream));
        HIPCHK(hipMemcpyAsync(st + sz_path + sz_u, b.cost.p, (size_t)n_utts * sizeof(double), hipMemcpyDeviceToHost, h->copy_stream));
        HIPCHK(hipMemcpyAsync(st + sz_path + 2 * sz_u, b.status.p, (size_t)2 * b.n_groups * sizeof(int), hipMemcpyDeviceToHost, h->copy_stream));
    }
    HIPCHK(hipEventRecord(b.done, h->copy_stream));
    HIPCHK(hipGetLastError());
    b.busy = true;
    h->bnext = slot ^ 1;
    *ticket_out = slot;
    return 0;
}

int snk_knn_viterbi_batch_collect(snk_handle h, int ticket, int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (ticket < 0 || ticket > 1 || !h->bslot[ticket].busy) return fail("snk_knn_viterbi_batch_collect: no batch behind ticket %d", ticket);
    if (!path_out || !path_len_out || !cost_out) return fail("snk_knn_viterbi_batch_collect: null output");
    BatchSlot &b = h->bslot[ticket];
    const size_t sz_path = ((size_t)b.total * sizeof(int64_t) + 63) & ~(size_t)63;
    const size_t sz_u = ((size_t)b.n_utts * 8 + 63) & ~(size_t)63;
    HIPCHK(hipEventSynchronize(b.done));          // this batch only: the one submitted after it may still run
    HIPCHK(hipGetLastError());
    b.busy = false;
    char *st = (char *)b.stage.p;
    // deferred K-NN status words: redo the (rare) group whose sampled thresholds overflowed a list
    const int *status = reinterpret_cast<const int *>(st + sz_path + 2 * sz_u);
    if (b.ball_limit >= 0.0 && !h->filter_coarse)
        for (int g = 0; g < b.n_groups; ++g)
            if ((double)(unsigned int)status[b.n_groups + g] > b.ball_limit) { h->filter_coarse = true; h->ball_switches += 1; break; }
    if (b.coarse_limit >= 0.0 && !h->filter_onepass)
        for (int g = 0; g < b.n_groups; ++g)
            if ((double)(unsigned int)status[b.n_groups + g] > b.coarse_limit) { h->filter_onepass = true; h->onepass_switches += 1; break; }
    bool redone = false;
    for (int g = 0; g < b.n_groups; ++g) {
        if (status[g] == 0) continue;
        if (status[g] & 2) h->tie_overflow = 1;
        const int64_t r0 = b.offs[b.first[g]], rows = b.offs[b.first[g + 1]] - r0;
        const int saved = h->precision;
        h->precision = 0;
        const int rc = knn_device(h, b.Qall.as<double>() + r0 * b.D, rows, b.K, nullptr,
                                  b.cand.as<int64_t>() + r0 * b.K, b.dist.as<double>() + r0 * b.K, nullptr);
        h->precision = saved;
        if (rc) return rc;
        CHK(viterbi_group(h, g, b.offs.data(), b.first[g], b.first[g + 1], b.K, b.cand.as<int64_t>(), b.dist.as<double>(), false,
                          b.path.as<int64_t>(), b.plen.as<int64_t>(), b.cost.as<double>(), b.n_utts));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->batch_redos += 1;
        redone = true;
    }
    if (redone) {
        HIPCHK(hipMemcpyAsync(st, b.path.p, (size_t)b.total * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(st + sz_path, b.plen.p, (size_t)b.n_utts * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(st + sz_path + sz_u, b.cost.p, (size_t)b.n_utts * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    memcpy(path_out, st, (size_t)b.total * sizeof(int64_t));
    memcpy(path_len_out, st + sz_path, (size_t)b.n_utts * sizeof(int64_t));
    memcpy(cost_out, st + sz_path + sz_u, (size_t)b.n_utts * sizeof(double));
    collect_timers(h);
    return 0;
}

int snk_knn_viterbi_batch(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D,
                          int K, int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    if (!path_out || !path_len_out || !cost_out) return fail("snk_knn_viterbi_batch: null/empty argument");
    if (h && (h->bslot[0].busy || h->bslot[1].busy))
        return fail("snk_knn_viterbi_batch: a submitted batch is still in flight (collect it first)");
    int ticket = -1;
    CHK(snk_knn_viterbi_batch_submit(h, Q, row_offsets, n_utts, D, K, &ticket));
    return snk_knn_viterbi_batch_collect(h, ticket, path_out, path_len_out, cost_out);
}

// ---------------------------------------------------------------------------
// greedy search
// ---------------------------------------------------------------------------
int snk_set_greedy_layout(snk_handle h, int multiepoch, int last_frame_as_target, int join_split_mode)
{
    if (!h) return fail("null handle");
    if (!h->have_db || !h->have_join) return fail("snk_set_greedy_layout: upload the database first");
    if (multiepoch < 1 || multiepoch > 16) return fail("snk_set_greedy_layout: multiepoch=%d outside 1..16", multiepoch);
    if (join_split_mode != 0 && join_split_mode != 1) return fail("snk_set_greedy_layout: join_split_mode must be 0 or 1");
    if (join_split_mode == 1 && (h->Dj % 2)) return fail("snk_set_greedy_layout: join_split_mode 1 needs an even number of join columns");
    const int64_t Nrep = h->Njc - 1;            // rows of unit_start_data / unit_end_data
    if (Nrep != h->N) return fail("snk_set_greedy_layout: join_contexts rows (%lld) != N+1", (long long)h->Njc);
    if (h->N < multiepoch) return fail("snk_set_greedy_layout: database shorter than one multiepoch window");
    GreedyLayout g{};
    g.me = multiepoch;
    g.last_frame_as_target = last_frame_as_target ? 1 : 0;
    g.join_split_mode = join_split_mode;
    g.Nwin = h->N - multiepoch + 1;
    if (join_split_mode == 0) {
        // prev = unit_start_data = JC[:-1] (row i); current = unit_end_data = JC[1:], shifted by
        // the multiepoch overlap (synth_simple.py:194-195,213-214): window i -> JC row i + me
        g.jdim = h->Dj; g.prev_col0 = 0; g.cur_col0 = 0; g.prev_row0 = 0; g.cur_row0 = multiepoch;
    } else {
        // synth_halfphone.py:552-553: halves of the unit_start_data columns
        g.jdim = h->Dj / 2; g.prev_col0 = 0; g.cur_col0 = h->Dj / 2; g.prev_row0 = 0; g.cur_row0 = multiepoch - 1;
    }
    if (!greedy_supported(g, h->Dt))
        return fail("snk_set_greedy_layout: too many scan columns for the greedy step's table (join %d + %d epochs x %d)", g.jdim, multiepoch, h->Dt);
    h->glay = g;
    h->have_glay = true;
    h->gtiles_ready = false; h->gt16_ready = false;
    h->gh_ready = false; h->gj_ready = false;
    return 0;
}

// Up to greedy32_max_utts() utterances through the float32 persistent scan (greedy32_kernels.hip).  Returns in
// *undecided whether the launch stopped at a step it could not decide (the caller then runs the exact scan).
static int greedy32_group(snk_engine *h, int nu, const int64_t *q_off, const int64_t *ns, const int64_t *oo, const int64_t *st,
                          bool approx, bool want_dist, bool *undecided)
{
    const GreedyLayout &g = h->glay;
    // the target term of all steps as one matrix product per utterance; the scan then streams the join columns only
    G32Hoist hst{};
    bool hoist = h->greedy_hoist && greedy_hoist_supported(g, h->Dt);
    if (hoist) {
        const int64_t Wp = greedy_hoist_pitch(g);
        const int KA = greedy_hoist_k(g, h->Dt);
        int64_t rows = 0, prows = 0;
        for (int u = 0; u < nu; ++u) { rows += ns[u]; prows += greedy_hoist_rows(ns[u]); }
        const double w_bytes = (double)rows * (double)Wp * 4.0;
        if (w_bytes > h->greedy_hoist_max_gb * 1e9) hoist = false;
        if (hoist && w_bytes > (double)h->gh_W.bytes) {
            // the product must fit beside the voice: a device that cannot hold it keeps the scan that computes the target term itself
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || 1.125 * w_bytes - (double)h->gh_W.bytes > 0.9 * (double)free_b) hoist = false;      // (ensure() frees, then asks for 9/8)
        }
        if (hoist) {
            if (!h->gh_ready) {
                CHK(h->gh_nw.ensure((size_t)Wp * sizeof(double)));
                CHK(h->gh_max.ensure(64));
                launch_hoist_window_norms(g, h->fnorm.as<double>(), h->gh_nw.as<double>(), h->gh_max.as<unsigned long long>(), h->stream);
                HIPCHK(hipGetLastError());
                CHK(d2h_sync(h, &h->gh_fwmax2, h->gh_max.p, sizeof(double), h->stream));
                h->gh_ready = true;
            }
            // float16 join tiles: databases that are streamed from HBM (scans beyond 192 MB; or forced), up to three utterances
            // per scan.  Decided before the product: such a scan takes the target values from the bf16 pipe
            bool scan16 = false;
            if (h->greedy_f16 && nu <= 3) {
                CHK(h->gh_max.ensure(64));
                if (!h->gt16_ready) {
                    // once per database and layout: the float16 tiles (refused if a value leaves the float16 range)
                    CHK(h->gtiles16.ensure(greedy_tile16_bytes(g)));
                    unsigned int *mx = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(h->gh_max.p) + 32);
                    launch_greedy_tiles16(g, h->JC_unw.as<float>(), h->Jp, h->gtiles16.p, mx, h->stream);
                    HIPCHK(hipGetLastError());
                    float mabs = 0.f;
                    CHK(d2h_sync(h, &mabs, mx, sizeof(float), h->stream));
                    h->gt16_ok = mabs < 6.0e4f;
                    h->gt16_ready = true;
                }
                if (h->gt16_ok && !h->gj_ready) {
                    // once per set of weights: max ||w o S'[i]||^2 and ||w||^2
                    unsigned long long *o2 = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(h->gh_max.p) + 16);
                    launch_greedy_join_norms(g, h->gtiles.as<float>(), h->wj.as<double>(), o2, h->stream);
                    HIPCHK(hipGetLastError());
                    double v[2] = {0.0, 0.0};
                    CHK(d2h_sync(h, v, o2, sizeof(v), h->stream));
                    h->g16_delta = 4.8828125e-4 * sqrt(v[0]) + 2.98023223876953125e-8 * sqrt(v[1]);
                    h->gj_ready = true;
                }
                if (h->gt16_ok) {
                    hst.JT16 = h->gtiles16.p; hst.f16_delta = h->g16_delta; hst.f16_force = h->greedy_f16 == 2;
                    // (launch_greedy32 takes the float16 tiles under the same condition)
                    scan16 = hst.f16_force || (double)g.Nwin * (double)(g.jdim + 1) * 4.0 > (double)((size_t)192 << 20);
                    if (scan16) h->greedy_f16_launches += 1;
                }
            }
            const bool fast = scan16 && h->greedy_hoist_fast && greedy_hoist16_supported(g, h->Dt);
            CHK(h->gh_aq.ensure((size_t)prows * KA * sizeof(double)));
            CHK(h->gh_qn2.ensure((size_t)prows * sizeof(double)));
            CHK(h->gh_W.ensure((size_t)rows * (size_t)Wp * sizeof(float)));
            int64_t r0 = 0, p0 = 0;
            for (int u = 0; u < nu; ++u) {
                float *W = h->gh_W.as<float>() + (size_t)r0 * Wp;
                double *qn2 = h->gh_qn2.as<double>() + p0;
                (fast ? launch_hoist_product16 : launch_hoist_product)(g, h->F_unw.as<float>(), h->Fp, h->N, h->Dt, h->wt.as<double>(),
                                     h->Qraw.as<double>(), q_off[u], ns[u], h->gh_nw.as<double>(), h->gh_aq.as<double>() + (size_t)p0 * KA, qn2, W, h->stream);
                hst.W[u] = W; hst.qn2[u] = qn2;
                r0 += ns[u]; p0 += greedy_hoist_rows(ns[u]);
            }
            HIPCHK(hipGetLastError());
            hst.Wp = Wp; hst.c = fast ? greedy_hoist_c16(g, h->Dt) : greedy_hoist_c(g, h->Dt); hst.fwmax2 = h->gh_fwmax2;
            if (fast) h->greedy_hoist16_launches += 1;
            h->greedy_hoist_launches += 1;
        }
    }
    if (!hoist && nu > greedy32_max_utts(false)) {
        // without the product a scan serves three utterances: two launches
        const int n1 = greedy32_max_utts(false);
        bool u1 = false, u2 = false;
        CHK(greedy32_group(h, n1, q_off, ns, oo, st, approx, want_dist, &u1));
        CHK(greedy32_group(h, nu - n1, q_off + n1, ns + n1, oo + n1, st + n1, approx, want_dist, &u2));
        *undecided = u1 || u2;
        return 0;
    }
    const int nblk = greedy32_blocks(g, h->Dt, h->n_cus, hoist);
    CHK(h->g32_blk.ensure(greedy32_block_bytes(nblk)));
    CHK(h->g32_ctl.ensure(256));
    unsigned int *gen = h->g32_ctl.as<unsigned int>();
    int64_t *status = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(h->g32_ctl.p) + 16);
    // one utterance against a database that fits the chip's LDS: resident scan, every workgroup decides for itself
    // (float16 tiles forced on by the caller -- tests -- are the streamed scan's)
    const bool resident = hoist && nu == 1 && h->greedy_resident && greedy_res_supported(g, h->Dt, h->n_cus) && !hst.f16_force;
    if (resident) {
        CHK(h->g32_res.ensure(greedy_res_record_bytes(g) + 256));
        launch_greedy_res(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                          h->wj.as<double>(), h->gtiles.as<float>(), h->Qraw.as<double>(), q_off[0], ns[0], oo[0], st[0],
                          (approx ? 1 : 0) | (h->greedy_test_stall ? 256 : 0) | (h->greedy_fenced ? 512 : 0), h->g32_res.p, status, h->gpath.as<int64_t>(), &hst, h->stream);
        h->greedy_resident_launches += 1;
    } else
    launch_greedy32(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                    h->wj.as<double>(), h->gtiles.as<float>(), h->Qraw.as<double>(), nu, q_off, ns, oo, st, (approx ? 1 : 0) | (h->greedy_test_stall ? 256 : 0) | (h->greedy_fenced ? 512 : 0) | (h->greedy_speculate ? 0 : 2048),
                    h->g32_blk.p, h->n_cus, gen,
                    status, h->gpath.as<int64_t>(), hoist ? &hst : nullptr, h->stream);
    HIPCHK(hipGetLastError());
    int64_t stv[8] = {0, 0, 0, 0, 0, 0, 0, 0};         // undecided step + 1 | second-phase rounds | windows decided by exact totals | watchdog | (resident scan: why)
    CHK(d2h_sync(h, stv, status, sizeof(stv), h->stream));
    greedy32_trace_dump();
    greedy_res_trace_dump();
    *undecided = stv[0] != 0;
    for (int i = 0; i < 8; ++i) h->greedy_last_status[i] = stv[i];
    h->greedy_second_rounds += stv[1];
    h->greedy_exact_windows += stv[2];
    h->greedy_stalls += stv[3];
    if (!*undecided && want_dist) {
        for (int u = 0; u < nu; ++u)
            launch_greedy32_dist(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                                 h->wj.as<double>(), h->Qraw.as<double>(), u, q_off[u], ns[u], oo[u], st[u], h->gpath.as<int64_t>(),
                                 h->gdist.as<double>(), h->stream);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// greedy_mode 2 (default): the float32 scan where it pays -- always with the hoisted target term (the scan streams the
// join columns only: 202 against 262 us per step at 1.5 M units, 22 against 28 at 65 536); without it, when several
// utterances share every pass over the database (three per scan); a single utterance then goes through the exact
// scan.  Same paths and distances either way.
static bool use_greedy32(const snk_engine *h, int n_utts)
{
    if (!greedy32_supported(h->glay, h->Dt)) return false;
    const bool hoist = h->greedy_hoist && greedy_hoist_supported(h->glay, h->Dt);
    return h->greedy_mode == 1 || (h->greedy_mode == 2 && (n_utts >= 2 || hoist));
}

int snk_greedy(snk_handle h, const double *Q, int64_t T, int D, int64_t start_state, double eps,
               int64_t *path_out, double *dist_out, int64_t *nsteps_out)
{
    CHK(check_ready(h, true, true));
    CHK(no_batch_in_flight(h, "snk_greedy"));
    HIPCHK(hipSetDevice(h->device));
    if (!h->have_glay) return fail("snk_greedy: greedy layout not set (snk_set_greedy_layout)");
    if (!path_out || !nsteps_out) return fail("snk_greedy: null output");
    if (!(eps >= 0.0)) return fail("snk_greedy: search_epsilon must be >= 0");
    const GreedyLayout &g = h->glay;
    if (start_state >= g.Nwin) return fail("snk_greedy: start_state %lld out of range", (long long)start_state);
    CHK(upload_queries(h, Q, T, D));
    const int64_t nsteps = T / g.me;          // py2 integer division: tail frames dropped
    *nsteps_out = nsteps;
    if (nsteps == 0) { HIPCHK(hipStreamSynchronize(h->stream)); collect_timers(h); return 0; }
    if (!h->gtiles_ready) {
        // the scan reads a lane-major copy of its columns (greedy_kernels.hip); built on first use
        CHK(h->gtiles.ensure(greedy_tile_bytes(g, h->Dt)));
        launch_greedy_tiles(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->JC_unw.as<float>(), h->Jp, h->gtiles.as<float>(), h->stream);
        HIPCHK(hipGetLastError());
        h->gtiles_ready = true;
    }
    const int nblk = greedy_blocks(g, h->Dt, h->n_cus);
    CHK(h->gprev.ensure(2 * greedy_table_doubles(g, h->Dt) * sizeof(double) + 512));   // + slack: the scan warms whole 512-byte spans     // (weight, reference) tables
    CHK(h->gsync.ensure(greedy_counter_bytes()));
    CHK(h->gblkmin.ensure((size_t)nblk * sizeof(double)));
    CHK(h->gblkarg.ensure((size_t)nblk * sizeof(int64_t)));
    CHK(h->gpath.ensure((size_t)nsteps * sizeof(int64_t)));
    CHK(h->gdist.ensure((size_t)nsteps * sizeof(double)));
    bool exact_scan = true;
    if (use_greedy32(h, 1)) {
        // float32 prefilter scan, one persistent launch; search_epsilon >= 1e-3: the float32 minimum is the answer
        const int64_t zero = 0;
        bool undecided = false;
        {
            StageTimer t(h, h->stream, TM_GREEDY_STEPS);
            CHK(greedy32_group(h, 1, &zero, &nsteps, &zero, &start_state, eps >= 1e-3, dist_out != nullptr, &undecided));
        }
        exact_scan = undecided;
        if (undecided) h->greedy_fallbacks += 1;
    }
    if (exact_scan) {
        StageTimer t(h, h->stream, TM_GREEDY_STEPS);
        launch_greedy(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                      h->wj.as<double>(), h->gtiles.as<float>(), h->Qraw.as<double>(), nsteps, start_state, h->gprev.as<double>(),
                      h->gblkmin.as<double>(), h->gblkarg.as<int64_t>(), nblk, h->n_cus, h->gsync.as<unsigned int>(), h->gpath.as<int64_t>(),
                      h->gdist.as<double>(), h->stream);
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

// Several utterances through the greedy search together: up to three share every scan of the database
// (one weighted value per column and window, one comparison per utterance), so the database is read
// once per step for all of them.  Results equal snk_greedy utterance by utterance.
int snk_greedy_batch(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D,
                     const int64_t *start_states, double eps, int64_t *path_out, double *dist_out,
                     int64_t *nsteps_out)
{
    CHK(check_ready(h, true, true));
    CHK(no_batch_in_flight(h, "snk_greedy_batch"));
    HIPCHK(hipSetDevice(h->device));
    if (!h->have_glay) return fail("snk_greedy_batch: greedy layout not set (snk_set_greedy_layout)");
    if (!Q || !row_offsets || n_utts < 1 || !path_out || !nsteps_out) return fail("snk_greedy_batch: null/empty argument");
    if (D != h->Dt) return fail("query matrix has %d columns, database has %d", D, h->Dt);
    if (!(eps >= 0.0)) return fail("snk_greedy_batch: search_epsilon must be >= 0");
    const GreedyLayout &g = h->glay;
    const int64_t total = row_offsets[n_utts];
    std::vector<int64_t> nsteps((size_t)n_utts), out_off((size_t)n_utts + 1, 0);
    for (int u = 0; u < n_utts; ++u) {
        const int64_t T = row_offsets[u + 1] - row_offsets[u];
        if (T < 1) return fail("snk_greedy_batch: utterance %d has no rows", u);
        if (start_states && start_states[u] >= g.Nwin) return fail("snk_greedy_batch: start_state %lld out of range", (long long)start_states[u]);
        nsteps[(size_t)u] = T / g.me;             // py2 integer division: tail frames dropped
        nsteps_out[u] = nsteps[(size_t)u];
        out_off[(size_t)u + 1] = out_off[(size_t)u] + nsteps[(size_t)u];
    }
    const int64_t total_steps = out_off[(size_t)n_utts];
    CHK(upload_queries(h, Q, total, D));
    if (total_steps == 0) { HIPCHK(hipStreamSynchronize(h->stream)); collect_timers(h); return 0; }
    if (!h->gtiles_ready) {
        CHK(h->gtiles.ensure(greedy_tile_bytes(g, h->Dt)));
        launch_greedy_tiles(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->JC_unw.as<float>(), h->Jp, h->gtiles.as<float>(), h->stream);
        HIPCHK(hipGetLastError());
        h->gtiles_ready = true;
    }
    const int ub = greedy_max_utts(g, h->Dt);
    const int nblk = greedy_blocks(g, h->Dt, h->n_cus, ub);
    CHK(h->gprev.ensure(2 * greedy_table_doubles(g, h->Dt, ub) * sizeof(double) + 512));
    CHK(h->gsync.ensure(greedy_counter_bytes()));
    CHK(h->gblkmin.ensure((size_t)ub * nblk * sizeof(double)));
    CHK(h->gblkarg.ensure((size_t)ub * nblk * sizeof(int64_t)));
    CHK(h->gpath.ensure((size_t)total_steps * sizeof(int64_t)));
    CHK(h->gdist.ensure((size_t)total_steps * sizeof(double)));
    // utterances of similar length share a scan (the scan runs for the longest of its group)
    std::vector<int> order((size_t)n_utts);
    for (int u = 0; u < n_utts; ++u) order[(size_t)u] = u;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return nsteps[(size_t)x] > nsteps[(size_t)y]; });
    const bool g32 = use_greedy32(h, n_utts);
    const int ub_run = g32 ? greedy32_max_utts(h->greedy_hoist && greedy_hoist_supported(g, h->Dt)) : ub;
    {
        StageTimer t(h, h->stream, TM_GREEDY_STEPS);
        for (int i = 0; i < n_utts; i += ub_run) {
            int nu = 0;
            int64_t q_off[6], ns[6], oo[6], st[6];
            for (; nu < ub_run && i + nu < n_utts; ++nu) {
                const int u = order[(size_t)(i + nu)];
                if (nsteps[(size_t)u] == 0) break;             // sorted: the rest have no steps either
                q_off[nu] = row_offsets[u]; ns[nu] = nsteps[(size_t)u]; oo[nu] = out_off[(size_t)u];
                st[nu] = start_states ? start_states[u] : -1;
            }
            if (nu == 0) break;
            if (g32) {
                bool undecided = false;
                CHK(greedy32_group(h, nu, q_off, ns, oo, st, eps >= 1e-3, dist_out != nullptr, &undecided));
                if (!undecided) continue;
                h->greedy_fallbacks += 1;
            }
            // exact scan, a launch per step (two utterances per scan at most)
            for (int j = 0; j < nu; j += ub) {
                const int n2 = (nu - j < ub) ? nu - j : ub;
                launch_greedy_batch(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                                    h->wj.as<double>(), h->gtiles.as<float>(), h->Qraw.as<double>(), n2, q_off + j, ns + j, oo + j, st + j,
                                    h->gprev.as<double>(), h->gblkmin.as<double>(), h->gblkarg.as<int64_t>(),
                                    greedy_blocks(g, h->Dt, h->n_cus, n2), h->n_cus, h->gsync.as<unsigned int>(),
                                    h->gpath.as<int64_t>(), h->gdist.as<double>(), h->stream);
            }
        }
    }
    HIPCHK(hipGetLastError());
    {
        StageTimer t(h, h->stream, TM_D2H);
        D2HPart parts[2] = {{path_out, h->gpath.p, (size_t)total_steps * sizeof(int64_t)},
                            {dist_out, h->gdist.p, dist_out ? (size_t)total_steps * sizeof(double) : 0}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    collect_timers(h);
    return 0;
}

int snk_path_scores(snk_handle h, const double *Q, const int64_t *path, int64_t L, int mode,
                    double *tsq_out, double *jsq_out)
{
    CHK(check_ready(h, true, true));
    HIPCHK(hipSetDevice(h->device));
    if (!Q || !path || !tsq_out || L < 1) return fail("snk_path_scores: null/empty argument");
    if (mode != 0 && mode != 1) return fail("snk_path_scores: mode must be 0 (viterbi) or 1 (greedy)");
    if (mode == 1 && !h->have_glay) return fail("snk_path_scores: greedy layout not set");
    GreedyLayout g = h->glay;
    int me = 1, nep = 1, jcols = h->Dj;
    if (mode == 1) { me = g.me; nep = (g.last_frame_as_target && me > 1) ? 2 : me; jcols = g.jdim; }
    const int64_t limit = (mode == 1) ? g.Nwin : h->N;
    for (int64_t l = 0; l < L; ++l)
        if (path[l] < 0 || path[l] >= limit) return fail("snk_path_scores: path[%lld]=%lld out of range", (long long)l, (long long)path[l]);
    const size_t qrows = (size_t)L * me;
    CHK(h->Qraw.ensure(qrows * h->Dt * sizeof(double)));
    CHK(h->gpath.ensure((size_t)L * sizeof(int64_t)));
    const size_t tbytes = (size_t)L * nep * h->Dt * sizeof(double);
    const size_t jbytes = (size_t)(L > 1 ? L - 1 : 1) * jcols * sizeof(double);
    CHK(h->d2tmp.ensure(tbytes + jbytes));
    CHK(h2d(h, h->Qraw.p, Q, qrows * h->Dt * sizeof(double), h->stream));
    if (!h->tsel.empty()) launch_mask_columns(h->Qraw.as<double>(), (int64_t)qrows, h->Dt, h->tmask.as<double>(), h->stream);
    CHK(h2d(h, h->gpath.p, path, (size_t)L * sizeof(int64_t), h->stream));
    double *tsq = h->d2tmp.as<double>();
    double *jsq = reinterpret_cast<double *>(reinterpret_cast<char *>(h->d2tmp.p) + tbytes);
    launch_path_scores(g, mode, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                       h->wj.as<double>(), h->Qraw.as<double>(), h->gpath.as<int64_t>(), L, tsq, jsq, jcols, h->stream);
    HIPCHK(hipGetLastError());
    {
        D2HPart parts[2] = {{tsq_out, tsq, tbytes}, {jsq_out, jsq, (jsq_out && L > 1) ? (size_t)(L - 1) * jcols * sizeof(double) : 0}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    return 0;
}

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
    if ((h->ball_pass_ran || h->coarse_pass_ran) && h->cpairctl.p) {
        unsigned int listed = 0;                  // (of the last group's call: enough to judge the voice)
        CHK(d2h_sync(h, &listed, h->cpairctl.p, sizeof(listed), h->stream));
        note_ball_pairs(h, listed);
    }
    for (int g = 0; g < n_groups; ++g) {
        if (st[g] == 0) continue;
        if (st[g] & 2) h->tie_overflow = 1;
        const int64_t r0 = g * step, rows = (r0 + step <= total) ? step : total - r0;
        const int saved = h->precision;
        h->precision = 0;
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
        const std::vector<int> first = group_utterances(h, row_offsets, n_utts);
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

// Viterbi of a batch of utterances whose candidates the caller already has (label-driven preselection:
// preselect_units_quinphone / monophone_then_acoustic, synth_halfphone.py:1315-1396): what the tail of
// snk_knn_viterbi_batch does -- join bounds, sparse exact recursion per group on the side streams -- without the K-NN.
int snk_viterbi_batch(snk_handle h, const int64_t *cand, const double *tdist, const int64_t *row_offsets, int n_utts, int K,
                      int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    CHK(check_ready(h, false, true));
    CHK(no_batch_in_flight(h, "snk_viterbi_batch"));
    HIPCHK(hipSetDevice(h->device));
    if (!cand || !tdist || !row_offsets || n_utts < 1 || !path_out || !path_len_out || !cost_out)
        return fail("snk_viterbi_batch: null/empty argument");
    if (K < 1 || K > 208) return fail("viterbi: n_candidates=%d outside 1..208", K);
    if (h->Njc != h->N + 1 && h->N > 0) return fail("snk_viterbi_batch: join_contexts rows != N+1");
    const int64_t total = row_offsets[n_utts];
    for (int u = 0; u < n_utts; ++u)
        if (row_offsets[u + 1] - row_offsets[u] < 1) return fail("snk_viterbi_batch: utterance %d has no rows", u);
    CHK(h->mcand.ensure((size_t)total * K * sizeof(int64_t)));
    CHK(h->mdist.ensure((size_t)total * K * sizeof(double)));
    CHK(h->res_path.ensure((size_t)total * sizeof(int64_t)));
    CHK(h->res_plen.ensure((size_t)n_utts * sizeof(int64_t)));
    CHK(h->res_cost.ensure((size_t)n_utts * sizeof(double)));
    {
        StageTimer t(h, h->stream, TM_H2D);
        CHK(h2d(h, h->mcand.p, cand, (size_t)total * K * sizeof(int64_t), h->stream));
        CHK(h2d(h, h->mdist.p, tdist, (size_t)total * K * sizeof(double), h->stream));
    }
    {
        const std::vector<int> first = group_utterances(h, row_offsets, n_utts);
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

// the three collectives of the sharded search, on the engine's stream
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
        if ((h->ball_pass_ran || h->coarse_pass_ran) && h->cpairctl.p) {
            // (the host is waiting here anyway: what the step's last K-NN call listed decides whether this voice keeps its filter)
            unsigned int listed = 0;
            CHK(d2h_sync(h, &listed, h->cpairctl.p, sizeof(listed), h->stream));
            note_ball_pairs(h, listed);
        }
        int64_t so = 0, ro = 0;
        for (int p = 0; p < G; ++p) {
            const int64_t ts = totall[(size_t)me * G + p], tr = totall[(size_t)p * G + me];
            if (ts < 0 || ts > t.rows_to[(size_t)p] * K || tr < 0 || tr > r_own * K) return fail("sharded exchange: inconsistent list totals between ranks");
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
        const std::vector<int> first = group_utterances(h, t.own_off.data(), n_own);
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
        if (n_own > 0) {
            HIPCHK(hipMemcpyAsync(st, t.res_path.p, (size_t)r_own * sizeof(int64_t), hipMemcpyDeviceToHost, h->copy_stream));
            HIPCHK(hipMemcpyAsync(st + sz_path, t.res_plen.p, (size_t)n_own * sizeof(int64_t), hipMemcpyDeviceToHost, h->copy_stream));
            HIPCHK(hipMemcpyAsync(st + sz_path + sz_u, t.res_cost.p, (size_t)n_own * sizeof(double), hipMemcpyDeviceToHost, h->copy_stream));
        }
        if (n_status > 0)
            HIPCHK(hipMemcpyAsync(st + sz_path + 2 * sz_u, t.status.p, (size_t)n_status * sizeof(int), hipMemcpyDeviceToHost, h->copy_stream));
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
    const std::string msg = g_err;
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
    if (h->bslot[0].busy || h->bslot[1].busy)
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

// ---------------------------------------------------------------------------
// waveform-side gather
// ---------------------------------------------------------------------------
int snk_upload_frames(snk_handle h, const float *spec, const double *fzv, int64_t rows, int H)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!spec || !fzv || rows < 1 || H < 1) return fail("snk_upload_frames: null/empty argument");
    const size_t W = (size_t)3 * H;
    CHK(h->frames_spec.ensure((size_t)rows * W * sizeof(float)));
    CHK(h->frames_fzv.ensure((size_t)rows * 2 * sizeof(double)));
    CHK(h2d(h, h->frames_spec.p, spec, (size_t)rows * W * sizeof(float), h->stream));
    CHK(h2d(h, h->frames_fzv.p, fzv, (size_t)rows * 2 * sizeof(double), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->frames_rows = rows;
    h->frames_W = (int)W;
    return 0;
}

int snk_concat_fragments(snk_handle h, const int64_t *first_row, const int64_t *utt_lo, const int64_t *utt_hi,
                         int64_t n, int multiepoch, int overlap, const double *in_taper,
                         double *spec_out, double *fz_out)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->frames_rows < 1) return fail("snk_concat_fragments: no analysis frames uploaded (snk_upload_frames)");
    if (!first_row || !utt_lo || !utt_hi || !spec_out || !fz_out || n < 1) return fail("snk_concat_fragments: null/empty argument");
    if (multiepoch < 1) return fail("snk_concat_fragments: multiepoch < 1");
    if (overlap < 0 || (overlap % 2) != 0) return fail("snk_concat_fragments: frame overlap should be even number");
    if (overlap > 0 && !in_taper) return fail("snk_concat_fragments: null taper");
    if (overlap > 0 && 2 * overlap > multiepoch + overlap)
        return fail("snk_concat_fragments: taper_length (%d) too long for (padded) unit length (%d)", overlap, multiepoch + overlap);
    for (int64_t k = 0; k < n; ++k) {
        if (utt_lo[k] < 0 || utt_hi[k] > h->frames_rows || utt_lo[k] >= utt_hi[k] || first_row[k] < utt_lo[k] ||
            first_row[k] >= utt_hi[k])
            return fail("snk_concat_fragments: unit %lld lies outside its utterance / the uploaded frames", (long long)k);
        // the reference slices silently short (and asserts) when a window runs past its utterance without overlap
        if (overlap == 0 && first_row[k] + multiepoch > utt_hi[k])
            return fail("snk_concat_fragments: unit %lld runs past the end of its utterance (needs overlap > 0)", (long long)k);
    }
    const int64_t rows_out = n * multiepoch;
    const size_t W = (size_t)h->frames_W;
    const size_t in_bytes = (size_t)n * 3 * sizeof(int64_t) + (size_t)(overlap > 0 ? overlap : 1) * sizeof(double);
    CHK(h->cc_in.ensure(in_bytes));
    CHK(h->cc_out.ensure((size_t)rows_out * (W + 1) * sizeof(double)));
    int64_t *d_first = h->cc_in.as<int64_t>(), *d_lo = d_first + n, *d_hi = d_lo + n;
    double *d_taper = reinterpret_cast<double *>(d_hi + n);
    CHK(h2d(h, d_first, first_row, (size_t)n * sizeof(int64_t), h->stream));
    CHK(h2d(h, d_lo, utt_lo, (size_t)n * sizeof(int64_t), h->stream));
    CHK(h2d(h, d_hi, utt_hi, (size_t)n * sizeof(int64_t), h->stream));
    if (overlap > 0) CHK(h2d(h, d_taper, in_taper, (size_t)overlap * sizeof(double), h->stream));
    double *d_spec = h->cc_out.as<double>(), *d_fz = d_spec + (size_t)rows_out * W;
    launch_concat_fragments(h->frames_spec.as<float>(), (int)W, h->frames_fzv.as<double>(), d_first, d_lo, d_hi, n,
                            multiepoch, overlap, d_taper, d_spec, d_fz, h->stream);
    HIPCHK(hipGetLastError());
    {
        D2HPart parts[2] = {{spec_out, d_spec, (size_t)rows_out * W * sizeof(double)}, {fz_out, d_fz, (size_t)rows_out * sizeof(double)}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    return 0;
}

// ---------------------------------------------------------------------------
// timers / options / self test
// ---------------------------------------------------------------------------
int snk_timer_count(void) { return TM_COUNT; }
const char *snk_timer_name(int i) { return (i >= 0 && i < TM_COUNT) ? kTimerNames[i] : ""; }

int snk_get_timers(snk_handle h, double *ms_out, int capacity)
{
    if (!h || !ms_out) return -1;
    // layout: [total_ms x TM_COUNT][launch count x TM_COUNT]
    int n = 0;
    for (int i = 0; i < TM_COUNT && n < capacity; ++i) ms_out[n++] = h->tm_ms[i];
    for (int i = 0; i < TM_COUNT && n < capacity; ++i) ms_out[n++] = (double)h->tm_n[i];
    return n;
}

int snk_reset_timers(snk_handle h)
{
    if (!h) return fail("null handle");
    for (int i = 0; i < TM_COUNT; ++i) { h->tm_ms[i] = 0; h->tm_n[i] = 0; }
    if (h->margin_stat.p && !h->bslot[0].busy && !h->bslot[1].busy && !h->sticket[0].busy && !h->sticket[1].busy) {
        const unsigned int init[2] = {0u, 0x7f800000u};
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        CHK(h2d_sync(h, h->margin_stat.p, init, sizeof(init)));
    }
    return 0;
}

int snk_set_option(snk_handle h, const char *name, double value)
{
    if (!h || !name) return fail("snk_set_option: null argument");
    if (!strcmp(name, "list_capacity")) {
        if (value < 64 || value > 8192) return fail("list_capacity must be in 64..8192");
        h->cap = (int)value;
    } else if (!strcmp(name, "sample_fraction")) {
        if (!(value > 0.0 && value <= 1.0)) return fail("sample_fraction must be in (0,1]");
        h->sample_frac = value;
    } else if (!strcmp(name, "min_sample_slabs")) {
        if (!(value >= 1.0 && value <= 65536.0)) return fail("min_sample_slabs must be in 1..65536");
        h->min_sample_slabs = (int)value;
    } else if (!strcmp(name, "db_tiles_per_wave")) {
        h->nt_override = (int)value;
    } else if (!strcmp(name, "f32_tiles_per_wave")) {
        if (value != 2.0 && value != 4.0 && value != 8.0) return fail("f32_tiles_per_wave must be 2, 4 or 8");
        h->nt16 = (int)value;
        h->have_weights = false;          // operands are laid out per slab: set_weights must be called again
    } else if (!strcmp(name, "precision")) {
        if (value != 0.0 && value != 1.0) return fail("precision must be 0 (f64 sweep) or 1 (f32 prefilter + exact f64 re-rank)");
        h->precision = (int)value;
    } else if (!strcmp(name, "prefilter")) {
        if (value != 0.0 && value != 1.0 && value != 2.0) return fail("prefilter must be 0 (float32 operands), 1 or 2 (bf16-split operands where the shape has a variant: 3 / 4 MFMA terms per product)");
        CHK(no_batch_in_flight(h, "snk_set_option(prefilter)"));
        h->prefilter = (int)value;
        h->have_weights = false;          // the bf16 operands are built by set_weights
    } else if (!strcmp(name, "prefilter_balls")) {
        if (value != 0.0 && value != 1.0) return fail("prefilter_balls must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(prefilter_balls)"));
        h->prefilter_balls = (int)value;
        h->have_weights = false;          // the ball operand is built by set_weights
    } else if (!strcmp(name, "prefilter_super_balls")) {
        if (value != 0.0 && value != 1.0) return fail("prefilter_super_balls must be 0 or 1");
        h->prefilter_super_balls = (int)value;
    } else if (!strcmp(name, "prefilter_ball_bound")) {
        if (value != 0.0 && value != 1.0) return fail("prefilter_ball_bound must be 0 or 1");
        h->prefilter_ball_bound = (int)value;
    } else if (!strcmp(name, "coarse_gate_fraction")) {
        if (!(value >= 0.0 && value <= 1.0)) return fail("coarse_gate_fraction must be in 0..1");
        h->coarse_gate_fraction = value;
    } else if (!strcmp(name, "prefilter_two_pass")) {
        if (value != 0.0 && value != 1.0) return fail("prefilter_two_pass must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(prefilter_two_pass)"));
        h->prefilter_two_pass = (int)value;
    } else if (!strcmp(name, "reserved_cus")) {
        if (value < 0 || value > 64) return fail("reserved_cus must be in 0..64");
        h->reserved_cus = (int)value;
    } else if (!strcmp(name, "batch_rows")) {
        if (value < 0 || value > SNK_KNN_MAX_ROWS) return fail("batch_rows must be in 0..%d (0: one K-NN call per utterance)", (int)SNK_KNN_MAX_ROWS);
        h->batch_rows = (int)value;
    } else if (!strcmp(name, "greedy_f16")) {
        if (value != 0.0 && value != 1.0 && value != 2.0) return fail("greedy_f16 must be 0, 1 (streamed databases) or 2 (always)");
        h->greedy_f16 = (int)value;
    } else if (!strcmp(name, "greedy_test_stall")) {
        h->greedy_test_stall = value != 0.0;
    } else if (!strcmp(name, "greedy_fenced")) {
        h->greedy_fenced = value != 0.0;
    } else if (!strcmp(name, "greedy_resident")) {
        if (value != 0.0 && value != 1.0) return fail("greedy_resident must be 0 or 1");
        h->greedy_resident = (int)value;
    } else if (!strcmp(name, "greedy_hoist")) {
        if (value != 0.0 && value != 1.0) return fail("greedy_hoist must be 0 or 1");
        h->greedy_hoist = (int)value;
    } else if (!strcmp(name, "greedy_speculate")) {
        if (value != 0.0 && value != 1.0) return fail("greedy_speculate must be 0 or 1");
        h->greedy_speculate = (int)value;
    } else if (!strcmp(name, "greedy_hoist_fast")) {
        if (value != 0.0 && value != 1.0) return fail("greedy_hoist_fast must be 0 or 1");
        h->greedy_hoist_fast = (int)value;
    } else if (!strcmp(name, "greedy_hoist_max_gb")) {
        if (!(value >= 0.0)) return fail("greedy_hoist_max_gb must be >= 0");
        h->greedy_hoist_max_gb = value;
    } else if (!strcmp(name, "greedy_mode")) {
        if (value != 0.0 && value != 1.0 && value != 2.0) return fail("greedy_mode must be 0 (exact scan, a launch per step), 1 (float32 prefilter scan, one launch) or 2 (auto)");
        h->greedy_mode = (int)value;
    } else if (!strcmp(name, "viterbi_mode")) {
        if (value != 0.0 && value != 1.0 && value != 2.0) return fail("viterbi_mode must be 0 (dense exact join + recursion), 1 (lower bounds + sparse exact recursion) or 2 (auto)");
        CHK(no_batch_in_flight(h, "snk_set_option(viterbi_mode)"));
        h->viterbi_mode = (int)value;
    } else if (!strcmp(name, "viterbi_lb_chunk") || !strcmp(name, "viterbi_lb_warm") || !strcmp(name, "viterbi_lb_chunk_max_utts")) {
        if (!(value >= 0.0 && value <= 1e6) || value != (double)(int)value) return fail("%s must be a small non-negative integer", name);
        if (!strcmp(name, "viterbi_lb_warm") && value < 1.0) return fail("viterbi_lb_warm must be >= 1");
        CHK(no_batch_in_flight(h, "snk_set_option(viterbi_lb_*)"));
        (!strcmp(name, "viterbi_lb_chunk") ? h->lb_chunk : !strcmp(name, "viterbi_lb_warm") ? h->lb_warm : h->lb_chunk_max_utts) = (int)value;
    } else if (!strcmp(name, "viterbi_sparse_waves")) {
        // process-wide (a debugging / A-B switch): which form of the sparse exact recursion runs; same results
        if (value != 1.0 && value != 4.0) return fail("viterbi_sparse_waves must be 1 (one compute wavefront per utterance) or 4");
        CHK(no_batch_in_flight(h, "snk_set_option(viterbi_sparse_waves)"));
        set_viterbi_sparse_waves((int)value);
    } else if (!strcmp(name, "shard_gather_queries")) {
        if (value != 0.0 && value != 1.0) return fail("shard_gather_queries must be 0 or 1 (the same on every rank)");
        h->shard_gather_queries = (int)value;
    } else if (!strcmp(name, "shard_compact")) {
        if (value != 0.0 && value != 1.0) return fail("shard_compact must be 0 or 1 (the same on every rank)");
        h->shard_compact = (int)value;
    } else if (!strcmp(name, "shard_refine")) {
        if (value != 0.0 && value != 1.0) return fail("shard_refine must be 0 or 1 (the same on every rank)");
        h->shard_refine = (int)value;
    } else if (!strcmp(name, "join_bounds_stream")) {
        if (value != 0.0 && value != 1.0) return fail("join_bounds_stream must be 0 (main stream) or 1 (side stream of the group)");
        CHK(no_batch_in_flight(h, "snk_set_option(join_bounds_stream)"));
        h->join_bounds_stream = (int)value;
    } else if (!strcmp(name, "join_exact_form")) {
        if (value != 0.0 && value != 1.0) return fail("join_exact_form must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(join_exact_form)"));
        set_join_exact_form((int)value);
    } else if (!strcmp(name, "viterbi_weights")) {
        if (value != 0.0 && value != 1.0) return fail("viterbi_weights must be 0 (float64) or 1 (OpenFST's float32 weights)");
        CHK(no_batch_in_flight(h, "snk_set_option(viterbi_weights)"));
        h->viterbi_weights = (int)value;
    } else if (!strcmp(name, "join_lb_variant")) {
        if (value != 0.0 && value != 1.0) return fail("join_lb_variant must be 0 or 1");
        h->join_lb_variant = (int)value;
    } else if (!strcmp(name, "join_beta")) {
        if (!(value >= 0.0 && value <= 10.0)) return fail("join_beta must be in 0..10");
        h->join_beta = value;
    } else if (!strcmp(name, "pool_chunk_limit")) {
        if (value < 0 || value > 1e6) return fail("pool_chunk_limit must be in 0..1e6");
        h->pool_chunk_limit = (int)value;
    } else if (!strcmp(name, "timers")) {
        h->timers_on = value != 0.0;
    } else {
        return fail("snk_set_option: unknown option '%s'", name);
    }
    return 0;
}

int snk_get_info(snk_handle h, const char *name, double *out)
{
    if (!h || !name || !out) return fail("snk_get_info: null argument");
    if (!strcmp(name, "n_units")) *out = (double)h->N;
    else if (!strcmp(name, "target_dim")) *out = h->Dt;
    else if (!strcmp(name, "join_dim")) *out = h->Dj;
    else if (!strcmp(name, "last_knn_retries")) *out = h->last_retries;
    else if (!strcmp(name, "list_capacity")) *out = h->cap;
    else if (!strcmp(name, "tie_overflow")) *out = h->tie_overflow;
    else if (!strcmp(name, "exact_row_fallbacks")) *out = h->exact_row_fallbacks;
    else if (!strcmp(name, "batch_redos")) *out = h->batch_redos;
    else if (!strcmp(name, "pool_overflows")) *out = h->pool_overflows;
    else if (!strcmp(name, "viterbi_mode")) *out = h->viterbi_mode;
    else if (!strcmp(name, "join_lb_variant")) *out = h->join_lb_variant;
    else if (!strcmp(name, "viterbi_weights")) *out = h->viterbi_weights;
    else if (!strcmp(name, "greedy_mode")) *out = h->greedy_mode;
    else if (!strcmp(name, "greedy_hoist")) *out = h->greedy_hoist;
    else if (!strcmp(name, "greedy_hoist_launches")) *out = (double)h->greedy_hoist_launches;
    else if (!strcmp(name, "greedy_hoist_fast")) *out = h->greedy_hoist_fast;
    else if (!strcmp(name, "greedy_speculate")) *out = h->greedy_speculate;
    else if (!strcmp(name, "greedy_hoist16_launches")) *out = (double)h->greedy_hoist16_launches;
    else if (!strcmp(name, "greedy_last_undecided_step")) *out = (double)h->greedy_last_status[0] - 1.0;     // -1: every step was decided
    else if (!strcmp(name, "greedy_last_watchdog")) *out = (double)h->greedy_last_status[3];
    else if (!strcmp(name, "greedy_last_speculated")) *out = (double)h->greedy_last_status[4];         // streamed scan: steps decided before the gather
    else if (!strcmp(name, "greedy_last_several_holders")) *out = (double)h->greedy_last_status[5];   // streamed scan: steps with windows inside the bound in several workgroups
    else if (!strcmp(name, "greedy_last_why_candidates")) *out = (double)h->greedy_last_status[4];
    else if (!strcmp(name, "greedy_last_why_third")) *out = (double)h->greedy_last_status[5];
    else if (!strcmp(name, "greedy_last_why_min")) { double v; memcpy(&v, &h->greedy_last_status[6], 8); *out = v; }
    else if (!strcmp(name, "greedy_last_why_tau")) { double v; memcpy(&v, &h->greedy_last_status[7], 8); *out = v; }
    else if (!strcmp(name, "greedy_resident")) *out = h->greedy_resident;
    else if (!strcmp(name, "greedy_resident_launches")) *out = (double)h->greedy_resident_launches;
    else if (!strcmp(name, "greedy_f16")) *out = h->greedy_f16;
    else if (!strcmp(name, "greedy_f16_launches")) *out = (double)h->greedy_f16_launches;
    else if (!strcmp(name, "greedy_f16_delta")) *out = h->g16_delta;
    else if (!strcmp(name, "greedy_exact_windows")) *out = (double)h->greedy_exact_windows;
    else if (!strcmp(name, "greedy_second_rounds")) *out = (double)h->greedy_second_rounds;
    else if (!strcmp(name, "greedy_stalls")) *out = (double)h->greedy_stalls;
    else if (!strcmp(name, "greedy_fallbacks")) *out = h->greedy_fallbacks;
    else if (!strcmp(name, "greedy_second_phase_rounds") || !strcmp(name, "greedy_exact_windows")) {
        // statistics of the most recent float32 scan launch: steps that needed every lane's candidates; windows
        // whose canonical float64 totals decided a step
        int64_t v[3] = {0, 0, 0};
        if (h->g32_ctl.p) CHK(d2h_sync(h, v, reinterpret_cast<char *>(h->g32_ctl.p) + 16, sizeof(v), h->stream));
        *out = (double)v[!strcmp(name, "greedy_exact_windows") ? 2 : 1];
    }
    else if (!strcmp(name, "dense_cells") || !strcmp(name, "dense_steps") || !strcmp(name, "dense_exact_costs") || !strcmp(name, "set_overflows")) {
        unsigned long long v[4] = {0, 0, 0, 0};
        if (h->vstats.p) { HIPCHK(hipDeviceSynchronize()); CHK(d2h_sync(h, v, h->vstats.p, sizeof(v), h->stream)); }
        *out = (double)v[!strcmp(name, "dense_steps") ? 1 : (!strcmp(name, "dense_exact_costs") ? 2 : (!strcmp(name, "set_overflows") ? 3 : 0))];
    }
    else if (!strcmp(name, "f16_ready")) *out = h->f16_ready ? 1 : 0;
    else if (!strcmp(name, "f16_fallbacks")) *out = h->f16_fallbacks;
    else if (!strcmp(name, "last_f16_status")) *out = h->last_f16_status;
    else if (!strcmp(name, "pool_chunks_used")) { unsigned int v[2] = {0, 0}; CHK(d2h_sync(h, v, h->poolctl.p, sizeof(v), h->stream)); *out = v[0] + 1e6 * v[1]; }
    else if (!strcmp(name, "precision")) *out = h->precision;
    else if (!strcmp(name, "prefilter")) *out = h->prefilter;
    else if (!strcmp(name, "prefilter_bf16_active")) *out = h->bf16_ready ? 1 : 0;
    else if (!strcmp(name, "prefilter_rho_lo") || !strcmp(name, "prefilter_rho_res")) {
        // sqrt of the largest ||fl||^2 / ||f||^2 (||rf||^2 / ||f||^2) over the database rows: 2^-8 (2^-16) at worst
        double rho[2] = {0.0, 0.0};
        if (h->bf16_ready && h->rho16.p) {
            HIPCHK(hipSetDevice(h->device));
            HIPCHK(hipStreamSynchronize(h->stream));
            CHK(d2h_sync(h, rho, h->rho16.p, sizeof(rho), h->stream));
        }
        *out = sqrt(rho[name[14] == 'l' ? 0 : 1]);
    }
    else if (!strcmp(name, "prefilter_margin_rows") || !strcmp(name, "prefilter_min_margin")) {
        // since the engine was created (or the last snk_reset_timers): rows of prefilter K-NN calls whose exact K-th key
        // came within 2 eps of the filter threshold, and the smallest (threshold - exact K-th key) / eps seen
        unsigned int v[2] = {0u, 0x7f800000u};
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        CHK(d2h_sync(h, v, h->margin_stat.p, sizeof(v), h->stream));
        float r; memcpy(&r, &v[1], 4);
        *out = name[10] == 'm' && name[11] == 'a' ? (double)v[0] : (double)r;
    }
    else if (!strcmp(name, "prefilter_two_pass")) *out = h->prefilter_two_pass;
    else if (!strcmp(name, "prefilter_balls")) *out = h->prefilter_balls;
    else if (!strcmp(name, "prefilter_ball_bound")) *out = h->prefilter_ball_bound;
    else if (!strcmp(name, "prefilter_super_balls")) *out = h->prefilter_super_balls;
    else if (!strcmp(name, "wide_launches")) *out = (double)h->wide_launches;
    else if (!strcmp(name, "shard_compact")) *out = h->shard_compact;
    else if (!strcmp(name, "shard_last_sent_mb")) *out = h->shard_last_sent_mb;
    else if (!strcmp(name, "shard_last_padded_mb")) *out = h->shard_last_padded_mb;
    else if (!strcmp(name, "wide_ready")) *out = h->wide16_ready ? 1 : 0;
    else if (!strcmp(name, "filter_onepass")) *out = h->filter_onepass ? 1 : 0;     // 1: the coarse sweep listed most pairs for this voice: one-pass sweep since
    else if (!strcmp(name, "filter_coarse")) *out = h->filter_coarse ? 1 : 0;       // 1: the ball pass listed too many pairs for this voice
    else if (!strcmp(name, "ball_switches")) *out = (double)h->ball_switches;
    else if (!strcmp(name, "coarse_pairs") || !strcmp(name, "coarse_pair_overflow")) {
        // tile pairs the coarse pass of the most recent two-pass filter let through (debug / tuning aid)
        unsigned int v[2] = {0u, 0u};
        if (h->cpairctl.p) { HIPCHK(hipSetDevice(h->device)); HIPCHK(hipStreamSynchronize(h->stream)); CHK(d2h_sync(h, v, h->cpairctl.p, sizeof(v), h->stream)); }
        *out = (double)v[name[11] == 's' ? 0 : 1];
    }
    else if (!strcmp(name, "prefilter_mfma_unit")) *out = SNK_BF16_MFMA_UNIT;
    else if (!strcmp(name, "prefilter_eps_c")) *out = (h->bf16_ready && h->prefilter >= 1) ? h->eps_c_bf : h->eps_c;
    else if (!strcmp(name, "batch_rows")) *out = h->batch_rows;
    else if (!strcmp(name, "last_list_mean") || !strcmp(name, "last_list_max")) {
        // candidate-list lengths of the most recent K-NN call (debug / tuning aid)
        const int64_t n = h->last_T;
        if (n <= 0) { *out = 0; return 0; }
        std::vector<int> c((size_t)n);
        CHK(d2h_sync(h, c.data(), h->cnt.p, (size_t)n * sizeof(int), h->stream));
        double sum = 0, mx = 0;
        for (int64_t i = 0; i < n; ++i) { sum += c[i]; if (c[i] > mx) mx = c[i]; }
        *out = !strcmp(name, "last_list_max") ? mx : sum / (double)n;
    }
    else if (!strcmp(name, "device")) *out = h->device;
    else if (!strcmp(name, "db_tiles_per_wave")) { KnnPlan p = make_plan(h, 100); *out = p.nt; }
    else if (!strcmp(name, "sample_slabs")) { KnnPlan p = make_plan(h, 100); *out = (double)p.a_count; }
    else if (!strcmp(name, "n_slabs")) { KnnPlan p = make_plan(h, 100); *out = (double)p.n_slabs; }
    else return fail("snk_get_info: unknown item '%s'", name);
    return 0;
}

static int selftest_mfma16(snk_engine *h, double *err_out)
{
    float A[64], B[64], C[1024], R[1024];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 2; ++k) A[i * 2 + k] = (float)((3 * i + 7 * k + 1) % 11 - 5);
    for (int k = 0; k < 2; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((5 * k - 2 * j + (k * j) % 3) % 7);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float acc = 0; for (int k = 0; k < 2; ++k) acc += A[i * 2 + k] * B[k * 32 + j]; R[i * 32 + j] = acc; }
    DevBuf dA, dB, dC;
    CHK(dA.ensure(sizeof(A))); CHK(dB.ensure(sizeof(B))); CHK(dC.ensure(sizeof(C)));
    CHK(h2d_sync(h, dA.p, A, sizeof(A)));
    CHK(h2d_sync(h, dB.p, B, sizeof(B)));
    HIPCHK(hipMemset(dC.p, 0, sizeof(C)));
    launch_mfma16_selftest(dA.as<float>(), dB.as<float>(), dC.as<float>(), h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    CHK(d2h_sync(h, C, dC.p, sizeof(C), h->stream));
    double err = 0;
    for (int i = 0; i < 1024; ++i) err = fmax(err, fabs((double)C[i] - (double)R[i]));
    dA.release(); dB.release(); dC.release();
    *err_out = err;
    return 0;
}

// One v_mfma_f32_32x32x16_bf16 on the caller's bit patterns (include/snk.h): the probe behind the accumulation term of
// the bf16-split prefilter's bound.
int snk_probe_mfma_bf16(snk_handle h, const uint16_t *A, const uint16_t *B, const float *C, float *D_out)
{
    if (!h) return fail("null handle");
    if (!A || !B || !C || !D_out) return fail("snk_probe_mfma_bf16: null argument");
    HIPCHK(hipSetDevice(h->device));
    DevBuf dA, dB, dC, dD;
    CHK(dA.ensure(512 * 2)); CHK(dB.ensure(512 * 2)); CHK(dC.ensure(1024 * 4)); CHK(dD.ensure(1024 * 4));
    CHK(h2d_sync(h, dA.p, A, 512 * 2));
    CHK(h2d_sync(h, dB.p, B, 512 * 2));
    CHK(h2d_sync(h, dC.p, C, 1024 * 4));
    launch_mfma_bf16_probe(dA.as<unsigned short>(), dB.as<unsigned short>(), dC.as<float>(), dD.as<float>(), h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    CHK(d2h_sync(h, D_out, dD.p, 1024 * 4, h->stream));
    dA.release(); dB.release(); dC.release(); dD.release();
    return 0;
}

int snk_selftest_mfma(snk_handle h, double *max_abs_err_out)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    double A[64], B[64], C[256], R[256];
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 4; ++k) A[i * 4 + k] = (double)(3 * i + 7 * k + 1);      // asymmetric
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 16; ++j) B[k * 16 + j] = (double)(5 * k - 2 * j + (k * j) % 3);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double acc = 0;
            for (int k = 0; k < 4; ++k) acc += A[i * 4 + k] * B[k * 16 + j];
            R[i * 16 + j] = acc;
        }
    DevBuf dA, dB, dC;
    CHK(dA.ensure(sizeof(A))); CHK(dB.ensure(sizeof(B))); CHK(dC.ensure(sizeof(C)));
    CHK(h2d_sync(h, dA.p, A, sizeof(A)));
    CHK(h2d_sync(h, dB.p, B, sizeof(B)));
    HIPCHK(hipMemset(dC.p, 0, sizeof(C)));
    launch_mfma_selftest(dA.as<double>(), dB.as<double>(), dC.as<double>(), h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    CHK(d2h_sync(h, C, dC.p, sizeof(C), h->stream));
    double err = 0;
    for (int i = 0; i < 256; ++i) err = fmax(err, fabs(C[i] - R[i]));
    dA.release(); dB.release(); dC.release();
    double err16 = 0;
    CHK(selftest_mfma16(h, &err16));          // f16 32x32x16 operand / result maps of the prefilter
    // the bf16 instruction of the split prefilter: operand map (small integers: exact) and the accumulation assumption
    // behind its key bound on the pattern that shows the unit's cut (one product of 1, fifteen just under 2^-24)
    double errbf = 0;
    {
        uint16_t A[512], B[512];
        float C[1024], D[1024];
        auto bits = [](float x) { unsigned int u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); };   // exact for the values used
        for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = bits((float)((3 * i + 5 * k) % 9 - 4));
        for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = bits((float)((7 * k - 2 * j) % 5));
        for (int i = 0; i < 1024; ++i) C[i] = (float)(i % 7);
        CHK(snk_probe_mfma_bf16(h, A, B, C, D));
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double acc = (double)C[i * 32 + j];
            for (int k = 0; k < 16; ++k) acc += (double)((3 * i + 5 * k) % 9 - 4) * (double)((7 * k - 2 * j) % 5);
            errbf = fmax(errbf, fabs((double)D[i * 32 + j] - acc));
        }
        for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = bits(k == 0 ? 1.0f : 0.000244140625f);          // 2^-12
        for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = bits(k == 0 ? 1.0f : 0.000236511230469f);        // 1.9375 2^-13
        for (int i = 0; i < 1024; ++i) C[i] = 0.f;
        CHK(snk_probe_mfma_bf16(h, A, B, C, D));
        const double small = 0.000244140625 * 0.000236511230469, exact = 1.0 + 15.0 * small, mass = exact;
        for (int i = 0; i < 1024; ++i)
            if (fabs((double)D[i] - exact) > SNK_BF16_MFMA_UNIT * mass) errbf = fmax(errbf, fabs((double)D[i] - exact));
    }
    if (max_abs_err_out) *max_abs_err_out = fmax(fmax(err, err16), errbf);
    return 0;
}

}  // extern "C"
