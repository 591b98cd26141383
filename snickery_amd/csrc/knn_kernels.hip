// K-NN candidate preselection for gfx950 (MI355X): brute force over the whole unit
// database in float64 on the MFMA pipe (v_mfma_f64_16x16x4_f64).
//
// Replaces scipy.spatial.cKDTree(F).query(U, k=K) of the reference
// (script/synth_halfphone.py:379,1364; per-phone trees :385-402,:1384).
//
// Pipeline (all on one stream):
//   prepare_queries : pad Q to [Tpad][Dpad], row norms
//   sweep<MODE 0>   : stage A, strided sample of DB slabs -> per (row, lane-group) minima
//   threshold       : K-th smallest group minimum  = valid upper bound of the K-th NN key
//   sweep<MODE 1>   : stage B, whole DB; (row, unit, key) entries with key <= threshold are
//                     appended to wave-private chunks of a global entry pool (no atomics and
//                     no waits in the MFMA loop: plain stores at a wave-uniform cursor)
//   bucket          : entry pool -> per-row candidate lists
//   finalize        : per row: sort list, take top K (+ near ties), recompute those
//                     distances exactly in the canonical (oracle) order, final sort
//
// "key" = ||f||^2 - 2 q.f  (the row-constant ||q||^2 is dropped for ranking).
//
// Sweep kernel design (DB-stationary): one wavefront owns a slab of 16*NT DB rows and keeps
// its MFMA B-fragments in registers for the whole kernel; the query tiles (16 rows) stream
// through as A-fragments straight from L2 (Q is a few hundred KB), double buffered in
// registers.  No LDS tiles, no workgroup barriers: every DB element is read from HBM exactly
// once per sweep and the kernel is bound by the f64 MFMA rate.  The k index of the MFMA is
// permuted (lane group g supplies elements 16g..16g+15 of each 64-column chunk) so that every
// lane loads 128 contiguous bytes per fragment.
#include "snk_internal.h"
#include <type_traits>
#include <float.h>

namespace snk {

typedef double d4 __attribute__((ext_vector_type(4)));

#define POOL_CHUNK 2048         // entries per wave-private chunk of the entry pool
struct __attribute__((aligned(16))) PoolEntry { double key; int idx; int row; };

// ---------------------------------------------------------------------------
// f64 MFMA C/D fragment mapping (gfx950): lane l holds D[row = (l>>4) + 4*r][col = l&15]
// for r = 0..3.  A: lane supplies A[row = l&15][k = l>>4]; B: B[k = l>>4][col = l&15].
// snk_selftest_mfma() checks this mapping on the device.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int frag_row(int lane, int r) { return (lane >> 4) + 4 * r; }

template <int NT, int DCH, int MODE, bool CLS, int WPS>
__global__ void __launch_bounds__(256, WPS)
knn_sweep(const double *__restrict__ Fw, const double *__restrict__ fnorm,
          const double *__restrict__ Qf, const double *__restrict__ thr,
          int nQT, int64_t row_stride, int64_t row_limit, int64_t n_slabs,
          int64_t wave_stride, int64_t tile_stride, unsigned int *__restrict__ slab_counter,
          int qsplit, int64_t n_main_slabs, int qsplit_tail,
          double *__restrict__ gmin, int64_t G,
          PoolEntry *__restrict__ pool, unsigned int *__restrict__ pool_ctl,
          int *__restrict__ chunk_fill, int max_chunks,
          const int32_t *__restrict__ unit_class, const int32_t *__restrict__ query_class)
{
    constexpr int KS = DCH * 16;          // MFMA k-steps per output tile
    constexpr int DP = DCH * 64;          // padded feature columns
    // per-wave LDS staging of passing entries: the MFMA loop itself issues no vector-memory
    // stores (stores share vmcnt with the query prefetch, and the rotate's wait would stall on
    // a store issued late in the tile); the stage is flushed to the pool at the next tile top,
    // a whole tile before anything waits on vmcnt again
    constexpr int STAGE_CAP = 64 * (((WPS >= 2) ? 1 : ((NT >= 2) ? 2 : 1)) * 4) + 256;  // one step's worst case + slack
    __shared__ PoolEntry stage[(MODE == 1) ? 4 : 1][(MODE == 1) ? STAGE_CAP : 1];

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;

    // Persistent waves: the grid covers (CUs - reserved) compute units once; every wave pulls
    // slab indices from a device-wide counter until the database is exhausted (dynamic tail
    // balancing; the reserved CUs stay free for the Viterbi recursion of the previous utterance).
    auto grab_slab = [&]() -> int64_t {
        unsigned int v = 0;
        if (lane == 0) v = atomicAdd(slab_counter, 1u);
        return (int64_t)__builtin_amdgcn_readfirstlane(v);
    };

    // entry-pool cursor (MODE 1): this wave appends to a private chunk of the global pool at a
    // wave-uniform position; a new chunk costs one returning atomic per POOL_CHUNK entries.
    // pool_ctl[0] = chunks handed out, pool_ctl[1] = overflow flag.
    int chunk_id = -1;
    int cused = POOL_CHUNK;      // forces a chunk grab before the first append
    int lcount = 0;              // entries waiting in this wave's LDS stage
    auto new_chunk = [&]() {
        if (chunk_id >= 0 && lane == 0) chunk_fill[chunk_id] = cused;
        unsigned int c = 0;
        if (lane == 0) c = atomicAdd(&pool_ctl[0], 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        if ((int)c >= max_chunks) {          // pool exhausted: drop, the host retries
            if (lane == 0) pool_ctl[1] = 1u;
            chunk_id = -1;
        } else {
            chunk_id = (int)c;
        }
        cused = 0;
    };

    auto flush_stage = [&]() {
        if (cused + lcount > POOL_CHUNK) new_chunk();
        if (chunk_id >= 0) {
            for (int e = lane; e < lcount; e += 64)
                pool[(int64_t)chunk_id * POOL_CHUNK + cused + e] = stage[wv][e];
        }
        cused += lcount;
        lcount = 0;
    };

    // work item = (slab, part): the query tiles of a slab may be split over `qsplit` items so that
    // a small sweep (stage A) still spreads over every compute unit
    // whole rounds of items first; the slabs of the last, partial round are cut into more parts
    // so that the tail of the persistent sweep stays short
    const int64_t n_main_items = n_main_slabs * qsplit;
    const int64_t n_items = n_main_items + (n_slabs - n_main_slabs) * qsplit_tail;
    int64_t item = grab_slab();
    while (item < n_items) {
    const int64_t item_next = grab_slab();
    const bool tail = item >= n_main_items;
    const int qs = tail ? qsplit_tail : qsplit;
    const int64_t rel = tail ? item - n_main_items : item;
    const int64_t w = (tail ? n_main_slabs : 0) + rel / qs;
    const int part = (int)(rel % qs);
    const int qt_lo = (int)(((int64_t)nQT * part) / qs);
    const int qt_hi = (int)(((int64_t)nQT * (part + 1)) / qs);
    const int n_tiles = qt_hi - qt_lo;
    // Row mapping: tile nt, lane-row r16 of slab w holds database row
    //     (w*wave_stride + nt*tile_stride + r16) * row_stride
    // stage B (filter): wave_stride = 16*NT, tile_stride = 16, row_stride = 1 -> a contiguous slab.
    // stage A (minima): row_stride = sampling stride (uniform at single-unit granularity, which
    //   matters for speech databases where the neighbours of a target are runs of consecutive
    //   units), wave_stride = 16 and tile_stride = 16*slabs, so that the NT rows that share one
    //   minimum (same slab, same lane) lie far apart and a run of consecutive units lands in
    //   distinct groups.
    const int64_t base = w * wave_stride;

    // ---- B fragments: this wave's DB rows, resident for the whole kernel ----
    double b[NT][KS];
    double fn[NT];
    int ucls[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int64_t row = (base + nt * tile_stride + r16) * row_stride;
        if (row >= row_limit) row = row_limit;          // a padding row (norm = +inf)
        const double *src = Fw + row * DP + 16 * g;
#pragma unroll
        for (int ch = 0; ch < DCH; ++ch)
#pragma unroll
            for (int s = 0; s < 16; s += 2) {
                double2 v = *reinterpret_cast<const double2 *>(src + ch * 64 + s);
                b[nt][ch * 16 + s] = v.x;
                b[nt][ch * 16 + s + 1] = v.y;
            }
        fn[nt] = fnorm[row];
        if (CLS) ucls[nt] = unit_class[row];
    }

    // query tiles are visited in a per-wave rotated order so that concurrent waves
    // spread their list appends over all rows instead of hammering the same 16 counters
    int qt = qt_lo + (int)((w * 5) % n_tiles);

    double a_cur[KS], a_nxt[KS];
    double th_cur[4], th_nxt[4];
    int qc_cur[4], qc_nxt[4];
    // query tiles are read from the FRAGMENT-ORDER copy Qf[tile][chunk][pair][lane][2]: every
    // load instruction of the wave covers 1 KB of contiguous memory
    auto load_tile = [&](int t, double (&a)[KS], double (&th)[4], int (&qc)[4]) {
        const double *src = Qf + (int64_t)t * (16 * DP) + lane * 2;
#pragma unroll
        for (int ch = 0; ch < DCH; ++ch)
#pragma unroll
            for (int s = 0; s < 16; s += 2) {
                double2 v = *reinterpret_cast<const double2 *>(src + (ch * 8 + s / 2) * 128);
                a[ch * 16 + s] = v.x;
                a[ch * 16 + s + 1] = v.y;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qrow = t * 16 + frag_row(lane, r);
            if (MODE == 1) th[r] = thr[qrow];
            if (CLS) qc[r] = query_class[qrow];
        }
    };
    load_tile(qt, a_nxt, th_nxt, qc_nxt);

    // all prologue loads (B fragments, first query tile) land before the tile loop, so that
    // inside it the only vmcnt wait is the query double-buffer rotate
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(b[nt][s]));
        asm volatile("" : "+v"(fn[nt]));
        if (CLS) asm volatile("" : "+v"(ucls[nt]));
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(a_nxt[s]));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (MODE == 1) asm volatile("" : "+v"(th_nxt[r]));
        if (CLS) asm volatile("" : "+v"(qc_nxt[r]));
    }
    // Software pipeline over (tile, step): while the MFMA chain of one step issues, the VALU
    // epilogue of the PREVIOUS step (keys, threshold compares) runs in the MFMA shadows; only
    // the rare "some key passed" case branches, after the chain.
    constexpr int CH = (WPS >= 2) ? 1 : ((NT >= 2) ? 2 : 1);   // database tiles per step (independent MFMA chains)
    constexpr int NSTEP = NT / CH;
    constexpr int NEL = CH * 4;                  // results per lane per step
    constexpr int GAP = KS / NEL;                // MFMA k-steps between two epilogue elements
    d4 pacc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) pacc[j] = d4{0.0, 0.0, 0.0, 0.0};
    double th_prev[4];
    int qc_prev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { th_prev[r] = -DBL_MAX; qc_prev[r] = -2; }
    int qt_prev = qt;
    double mn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mn[r] = DBL_MAX;

    // one query tile: A holds this tile's fragments (prefetched a tile ago), A_load receives the
    // next tile's.  The two register sets ping-pong (the tile loop is unrolled by two), so no
    // fragment is ever copied.
    auto tile_body = [&](double (&A)[KS], double (&A_load)[KS], int it) {
        // the one vmcnt wait per tile: everything outstanding here is a whole tile old
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(A[s]));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == 1) asm volatile("" : "+v"(th_nxt[r]));
            if (CLS) asm volatile("" : "+v"(qc_nxt[r]));
            th_cur[r] = th_nxt[r]; qc_cur[r] = qc_nxt[r];
        }

        // entries staged during the previous tile go out now, a whole tile before the next wait
        if (MODE == 1 && lcount) flush_stage();

        // prefetch the next tile (unconditional: the last one wraps and is simply unused)
        const int qt_next = (qt + 1 == qt_hi) ? qt_lo : qt + 1;
        load_tile(qt_next, A_load, th_nxt, qc_nxt);

#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            // the pending step: previous step of this tile, or the last step of the previous tile
            constexpr bool dummy = false; (void)dummy;
            const int pnt = (st > 0) ? (st - 1) * CH : (NSTEP - 1) * CH;
            const double (&pth)[4] = (st > 0) ? th_cur : th_prev;
            const int (&pqc)[4] = (st > 0) ? qc_cur : qc_prev;
            const int pqt = (st > 0) ? qt : qt_prev;
            d4 acc[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
            // room in the LDS stage for this step's worst case (rarely taken)
            if (MODE == 1 && lcount > STAGE_CAP - 64 * NEL) flush_stage();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int j = 0; j < CH; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[s], b[st * CH + j][s], acc[j], 0, 0, 0);
                if (s % GAP == (GAP > 1 ? 1 : 0) && s / GAP < NEL) {
                    // one result of the PENDING step per MFMA gap: key, threshold test and (for the
                    // few that pass) a predicated 16-byte LDS store -- all in the MFMA shadow
                    const int e = s / GAP, j = e / 4, r = e % 4;
                    const double key = __builtin_fma(-2.0, pacc[j][r], fn[pnt + j]);
                    bool ok = true;
                    if (CLS) ok = (ucls[pnt + j] == pqc[r]);
                    if (MODE == 0) { if (ok) mn[r] = fmin(mn[r], key); }
                    else {
                        const bool pass = ok && (key <= pth[r]);
                        const unsigned long long m = __ballot(pass);
                        if (pass) {
                            const int rank = __builtin_amdgcn_mbcnt_hi(
                                (unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                            PoolEntry en;
                            en.key = key;
                            en.idx = (int)(base + (pnt + j) * tile_stride + r16);
                            en.row = pqt * 16 + frag_row(lane, r);
                            stage[wv][lcount + rank] = en;
                        }
                        lcount += __popcll(m);
                    }
                }
            }
            if (MODE == 0 && st == 0) {
                // minima of the previous tile are complete now
                if (it > 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        gmin[((int64_t)qt_prev * 16 + frag_row(lane, r)) * G + w * 16 + r16] = mn[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) mn[r] = DBL_MAX;
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) pacc[j] = acc[j];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { th_prev[r] = th_cur[r]; qc_prev[r] = qc_cur[r]; }
        qt_prev = qt;
        qt = qt_next;
    };
    for (int it = 0; it < n_tiles; it += 2) {
        tile_body(a_nxt, a_cur, it);
        if (it + 1 < n_tiles) tile_body(a_cur, a_nxt, it + 1);
    }
    // drain the last pending step of this slab (not overlapped: once per slab)
    {
        const int pnt = (NSTEP - 1) * CH;
        if (MODE == 1 && lcount > STAGE_CAP - 64 * NEL) flush_stage();
#pragma unroll
        for (int e = 0; e < NEL; ++e) {
            const int j = e / 4, r = e % 4;
            const double key = __builtin_fma(-2.0, pacc[j][r], fn[pnt + j]);
            bool ok = true;
            if (CLS) ok = (ucls[pnt + j] == qc_prev[r]);
            if (MODE == 0) { if (ok) mn[r] = fmin(mn[r], key); }
            else {
                const bool pass = ok && (key <= th_prev[r]);
                const unsigned long long m = __ballot(pass);
                if (pass) {
                    const int rank = __builtin_amdgcn_mbcnt_hi(
                        (unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    PoolEntry en;
                    en.key = key;
                    en.idx = (int)(base + (pnt + j) * tile_stride + r16);
                    en.row = qt_prev * 16 + frag_row(lane, r);
                    stage[wv][lcount + rank] = en;
                }
                lcount += __popcll(m);
            }
        }
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gmin[((int64_t)qt_prev * 16 + frag_row(lane, r)) * G + w * 16 + r16] = mn[r];
        }
    }
    item = item_next;
    }   // work-item loop

    if (MODE == 1) {
        if (lcount) flush_stage();
        if (chunk_id >= 0 && lane == 0) chunk_fill[chunk_id] = cused;
    }
}

// Persistent sweeps hand out (slab, part) items round by round.  When the item count is not a
// multiple of the resident waves the last round is partial: its slabs are cut into `*qtail` parts
// (>= qsplit, <= 8) so that about one full round of short items finishes the sweep.
void sweep_tail_split(int64_t n_slabs, int qsplit, int64_t waves, int nQT, int64_t *n_main, int *qtail)
{
    *n_main = n_slabs;
    *qtail = qsplit;
    if (waves <= 0) return;
    const int64_t items = n_slabs * qsplit;
    const int64_t full_rounds = items / waves;
    const int64_t main_slabs = (full_rounds * waves) / qsplit;
    const int64_t rest = n_slabs - main_slabs;
    if (rest <= 0 || full_rounds == 0) return;
    int q = qsplit;
    while (rest * q < waves && q * 2 <= nQT && q < 8) q *= 2;
    if (q == qsplit) return;
    *n_main = main_slabs;
    *qtail = q;
}

template <int NT, int DCH, int WPS>
static void launch_sweep_t(int mode, bool cls, int blocks, hipStream_t s,
                           const double *Fw, const double *fnorm, const double *Qp,
                           const double *thr, int nQT, int64_t rstride, int64_t rlimit, int64_t ns,
                           int64_t wstride, int64_t tstride, unsigned int *ctr, int qsplit,
                           int64_t n_main, int qtail, double *gmin, int64_t G, PoolEntry *pool, unsigned int *pool_ctl,
                           int *chunk_fill, int max_chunks,
                           const int32_t *uc, const int32_t *qc)
{
#define SNK_LAUNCH(MODE, CLS)                                                                    \
    hipLaunchKernelGGL((knn_sweep<NT, DCH, MODE, CLS, WPS>), dim3(blocks), dim3(256), 0, s, Fw,  \
                       fnorm, Qp, thr, nQT, rstride, rlimit, ns, wstride, tstride, ctr, qsplit, n_main, qtail, \
                       gmin, G,                                                                   \
                       pool, pool_ctl, chunk_fill, max_chunks, uc, qc)
    if (mode == 0) { if (cls) SNK_LAUNCH(0, true); else SNK_LAUNCH(0, false); }
    else           { if (cls) SNK_LAUNCH(1, true); else SNK_LAUNCH(1, false); }
#undef SNK_LAUNCH
}

static void launch_sweep(const KnnPlan &p, int mode, int64_t rstride, int64_t rlimit, int64_t ns,
                         int64_t wstride, int64_t tstride,
                         const double *Fw, const double *fnorm, const double *Qp,
                         const double *thr, int64_t Tpad, double *gmin, int64_t G, void *pool,
                         unsigned int *pool_ctl, int *chunk_fill, int max_chunks,
                         const int32_t *uc, const int32_t *qc, hipStream_t s)
{
    if (ns <= 0) return;
    const int wps = (p.nt == 4 && p.dch == 1) ? 2 : 1;       // NT=4 fits two waves per SIMD
    const int64_t max_blocks = (int64_t)p.grid_cus * wps;
    const int nQT = (int)(Tpad / 16);
    // split the query tiles of a slab while there are fewer slabs than ~2 per resident wave
    int qsplit = 1;
    while (ns * qsplit < 2 * 4 * max_blocks && qsplit * 2 <= nQT && qsplit < 8) qsplit *= 2;
    int64_t blocks = (ns * qsplit + 3) / 4;
    if (blocks > max_blocks) blocks = max_blocks;
    const bool cls = (uc != nullptr);
    unsigned int *ctr = p.slab_counter + mode;               // zeroed by knn_reset
    int64_t n_main = ns;
    int qtail = qsplit;
    sweep_tail_split(ns, qsplit, blocks * 4, nQT, &n_main, &qtail);
#define SNK_CASE(NT_, DCH_, WPS_)                                                              \
    if (p.nt == NT_ && p.dch == DCH_) {                                                        \
        launch_sweep_t<NT_, DCH_, WPS_>(mode, cls, (int)blocks, s, Fw, fnorm, Qp, thr, nQT, rstride, \
                                        rlimit, ns, wstride, tstride, ctr, qsplit, n_main, qtail, gmin, G, \
                                        reinterpret_cast<PoolEntry *>(pool), pool_ctl, chunk_fill, \
                                        max_chunks, uc, qc);                                   \
        return;                                                                                \
    }
    SNK_CASE(8, 1, 1) SNK_CASE(4, 1, 2) SNK_CASE(2, 1, 1)
    SNK_CASE(4, 2, 1) SNK_CASE(2, 2, 1)
    SNK_CASE(2, 3, 1)
    SNK_CASE(1, 4, 1)
#undef SNK_CASE
}

void launch_knn_minima(const KnnPlan &p, const double *Fw, const double *fnorm, const double *Qp,
                       int64_t Tpad, double *gmin, int64_t G, const int32_t *uc,
                       const int32_t *qc, hipStream_t s)
{
    launch_sweep(p, 0, p.a_stride, p.row_limit, p.a_count, 16, 16 * p.a_count, Fw, fnorm, Qp, nullptr, Tpad, gmin, G,
                 nullptr, nullptr, nullptr, 0, uc, qc, s);
}

void launch_knn_filter(const KnnPlan &p, const double *Fw, const double *fnorm, const double *Qp,
                       const double *thr, int64_t Tpad, void *pool, unsigned int *pool_ctl,
                       int *chunk_fill, int max_chunks, const int32_t *uc, const int32_t *qc,
                       hipStream_t s)
{
    launch_sweep(p, 1, 1, p.row_limit, p.n_slabs, 16 * p.nt, 16, Fw, fnorm, Qp, thr, Tpad, nullptr, 0,
                 pool, pool_ctl, chunk_fill, max_chunks, uc, qc, s);
}

// one launch instead of five memsets: list counters, status word, pool control, slab dispensers
// (+ up to three more regions of words to clear: control words and masks of the passes that follow -- each used to be a
// hipMemsetAsync of its own, a dispatch and a gap on the K-NN stream per call)
struct ResetExtra { unsigned int *p[3]; long long n[3]; };
__global__ void knn_reset_kernel(int *cnt, int64_t Tpad, int *status, unsigned int *pool_ctl,
                                 unsigned int *slab_counter, int *chunk_fill, int max_chunks, ResetExtra ex)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Tpad) cnt[i] = 0;
    if (i < max_chunks) chunk_fill[i] = 0;
    if (i == 0) { *status = 0; pool_ctl[0] = 0; pool_ctl[1] = 0; slab_counter[0] = 0; slab_counter[1] = 0; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
        for (long long k = i; k < ex.n[r]; k += (long long)gridDim.x * blockDim.x) ex.p[r][k] = 0u;
}

void launch_knn_reset(int *cnt, int64_t Tpad, int *status, unsigned int *pool_ctl,
                      unsigned int *slab_counter, int *chunk_fill, int max_chunks, hipStream_t s,
                      unsigned int *x0, long long n0, unsigned int *x1, long long n1, unsigned int *x2, long long n2)
{
    const int64_t n = Tpad > max_chunks ? Tpad : max_chunks;
    ResetExtra ex{{x0, x1, x2}, {x0 ? n0 : 0, x1 ? n1 : 0, x2 ? n2 : 0}};
    hipLaunchKernelGGL(knn_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cnt, Tpad, status,
                       pool_ctl, slab_counter, chunk_fill, max_chunks, ex);
}

int knn_pool_chunk_entries() { return POOL_CHUNK; }

size_t knn_pool_bytes(int max_chunks) { return (size_t)max_chunks * POOL_CHUNK * sizeof(PoolEntry); }

// ---------------------------------------------------------------------------
// bucket: entry pool -> per-row lists.  One workgroup per chunk (grid-stride); the returning
// atomics on the per-row counters are latency-hidden by thousands of threads in flight.
// status bit 0: a row list overflowed `cap`; bit 2: the entry pool overflowed (entries of ARBITRARY rows
// were dropped by the sweep: no row of the call can be trusted, whatever its own list length says).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
knn_bucket_kernel(const PoolEntry *__restrict__ pool, const unsigned int *__restrict__ pool_ctl,
                  const int *__restrict__ chunk_fill, int max_chunks, int Tpad, int n_valid, int *__restrict__ cnt,
                  double *__restrict__ lkey, int *__restrict__ lidx, int cap, int *__restrict__ status,
                  const int32_t *__restrict__ perm, int keys_f32)
{
    // per chunk: LDS histogram by row -> one global atomic per (chunk, row) reserves a run of
    // list slots -> entries of one row land contiguously.
    // The histogram holds 16-bit counters, two rows per word (a chunk has 2 048 entries; 32-bit atomics add 1 << 16 for the odd
    // row of a word and never carry): 64 KB for the 32 768 rows of a sharded search's calls instead of 128 -- two workgroups per
    // compute unit hide each other's atomics' round trips.  The run's base (32 bits) does not fit the counter: the row's first
    // entry keeps it in `base[its own index]` and leaves that index in the row's counter for the others.
    extern __shared__ int bsm[];
    unsigned int *hist = reinterpret_cast<unsigned int *>(bsm);                         // [(Tpad + 1) / 2]
    unsigned short *hist16 = reinterpret_cast<unsigned short *>(bsm);                   // the same, per row
    int *base = bsm + (Tpad + 1) / 2;                                                   // [POOL_CHUNK]
    int used = (int)pool_ctl[0];
    if (used > max_chunks) used = max_chunks;
    if (blockIdx.x == 0 && threadIdx.x == 0 && pool_ctl[1]) atomicOr(status, 4);
    constexpr int EPT = POOL_CHUNK / 256;
    static_assert(POOL_CHUNK < 65536, "entry indices and counts of a chunk travel in 16 bits");
    // the histogram is cleared ONCE; a chunk touches the slots of the rows it holds entries of and puts them back to zero
    // (clearing and scanning all Tpad slots per chunk made the kernel's time proportional to chunks x rows: 2.2 ms per
    // sharded step of 153 600 rows whatever the lists held)
    for (int i = threadIdx.x; i < (Tpad + 1) / 2; i += 256) hist[i] = 0u;
    __syncthreads();
    for (int c = blockIdx.x; c < used; c += gridDim.x) {
        const int n = chunk_fill[c];
        if (n == 0) continue;
        PoolEntry en[EPT];
        int rank[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = threadIdx.x + k * 256;
            rank[k] = -1;
            if (e < n) {
                en[k] = pool[(int64_t)c * POOL_CHUNK + e];
                if (en[k].idx < n_valid) {                                              // padding units never count
                    const int sh = 16 * (en[k].row & 1);
                    rank[k] = (int)((atomicAdd(&hist[en[k].row >> 1], 1u << sh) >> sh) & 0xffffu);
                }
            }
        }
        __syncthreads();
        // the first entry of a row in this chunk reserves the row's run of list slots
#pragma unroll
        for (int k = 0; k < EPT; ++k)
            if (rank[k] == 0) {
                const int e = threadIdx.x + k * 256;
                const int h = (int)hist16[en[k].row];
                base[e] = atomicAdd(&cnt[en[k].row], h);
                hist16[en[k].row] = (unsigned short)e;
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = threadIdx.x + k * 256;
            if (e < n && rank[k] >= 0) {
                const int slot = base[hist16[en[k].row]] + rank[k];
                if (slot < cap) {
                    // (keys_f32: the entry's key field holds the prefilter's float32 key in its low half -- knn16_kernels.hip PoolEntry16)
                    lkey[(int64_t)en[k].row * cap + slot] =
                        keys_f32 ? (double)__uint_as_float((unsigned int)(unsigned long long)__double_as_longlong(en[k].key)) : en[k].key;
                    // (a reordered operand -- kmeans_kernels.hip -- hands out positions: everything behind this line sees unit ids)
                    lidx[(int64_t)en[k].row * cap + slot] = perm ? perm[en[k].idx] : en[k].idx;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EPT; ++k)
            if (rank[k] == 0) hist16[en[k].row] = 0;
        __syncthreads();
    }
}

void launch_knn_bucket(const void *pool, const unsigned int *pool_ctl, const int *chunk_fill,
                       int max_chunks, int64_t Tpad, int64_t n_valid, int *cnt, double *lkey, int *lidx,
                       int cap, int *status, hipStream_t s, const int32_t *perm, bool keys_f32)
{
    static size_t attr[32] = {0};
    const size_t lds = (size_t)((Tpad + 1) / 2) * sizeof(int) + (size_t)POOL_CHUNK * sizeof(int);
    if (lds > 65536)
        lds_attr_ensure(attr, lds, [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_bucket_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    // as many workgroups per compute unit as its LDS holds (at most 6: 1 536 threads)
    int per_cu = (int)((size_t)(160 * 1024) / (lds + 1024));
    per_cu = per_cu < 1 ? 1 : per_cu > 6 ? 6 : per_cu;
    hipLaunchKernelGGL(knn_bucket_kernel, dim3(256 * per_cu), dim3(256), lds, s,
                       reinterpret_cast<const PoolEntry *>(pool), pool_ctl, chunk_fill, max_chunks,
                       (int)Tpad, (int)n_valid, cnt, lkey, lidx, cap, status, perm, keys_f32 ? 1 : 0);
}

// ---------------------------------------------------------------------------
// Row-sharded databases, second shared bound.  The bound the shards filter against comes from a sample and lets
// ~19 K candidates per row through, spread over the shards; every shard would re-rank its part exactly although
// the owner needs K in all.  Once a shard's list is there it knows better: the K-th smallest key of ITS list (plus
// the key error) bounds the K-th nearest key of the whole database from above, the minimum of those over the shards
// (one all-reduce) does too, and everything above it (plus the key error again) can go before the exact re-rank.
//   knn_local_kth_kernel: one wavefront per row; an upper bound of the K-th smallest list key from a 64-bin
//                         histogram (the upper edge of the bin the K-th key falls into), +eps[row]; DBL_MAX when the
//                         list holds fewer than K entries.
//   knn_list_prune_kernel: keeps the entries with key <= bound[row] + eps[row], in place, order kept.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
knn_local_kth_kernel(const int *__restrict__ cnt, const double *__restrict__ lkey, int cap, int K,
                     const double *__restrict__ eps, int64_t T, double *__restrict__ kth)
{
    __shared__ int hist[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;
    if (row >= T) return;
    int n = cnt[row];
    if (n > cap) n = cap;
    if (n < K) { if (lane == 0) kth[row] = DBL_MAX; return; }
    const double *key = lkey + row * cap;
    double kmin = DBL_MAX, kmax = -DBL_MAX;
    for (int i = lane; i < n; i += 64) { const double v = key[i]; kmin = fmin(kmin, v); kmax = fmax(kmax, v); }
    for (int off = 32; off > 0; off >>= 1) {
        kmin = fmin(kmin, __shfl_xor(kmin, off));
        kmax = fmax(kmax, __shfl_xor(kmax, off));
    }
    hist[wv][lane] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // one wavefront owns hist[wv]: program order is enough
    const double scale = (kmax > kmin) ? 64.0 / (kmax - kmin) : 0.0;
    for (int i = lane; i < n; i += 64) {
        int b = (int)((key[i] - kmin) * scale);
        b = b > 63 ? 63 : b;
        atomicAdd(&hist[wv][b], 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // inclusive prefix over the 64 bins (one per lane)
    int c = hist[wv][lane];
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(c, off);
        if (lane >= off) c += o;
    }
    const unsigned long long reach = __ballot(c >= K);          // bins at or past the K-th key
    const int b = __ffsll((long long)reach) - 1;
    if (lane == 0) {
        // every key of bins 0..b is below kmin + (b + 1) / scale (+ the rounding of the bin index: one more ulp-sized step)
        const double edge = (scale > 0.0 && b < 63) ? kmin + ((double)(b + 1) / scale) * (1.0 + 1e-12) + 1e-300 : kmax;
        kth[row] = edge + eps[row];
    }
}

__global__ void __launch_bounds__(256)
knn_list_prune_kernel(int *__restrict__ cnt, double *__restrict__ lkey, int *__restrict__ lidx, int cap,
                      const double *__restrict__ bound, const double *__restrict__ eps, int64_t T)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const int n_all = cnt[row];
    if (n_all > cap) return;                                    // overflowed list: left to the status path of finalize
    const double lim = bound[row];
    if (lim >= 0.5 * DBL_MAX) return;
    const double cut = lim + eps[row];
    double *key = lkey + row * cap;
    int *idx = lidx + row * cap;
    int w = 0;
    for (int i0 = 0; i0 < n_all; i0 += 64) {
        const int i = i0 + lane;
        double v = 0.0;
        int id = 0;
        const bool in = i < n_all;
        if (in) { v = key[i]; id = idx[i]; }
        const bool keep = in && v <= cut;
        const unsigned long long m = __ballot(keep);
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        // (the 64 entries of this round are in registers; the write positions lie below the round's own start + 64)
        if (keep) { key[w + rank] = v; idx[w + rank] = id; }
        w += __popcll(m);
    }
    if (lane == 0) cnt[row] = w;
}

void launch_knn_local_kth(const int *cnt, const double *lkey, int cap, int K, const double *eps, int64_t T, double *kth,
                          hipStream_t s)
{
    hipLaunchKernelGGL(knn_local_kth_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, s, cnt, lkey, cap, K, eps, T, kth);
}

void launch_knn_list_prune(int *cnt, double *lkey, int *lidx, int cap, const double *bound, const double *eps, int64_t T,
                           hipStream_t s)
{
    hipLaunchKernelGGL(knn_list_prune_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, s, cnt, lkey, lidx, cap, bound,
                       eps, T);
}

// ---------------------------------------------------------------------------
// query preparation
// ---------------------------------------------------------------------------
__global__ void prepare_queries_kernel(const double *__restrict__ Q, int64_t T, int D,
                                       double *__restrict__ Qp, double *__restrict__ Qf,
                                       double *__restrict__ qnorm, int64_t Tpad, int Dpad)
{
    const int64_t row = blockIdx.x;
    const int c = threadIdx.x;
    __shared__ double sq[256];
    double v = 0.0;
    if (row < T && c < D) v = Q[row * D + c];
    if (c < Dpad) {
        Qp[row * Dpad + c] = v;                       // row-major copy (exact re-rank)
        // fragment-order copy for the sweep: element c of row (tile, r16) belongs to chunk ch,
        // lane group g, k-step s: lane = g*16 + r16, pair = s/2, slot = s%2
        const int64_t tile = row >> 4;
        const int r16 = (int)(row & 15);
        const int ch = c >> 6, g = (c >> 4) & 3, sidx = c & 15;
        const int lane = g * 16 + r16;
        Qf[tile * (16 * Dpad) + ((int64_t)(ch * 8 + (sidx >> 1)) * 64 + lane) * 2 + (sidx & 1)] = v;
    }
    sq[c] = v * v;
    __syncthreads();
    if (c == 0) {
        double acc = 0.0;
        for (int i = 0; i < Dpad; ++i) acc += sq[i];
        qnorm[row] = acc;
    }
}

// snk_set_column_selection: target columns that are selected out are zeroed in the uploaded query rows
// (their database weights are zero, so the column then adds exactly +0.0 to every squared distance)
__global__ void mask_columns_kernel(double *__restrict__ Q, int64_t n, int D, const double *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && mask[i % D] == 0.0) Q[i] = 0.0;
}

void launch_mask_columns(double *Q, int64_t T, int D, const double *mask, hipStream_t s)
{
    const int64_t n = T * D;
    if (n > 0) hipLaunchKernelGGL(mask_columns_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Q, n, D, mask);
}

// rows wider than the sweeps take (Dpad > 256, see api_knn.hip knn_device): only the zero-padded row-major copy that
// the canonical-distance kernels read, and the squared norm
__global__ void prepare_queries_wide_kernel(const double *__restrict__ Q, int64_t T, int D, double *__restrict__ Qp,
                                            double *__restrict__ qnorm, int Dpad)
{
    const int64_t row = blockIdx.x;
    __shared__ double sq[256];
    double acc = 0.0;
    for (int c = threadIdx.x; c < Dpad; c += 256) {
        double v = 0.0;
        if (row < T && c < D) v = Q[row * D + c];
        Qp[row * Dpad + c] = v;
        acc += v * v;
    }
    sq[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 256; ++i) t += sq[i];
        qnorm[row] = t;
    }
}

void launch_prepare_queries(const double *Q, int64_t T, int D, double *Qp, double *Qf, double *qnorm,
                            int64_t Tpad, int Dpad, hipStream_t s)
{
    if (Dpad > 256) {
        hipLaunchKernelGGL(prepare_queries_wide_kernel, dim3((unsigned)Tpad), dim3(256), 0, s, Q, T, D, Qp, qnorm, Dpad);
        return;
    }
    hipLaunchKernelGGL(prepare_queries_kernel, dim3((unsigned)Tpad), dim3(256), 0, s, Q, T, D, Qp, Qf,
                       qnorm, Tpad, Dpad);
}

// ---------------------------------------------------------------------------
// bitonic helpers on (key, idx) pairs in LDS, ascending by (key, idx)
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool pair_less(double ka, int ia, double kb, int ib)
{
    return (ka < kb) || (ka == kb && ia < ib);
}

template <typename KeyT>
__device__ void bitonic_sort_pairs(KeyT *key, int *idx, int P)
{
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = ((i & k) == 0);
                    const KeyT ka = key[i], kb = key[ixj];
                    const int ia = idx[i], ib = idx[ixj];
                    const bool sw = up ? pair_less(kb, ib, ka, ia) : pair_less(ka, ia, kb, ib);
                    if (sw) { key[i] = kb; key[ixj] = ka; idx[i] = ib; idx[ixj] = ia; }
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// threshold: EXACT K-th smallest of the G group minima of one row, by an 8-bit MSB-first
// radix select on the order-preserving integer image of the float64 keys (8 passes over the
// row's minima, which sit in L2; 256-bin histogram in LDS per pass).
// Each group is one (slab, lane) pair = NT database rows, so at most NT*K database rows have a
// key <= the returned threshold when every row was visited (the overflow fallback relies on it).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long f64_sortable(double v)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_unsortable(unsigned long long u)
{
    u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
    return __longlong_as_double((long long)u);
}

#define THR_FOLD 512
#define THR_COLL 4096
__global__ void __launch_bounds__(256)
knn_threshold_kernel(const double *__restrict__ gmin, int64_t G, int64_t T, int K,
                     double *__restrict__ thr, int keep_min)
{
    __shared__ double key[THR_COLL];
    __shared__ int idx[THR_COLL];
    __shared__ int hist[256];
    __shared__ unsigned long long prefix_s;
    __shared__ int remaining_s, n_coll;
    __shared__ double bound_s;
    const int64_t row = blockIdx.x;
    if (row >= T) {                      // padding rows never pass
        if (threadIdx.x == 0) thr[row] = -DBL_MAX;
        return;
    }
    const double *src = gmin + row * G;
    if (G < K) {                         // fewer than K groups: accept everything
        if (threadIdx.x == 0) thr[row] = DBL_MAX;
        return;
    }
    double result;
    bool done = false;
    if (K <= THR_FOLD) {
        // fast exact path: fold to THR_FOLD bins by min; bins[K-1] bounds the answer and at most
        // K bins (K*ceil(G/THR_FOLD) values) lie at or below it; collect those and sort them
        for (int i = threadIdx.x; i < THR_FOLD; i += blockDim.x) { key[i] = DBL_MAX; idx[i] = i; }
        if (threadIdx.x == 0) n_coll = 0;
        __syncthreads();
        for (int64_t i0 = 0; i0 < G; i0 += THR_FOLD)
            for (int i = threadIdx.x; i < THR_FOLD && i0 + i < G; i += blockDim.x)
                key[i] = fmin(key[i], src[i0 + i]);
        __syncthreads();
        bitonic_sort_pairs(key, idx, THR_FOLD);
        if (threadIdx.x == 0) bound_s = key[K - 1];
        __syncthreads();
        const double bound = bound_s;
        __syncthreads();
        for (int64_t i = threadIdx.x; i < G; i += blockDim.x) {
            const double v = src[i];
            if (v <= bound) {
                const int slot = atomicAdd(&n_coll, 1);
                if (slot < THR_COLL) { key[slot] = v; idx[slot] = slot; }
            }
        }
        __syncthreads();
        const int n = n_coll;
        if (n <= THR_COLL) {
            int P = 2;
            while (P < n) P <<= 1;
            for (int i = n + threadIdx.x; i < P; i += blockDim.x) { key[i] = DBL_MAX; idx[i] = i; }
            __syncthreads();
            bitonic_sort_pairs(key, idx, P);
            result = key[K - 1];
            done = true;
        }
        __syncthreads();
    }
    if (!done) {
        // general exact path: 8-bit MSB-first radix select on the order-preserving integer image
        if (threadIdx.x == 0) { prefix_s = 0ull; remaining_s = K; }
        __syncthreads();
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            hist[threadIdx.x] = 0;
            __syncthreads();
            const unsigned long long prefix = prefix_s;
            const unsigned long long mask = pass ? (~0ull << (shift + 8)) : 0ull;
            for (int64_t i = threadIdx.x; i < G; i += blockDim.x) {
                const unsigned long long u = f64_sortable(src[i]);
                if ((u & mask) == prefix) atomicAdd(&hist[(int)((u >> shift) & 0xff)], 1);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int rem = remaining_s, bin = 0;
                for (; bin < 256; ++bin) {
                    if (hist[bin] >= rem) break;
                    rem -= hist[bin];
                }
                prefix_s = prefix | ((unsigned long long)bin << shift);
                remaining_s = rem;
            }
            __syncthreads();
        }
        result = f64_unsortable(prefix_s);
    }
    if (threadIdx.x == 0) {
        double v = result;
        if (v < DBL_MAX) v = v + fabs(v) * 1e-13 + 1e-300;
        if (keep_min) v = fmin(v, thr[row]);
        thr[row] = v;
    }
}

void launch_knn_threshold(const double *gmin, int64_t G, int64_t T, int64_t Tpad, int K,
                          double *thr, int keep_min, hipStream_t s)
{
    hipLaunchKernelGGL(knn_threshold_kernel, dim3((unsigned)Tpad), dim3(256), 0, s, gmin, G, T, K,
                       thr, keep_min);
}

__global__ void fill_threshold_kernel(double *thr, int64_t T, int64_t Tpad, double value)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Tpad) thr[i] = (i < T) ? value : -DBL_MAX;
}

void launch_fill_threshold(double *thr, int64_t T, int64_t Tpad, double value, hipStream_t s)
{
    hipLaunchKernelGGL(fill_threshold_kernel, dim3((unsigned)((Tpad + 255) / 256)), dim3(256), 0, s,
                       thr, T, Tpad, value);
}

// ---------------------------------------------------------------------------
// finalize: sort the row's candidate list, exact re-rank, output
// ---------------------------------------------------------------------------
// inclusive prefix sum over the 256 threads of a workgroup (every thread must call it)
__device__ __forceinline__ int block_incl_scan_256(int v, int *wsum /* LDS [4] */)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int c = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(c, off);
        if (lane >= off) c += o;
    }
    if (lane == 63) wsum[wv] = c;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    return c + base;
}

#define SEL_MAX 2048     // candidates re-ranked exactly per row (K + near ties / key error margin)
#define FIN_SMALL 512    // lists up to this length go through the small-LDS instance (13 workgroups per compute unit
                         // instead of 2: the short lists of a row-sharded database are latency-, not LDS-bound)

// CLASS 0: every row; 1: rows with at most FIN_SMALL entries (the others are left to CLASS 2); 2: the longer ones
// F32K: the list keys are float32 values (the matrix prefilter's) and are kept as such in LDS; together with the exact
// keys living where the list was (it is dead by then) a workgroup needs 47 KB instead of 80 and three fit a compute unit
// LEAN: only the float32 keys of the list live in LDS (16 KB for lists of up to 4 096 entries instead of 32: six workgroups per
// compute unit instead of three -- a row is a 40 us critical path, so the kernel's time goes with the occupancy); the unit ids
// of the selected entries are read from the list in global memory.  Rows the value-binned selection cannot serve (short lists,
// one-valued lists, more near ties than the selection holds) are flagged in `retry` and left to a second launch of the full form.
// SELX > 0: the instance for the rows that hold more near ties than SEL_MAX (retry[row] == 2, left by the full form when the
// launch has a big tier): a selection of SELX entries -- every list fits -- on 97 KB of LDS, for the few rows of a voice whose
// units are so dense in key space that thousands of them lie inside the key margin (12 M units of SURVEY 8d's walk).
template <int CLASS, bool F32K, bool LEAN = false, int SELX = 0>
__global__ void __launch_bounds__(256)
knn_finalize_kernel(const double *__restrict__ Fw, const float *__restrict__ F_unw, int Fp,
                    const double *__restrict__ wt, int Dpad, int D, const double *__restrict__ Qp,
                    const double *__restrict__ qnorm, int64_t T, int K,
                    const int *__restrict__ cnt, const double *__restrict__ lkey,
                    const int *__restrict__ lidx, int cap, int64_t id_offset,
                    const double *__restrict__ eps, const double *__restrict__ fnorm, double eps_c,
                    const double *__restrict__ cq,
                    int64_t *__restrict__ cand, double *__restrict__ dist,
                    double *__restrict__ d2_out, int *__restrict__ status, int *__restrict__ rowflag,
                    const double *__restrict__ thr, unsigned int *__restrict__ margin_stat, int *__restrict__ retry,
                    int big_tier)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int SELM = SELX ? SELX : (CLASS == 1) ? FIN_SMALL : SEL_MAX;
    const int64_t row = blockIdx.x;
    if (!LEAN && retry && (SELX ? retry[row] != 2 : retry[row] != 1)) return;     // second / third launch: only the rows the form before left
    const int n_all = cnt[row];
    if (CLASS == 1 && n_all > FIN_SMALL) return;
    if (CLASS == 2 && n_all <= FIN_SMALL) return;
    if (rowflag && threadIdx.x == 0) rowflag[row] = 0;
    if (n_all > cap) {                    // list overflowed: host re-tightens and retries
        if (threadIdx.x == 0) { atomicOr(status, 1); if (rowflag) rowflag[row] = 1; }
        return;
    }
    const int n = n_all;
    const bool verify = (big_tier & 2) != 0;        // optimistic thresholds (api_knn.hip): the list is proven to hold the K nearest, or the row is flagged
    if (n == 0) {
        if (verify && threadIdx.x == 0) atomicOr(status, 8);
        // nothing of this row lives here (a shard whose units are all beyond the shared bound): K padding entries
        for (int j = threadIdx.x; j < K; j += blockDim.x) {
            if (cand) cand[row * K + j] = -1;
            if (dist) dist[row * K + j] = SNK_VERY_BIG;
            if (d2_out) d2_out[row * K + j] = SNK_VERY_BIG * SNK_VERY_BIG;
        }
        return;
    }
    int P = 2;
    while (P < n) P <<= 1;
    using KeyT = typename std::conditional<F32K, float, double>::type;
    KeyT *key = reinterpret_cast<KeyT *>(smem);
    int *idx = reinterpret_cast<int *>(smem + (size_t)P * sizeof(KeyT));
    double *ex_key = reinterpret_cast<double *>(smem);      // over key / idx: only written once the selection is in ex_idx
    __shared__ int ex_idx[SELM];
    __shared__ int n_sel_s, hist[256], cut_bin_s, pre_s[256], wsum_s[4];
    __shared__ double red_min[4], red_max[4];
    const int kk = K < n ? K : n;         // entries that can be returned

    // ---- fast path: value-binned selection.  256 linear bins between the smallest and the
    // largest key of the list; every entry in the bins up to (and one past) the bin holding the
    // K-th smallest key is re-ranked exactly.  Falls back to the full sort when that set does
    // not fit SELM (massive ties).
    double kmin = DBL_MAX, kmax = -DBL_MAX;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = lkey[row * cap + i];
        key[i] = (KeyT)v;
        if (!LEAN) idx[i] = lidx[row * cap + i];
        kmin = fmin(kmin, v); kmax = fmax(kmax, v);
        // (the margin below used the largest ||f||^2 among THIS row's survivors: a dependent 8-byte gather per list entry,
        // 16 M of them per B* launch; the row's eps -- the same bound with the largest norm of the database -- is what the
        // filter assumed anyway)
    }
    // (wavefront minima / maxima by shuffles, the four wavefronts' through LDS: two barriers instead of nine)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = fmin(kmin, __shfl_xor(kmin, off, 64));
        kmax = fmax(kmax, __shfl_xor(kmax, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { red_min[threadIdx.x >> 6] = kmin; red_max[threadIdx.x >> 6] = kmax; }
    hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) n_sel_s = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        red_min[0] = fmin(fmin(red_min[0], red_min[1]), fmin(red_min[2], red_min[3]));
        red_max[0] = fmax(fmax(red_max[0], red_max[1]), fmax(red_max[2], red_max[3]));
    }
    __syncthreads();
    kmin = red_min[0]; kmax = red_max[0];
    // keys from the f32 prefilter are only good to +-e_i = c (2 |q| |f_i| + |f_i|^2): every selection
    // margin is widened by 2 max_i e_i over the survivors of THIS row (their norms, not the largest
    // norm of the whole database that the filter threshold has to assume)
    double margin = 0.0;
    if (eps) margin = 2.0 * eps[row];         // |key~ - key| <= eps[row] for every unit of the database (prepare_queries16[b])
    const double scale = (kmax > kmin) ? 256.0 / (kmax - kmin) : 0.0;
    // (lists of optimistic thresholds are a few K long: the selection serves them from K + K / 2 entries on)
    bool fast = (n > K + K / 2 && n > 256) && (scale > 0.0) && (kk == K);
    if (fast) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            int b = (int)(((double)key[i] - kmin) * scale);
            b = b > 255 ? 255 : b;
            atomicAdd(&hist[b], 1);
        }
        __syncthreads();
        // prefix sums of the 256 bins by the whole workgroup (a thread-0 loop over the bins, twice, was most of the
        // kernel's time on lists of ~2000 entries: ~500 dependent LDS reads per row)
        int *pre = pre_s, *wsum = wsum_s;
        {
            const int mine = hist[threadIdx.x];
            const int incl = block_incl_scan_256(mine, wsum);
            pre[threadIdx.x] = incl;
            if (incl >= K && incl - mine < K) cut_bin_s = (int)threadIdx.x;   // the bin holding the K-th smallest key
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int b = cut_bin_s;
            int cut = b + 1;                       // one bin past the K-th key's bin (near ties)
            if (margin > 0.0) {                    // ... and everything within the key error margin
                const double upper = kmin + (double)(b + 1) / scale + margin;
                const int cm = (int)((upper - kmin) * scale) + 1;
                if (cm > cut) cut = cm;
            }
            if (cut > 255) cut = 255;
            cut_bin_s = (pre[cut] <= SELM) ? cut : -1;
        }
        __syncthreads();
        fast = cut_bin_s >= 0;
    }
    if (LEAN && !fast) {                  // uniform: the full form takes this row
        if (threadIdx.x == 0) retry[row] = 1;
        return;
    }
    if (fast) {
        const int cut = cut_bin_s;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            int b = (int)(((double)key[i] - kmin) * scale);
            b = b > 255 ? 255 : b;
            if (b <= cut) {
                const int slot = atomicAdd(&n_sel_s, 1);
                ex_idx[slot] = LEAN ? lidx[row * cap + i] : idx[i];
            }
        }
        __syncthreads();
    } else {
        for (int i = n + threadIdx.x; i < P; i += blockDim.x) { key[i] = F32K ? (KeyT)FLT_MAX : (KeyT)DBL_MAX; idx[i] = 0x7fffffff; }
        __syncthreads();
        bitonic_sort_pairs(key, idx, P);
        if (threadIdx.x == 0) {
            int ns = kk;
            if (kk > 0 && kk < n) {
                // GEMM-form keys carry ~1e-13 relative error: re-rank every candidate within a
                // safety margin of the K-th key so that the exact order decides
                const double kth = (double)key[kk - 1];
                const double delta = 1e-10 * (fabs(kth) + qnorm[row] + 1.0) + margin;
                while (ns < n && ns < SELM && (double)key[ns] <= kth + delta) ++ns;
                if (ns == SELM && ns < n && (double)key[ns] <= kth + delta) {
                    if (!SELX && (big_tier & 1) && retry) { retry[row] = 2; ns = -1; }      // more near ties than this form holds: the big tier's row
                    else { atomicOr(status, 2); if (rowflag) rowflag[row] = 2; }
                }
            }
            n_sel_s = ns;
        }
        __syncthreads();
        if (n_sel_s < 0) return;                                                  // uniform
        for (int e = threadIdx.x; e < n_sel_s; e += blockDim.x) ex_idx[e] = idx[e];
        __syncthreads();
    }
    const int n_sel = n_sel_s;
    // what the kernel's roofline is priced on (snk_get_info: finalize_*) -- only while option roofline_counters is on (word 6):
    // 9 600 workgroups adding to ONE address serialise in L2 (+ 0.3 ms per launch when always on: profiles/r06_a)
    if (margin_stat && threadIdx.x == 0 && margin_stat[6]) {
        atomicAdd(reinterpret_cast<unsigned long long *>(margin_stat) + 1, (unsigned long long)n);
        atomicAdd(reinterpret_cast<unsigned long long *>(margin_stat) + 2, (unsigned long long)n_sel);
    }
    // exact squared distance in the canonical order: acc = acc + (q_c - f_c)*(q_c - f_c),
    // c ascending, separately rounded sub / mul / add (bit-identical to the oracle)
    int SP = 256;                         // sort size: the selection, padded to a power of two
    while (SP < n_sel || SP < K) SP <<= 1;
    for (int e = threadIdx.x; e < SP; e += blockDim.x) {
        double acc = DBL_MAX;
        int id = 0x7fffffff;
        if (e < n_sel) {
            id = ex_idx[e];
            const double *q = Qp + row * Dpad;
            acc = 0.0;
            if (F_unw) {
                // the weighted row is fl64(f32 * w): recomputed from the float32 row (half the gathered
                // bytes of the float64 copy, 16-byte loads); the same bits
                const float4 *f4 = reinterpret_cast<const float4 *>(F_unw + (int64_t)id * Fp);
                for (int c4 = 0; 4 * c4 < D; ++c4) {
                    const float4 v = f4[c4];
                    const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = 4 * c4 + i;
                        if (c < D) {
                            const double d = __dsub_rn(q[c], __dmul_rn((double)xs[i], wt[c]));
                            acc = __dadd_rn(acc, __dmul_rn(d, d));
                        }
                    }
                }
            } else {
                const double *f = Fw + (int64_t)id * Dpad;
                for (int c = 0; c < D; ++c) {
                    const double d = __dsub_rn(q[c], f[c]);
                    acc = __dadd_rn(acc, __dmul_rn(d, d));
                }
            }
        }
        ex_key[e] = acc;
        ex_idx[e] = id;
    }
    __syncthreads();
    // (rank-by-counting instead of the sort -- every entry counting the entries in front of it, no barriers -- was tried:
    // 0.45 -> 0.52 ms per 9 600 rows, the selections of 200-500 entries make it quadratic work on LDS reads)
    bitonic_sort_pairs(ex_key, ex_idx, SP);
    // Tripwire of the prefilter's key bound (include/snk.h: prefilter_margin_rows).  The filter kept every unit whose
    // approximate key lay under thr = (bound of the K-th nearest key) + eps, eps being the ASSUMED largest error of an
    // approximate key.  The exact K-th key is known now: room = thr - kth is what the keys of the true neighbours could
    // have been off by without being dropped.  Rows with room < 2 eps are the rows whose result would have been wrong
    // had the error assumption been violated by a factor of two; [1] keeps the smallest room / eps seen.
    // Optimistic thresholds (api_knn.hip: the filter's threshold came from FEWER than K sample minima -- an estimate, no bound):
    // the list holds every unit whose approximate key lies under thr.  If it has K entries and the exact K-th key k of the LIST
    // satisfies k + eps <= thr, every unit of the database with a true key <= k has an approximate key <= k + eps <= thr and is
    // in the list: the list's K nearest are the database's.  Accepted only with TWICE that room (k + 2 eps <= thr), so that a row
    // accepted here also passes the tripwire below -- a 2 x violation of the assumed eps would not have changed it; otherwise the
    // row is flagged and the caller redoes the call with guaranteed thresholds (status bit 8), where the tripwire judges it.
    bool flagged = false;
    if (verify && threadIdx.x == 0) {
        bool proven = kk == K && n >= K;
        if (proven && thr && eps && thr[row] < 0.5 * DBL_MAX) proven = (thr[row] - (ex_key[K - 1] - qnorm[row])) >= 2.0 * eps[row];
        if (!proven) { atomicOr(status, 8); flagged = true; }
    }
    if (thr && eps && margin_stat && threadIdx.x == 0 && kk == K && n > K && !flagged) {
        const double room = thr[row] - (ex_key[K - 1] - qnorm[row]);
        const double e = eps[row];
        if (e > 0.0 && thr[row] < 0.5 * DBL_MAX) {
            if (room < 2.0 * e) atomicAdd(&margin_stat[0], 1u);
            float ratio = (float)(room / e);
            if (!(ratio > 0.f)) ratio = 0.f;
            atomicMin(&margin_stat[1], __float_as_uint(ratio));
        }
    }
    for (int j = threadIdx.x; j < K; j += blockDim.x) {
        int64_t c = -1;
        double d2 = SNK_VERY_BIG * SNK_VERY_BIG, d = SNK_VERY_BIG;
        if (j < kk) { c = (int64_t)ex_idx[j] + id_offset; d2 = ex_key[j]; d = __dsqrt_rn(d2); }
        if (cand) cand[row * K + j] = c;
        if (dist) dist[row * K + j] = d;
        if (d2_out) d2_out[row * K + j] = d2;
    }
}

void launch_knn_finalize(const double *Fw, const float *F_unw, int Fp, const double *wt, int Dpad, int D, const double *Qp, const double *qnorm,
                         int64_t T, int K, const int *cnt, const double *lkey, const int *lidx,
                         int cap, int64_t id_offset, const double *eps, const double *fnorm, double eps_c, const double *cq,
                         int64_t *cand, double *dist, double *d2_out, int *status, int *rowflag, hipStream_t s, bool split_short,
                         const double *thr, unsigned int *margin_stat, int *retry, bool big_tier, bool verify_thr, bool retry_cleared)
{
    const int vf = verify_thr ? 2 : 0;                       // bit 1 of the kernels' big_tier word
    int P = 2;
    while (P < cap) P <<= 1;
    // prefilter keys (eps given) are float32 values: 8 bytes per list entry in LDS instead of 12; the exact keys of the
    // selection (SEL_MAX / FIN_SMALL doubles) live over the list
    const bool f32k = eps != nullptr;
    size_t shmem = (size_t)P * ((f32k ? sizeof(float) : sizeof(double)) + sizeof(int));
    if (shmem < (size_t)SEL_MAX * sizeof(double)) shmem = (size_t)SEL_MAX * sizeof(double);
    static size_t attr[32] = {0};
    lds_attr_ensure(attr, shmem, [&] {
#define SNK_FIN_ATTR(C_, F_) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_finalize_kernel<C_, F_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)
        SNK_FIN_ATTR(0, false); SNK_FIN_ATTR(0, true); SNK_FIN_ATTR(2, false); SNK_FIN_ATTR(2, true);
#undef SNK_FIN_ATTR
    });
#define SNK_FIN(C_, F_, SH_) hipLaunchKernelGGL((knn_finalize_kernel<C_, F_>), dim3((unsigned)T), dim3(256), SH_, s, Fw, F_unw, Fp, wt, Dpad, D, Qp, \
                       qnorm, T, K, cnt, lkey, lidx, cap, id_offset, eps, fnorm, eps_c, cq, cand, dist, d2_out, status, rowflag, thr, margin_stat, nullptr, vf)
    if (f32k && retry && !split_short && !rowflag) {
        // the lean form for (nearly) every row, the full form for the rows it flags
        size_t lean = (size_t)P * sizeof(float);
        if (lean < (size_t)SEL_MAX * sizeof(double)) lean = (size_t)SEL_MAX * sizeof(double);
        if (!retry_cleared) (void)hipMemsetAsync(retry, 0, (size_t)T * sizeof(int), s);
        hipLaunchKernelGGL((knn_finalize_kernel<0, true, true>), dim3((unsigned)T), dim3(256), lean, s, Fw, F_unw, Fp, wt, Dpad, D, Qp,
                           qnorm, T, K, cnt, lkey, lidx, cap, id_offset, eps, fnorm, eps_c, cq, cand, dist, d2_out, status, rowflag, thr, margin_stat, retry, vf);
        const int big = (big_tier && cap <= 8192) ? 1 : 0;
        hipLaunchKernelGGL((knn_finalize_kernel<0, true, false>), dim3((unsigned)T), dim3(256), shmem, s, Fw, F_unw, Fp, wt, Dpad, D, Qp,
                           qnorm, T, K, cnt, lkey, lidx, cap, id_offset, eps, fnorm, eps_c, cq, cand, dist, d2_out, status, rowflag, thr, margin_stat, retry, big | vf);
        if (big) {
            // third tier (a voice that has overflowed before: api_knn.hip knn_level): the rows with more near ties than SEL_MAX
            size_t shbig = (size_t)P * (sizeof(float) + sizeof(int));
            if (shbig < (size_t)8192 * sizeof(double)) shbig = (size_t)8192 * sizeof(double);
            static size_t attr_big[32] = {0};
            lds_attr_ensure(attr_big, shbig, [&] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_finalize_kernel<0, true, false, 8192>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shbig); });
            hipLaunchKernelGGL((knn_finalize_kernel<0, true, false, 8192>), dim3((unsigned)T), dim3(256), shbig, s, Fw, F_unw, Fp, wt, Dpad, D, Qp,
                               qnorm, T, K, cnt, lkey, lidx, cap, id_offset, eps, fnorm, eps_c, cq, cand, dist, d2_out, status, rowflag, thr, margin_stat, retry, vf);
        }
        return;
    }
    if (split_short && K <= FIN_SMALL / 2 && cap > FIN_SMALL) {
        const size_t small = (size_t)FIN_SMALL * ((f32k ? sizeof(float) : sizeof(double)) + sizeof(int)) > (size_t)FIN_SMALL * sizeof(double)
                                 ? (size_t)FIN_SMALL * ((f32k ? sizeof(float) : sizeof(double)) + sizeof(int)) : (size_t)FIN_SMALL * sizeof(double);
        if (f32k) { SNK_FIN(1, true, small); SNK_FIN(2, true, shmem); }
        else { SNK_FIN(1, false, small); SNK_FIN(2, false, shmem); }
        return;
    }
    if (f32k) SNK_FIN(0, true, shmem); else SNK_FIN(0, false, shmem);
#undef SNK_FIN
}

// ---------------------------------------------------------------------------
// Row-sharded databases: the exchange of the shards' lists, compacted.  Under the shared bound a shard keeps about K / G
// candidates per row (a row's neighbours mostly sit in one shard); the (rows, K) matrices the owners are sent were 86 % padding
// (215 MB per rank and B* step at G = 8).  A destination's block now is
//        [ counts: one byte per row, padded to 16 ][ d2: tot x 8 bytes ][ ids: tot x 8 bytes ]
// with the rows' valid entries (the lists are (distance, id)-ordered with the -1 padding at the end) back to back.
//   shard_count_kernel   valid entries per row
//   shard_scan_kernel    one workgroup per block: exclusive prefix of the counts (the row's first entry), the block's total
//   shard_pack_kernel    entries to their places in the send buffer
//   shard_unpack_kernel  the owner's side: a received block back into (rows, K) lists with padding, what merge_topk reads
// The block sizes have to be on the host for the transfers: one all-gather of the G totals and one device -> host copy per step.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
shard_count_kernel(const int64_t *__restrict__ ids, int64_t R, int K, unsigned char *__restrict__ cnt)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    int n = 0;
    while (n < K && ids[r * K + n] >= 0) ++n;
    cnt[r] = (unsigned char)n;
}

// block p: counts at cnt + cnt_off[p], rows[p] of them; offsets to off + off_off[p]; tot[p]
__global__ void __launch_bounds__(256)
shard_scan_kernel(const unsigned char *__restrict__ cnt, const int64_t *__restrict__ cnt_off, const int64_t *__restrict__ rows,
                  int *__restrict__ off, const int64_t *__restrict__ off_off, int64_t *__restrict__ tot)
{
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int p = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned char *c = cnt + cnt_off[p];
    int *o = off + off_off[p];
    const int64_t n = rows[p];
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < n; i0 += 256) {
        const int64_t i = i0 + threadIdx.x;
        const int v = i < n ? (int)c[i] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int x = __shfl_up(incl, d, 64);
            if (lane >= d) incl += x;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int base = carry_s;
        for (int w = 0; w < wv; ++w) base += wsum[w];
        if (i < n) o[i] = base + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[p] = carry_s;
}

// row r of the (R, K) lists belongs to destination block p = dest[r]'s ... found by the row ranges: row0[p] <= r < row0[p] + rows[p]
__global__ void __launch_bounds__(256)
shard_pack_kernel(const double *__restrict__ d2, const int64_t *__restrict__ ids, const unsigned char *__restrict__ cnt,
                  const int *__restrict__ off, const int64_t *__restrict__ row0, const int64_t *__restrict__ rows,
                  const int64_t *__restrict__ poff, const int64_t *__restrict__ tot, int G, int64_t R, int K,
                  unsigned char *__restrict__ out)
{
    // one wavefront per row
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    int p = 0;
    while (p + 1 < G && r >= row0[p] + rows[p]) ++p;
    const int64_t i = r - row0[p];
    unsigned char *blk = out + poff[p];
    const int64_t hdr = (rows[p] + 15) & ~(int64_t)15;
    const int n = cnt[r];
    if (lane == 0) blk[i] = (unsigned char)n;
    double *pd = reinterpret_cast<double *>(blk + hdr) + off[r];
    int64_t *pi = reinterpret_cast<int64_t *>(blk + hdr + tot[p] * 8) + off[r];
    for (int k = lane; k < n; k += 64) { pd[k] = d2[r * K + k]; pi[k] = ids[r * K + k]; }
}

// source block q: counts at in + roff[q] (r_own of them), entries behind them; offq[q * r_own + i] from shard_scan_kernel
__global__ void __launch_bounds__(256)
shard_unpack_kernel(const unsigned char *__restrict__ in, const int64_t *__restrict__ roff, const int64_t *__restrict__ totq,
                    const int *__restrict__ offq, int64_t r_own, int K, double *__restrict__ d2_out, int64_t *__restrict__ id_out)
{
    const int lane = threadIdx.x & 63, q = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= r_own) return;
    const unsigned char *blk = in + roff[q];
    const int64_t hdr = (r_own + 15) & ~(int64_t)15;
    int n = blk[i];
    if (n > K) n = K;
    const double *pd = reinterpret_cast<const double *>(blk + hdr) + offq[q * r_own + i];
    const int64_t *pi = reinterpret_cast<const int64_t *>(blk + hdr + totq[q] * 8) + offq[q * r_own + i];
    double *od = d2_out + ((int64_t)q * r_own + i) * K;
    int64_t *oi = id_out + ((int64_t)q * r_own + i) * K;
    for (int k = lane; k < K; k += 64) {
        od[k] = k < n ? pd[k] : SNK_VERY_BIG * SNK_VERY_BIG;
        oi[k] = k < n ? pi[k] : -1;
    }
}

void launch_shard_count(const int64_t *ids, int64_t R, int K, unsigned char *cnt, hipStream_t s)
{
    hipLaunchKernelGGL(shard_count_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, ids, R, K, cnt);
}
void launch_shard_scan(const unsigned char *cnt, const int64_t *cnt_off, const int64_t *rows, int *off, const int64_t *off_off,
                       int64_t *tot, int G, hipStream_t s)
{
    hipLaunchKernelGGL(shard_scan_kernel, dim3((unsigned)G), dim3(256), 0, s, cnt, cnt_off, rows, off, off_off, tot);
}
void launch_shard_pack(const double *d2, const int64_t *ids, const unsigned char *cnt, const int *off, const int64_t *row0,
                       const int64_t *rows, const int64_t *poff, const int64_t *tot, int G, int64_t R, int K, unsigned char *out,
                       hipStream_t s)
{
    hipLaunchKernelGGL(shard_pack_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, d2, ids, cnt, off, row0, rows, poff, tot, G, R, K, out);
}
void launch_shard_unpack(const unsigned char *in, const int64_t *roff, const int64_t *totq, const int *offq, int64_t r_own, int K, int G,
                         double *d2_out, int64_t *id_out, hipStream_t s)
{
    if (r_own < 1) return;
    hipLaunchKernelGGL(shard_unpack_kernel, dim3((unsigned)((r_own + 3) / 4), (unsigned)G), dim3(256), 0, s, in, roff, totq, offq, r_own, K,
                       d2_out, id_out);
}

// ---------------------------------------------------------------------------
// Last resort for rows the list pipeline cannot serve (more database units tied with -- or within
// rounding of -- the K-th neighbour than a candidate list holds: digital silence, mass duplicates):
// one workgroup per row computes the canonical distance to EVERY unit, radix-selects the K-th
// smallest value and takes the K smallest (distance, id) pairs, ties by lowest id.  Slow (a full
// float64 scan per row by one workgroup) and exact by construction.
//   scratch: n_rows x scratch_pitch doubles
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
knn_exact_rows_kernel(const double *__restrict__ Fw, int Dpad, int D, int64_t N, const double *__restrict__ Qp,
                      const int *__restrict__ rows, int K, double *__restrict__ scratch, int64_t scratch_pitch,
                      const int32_t *__restrict__ unit_class, const int32_t *__restrict__ query_class,
                      int64_t id_offset, int64_t *__restrict__ cand, double *__restrict__ dist,
                      double *__restrict__ d2_out)
{
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long prefix_s;
    __shared__ unsigned int want_s, n_less_s, taken_s, short_s, scan_s[1024];
    __shared__ double skey[256];
    __shared__ int sidx[256];
    const int tid = threadIdx.x;
    const int64_t row = rows[blockIdx.x];
    double *val = scratch + (int64_t)blockIdx.x * scratch_pitch;
    const double *q = Qp + row * Dpad;
    const int qc = query_class ? query_class[row] : 0;
    // 1. canonical squared distances (column by column, separately rounded sub / mul / add)
    for (int64_t i = tid; i < N; i += 1024) {
        double acc = __builtin_inf();
        if (!unit_class || unit_class[i] == qc) {
            const double *f = Fw + i * Dpad;
            acc = 0.0;
            for (int c = 0; c < D; ++c) {
                const double d = __dsub_rn(q[c], f[c]);
                acc = __dadd_rn(acc, __dmul_rn(d, d));
            }
        }
        val[i] = acc;
    }
    __syncthreads();
    // 2. radix select (8 x 8 bits; non-negative doubles order like their bit patterns): value of the
    //    K-th smallest entry and the number of entries strictly below it
    if (tid == 0) { prefix_s = 0ull; want_s = (unsigned int)K; n_less_s = 0u; short_s = 0u; }
    __syncthreads();
    for (int pass = 7; pass >= 0; --pass) {
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = prefix_s;
        const int shift = 8 * pass;
        for (int64_t i = tid; i < N; i += 1024) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(val[i]);
            if (pass == 7 || (b >> (shift + 8)) == (prefix >> (shift + 8)))
                atomicAdd(&hist[(b >> shift) & 255ull], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned int want = want_s, cum = 0u;
            int b = 0;
            for (; b < 256; ++b) { if (cum + hist[b] >= want) break; cum += hist[b]; }
            if (b == 256) { b = 255; short_s = 1u; }        // fewer than K entries in total (N < K)
            prefix_s = prefix | ((unsigned long long)b << shift);
            want_s = want - cum;
            n_less_s += cum;
        }
        __syncthreads();
    }
    const double vk = short_s ? __builtin_inf() : __longlong_as_double((long long)prefix_s);
    const unsigned int n_less = n_less_s < (unsigned int)K ? n_less_s : (unsigned int)K;   // entries strictly below vk
    const unsigned int n_tie = (unsigned int)K - n_less;     // how many entries equal to vk are taken (lowest ids)
    // 3. collect: everything below vk, then the ties in id order
    if (tid < 256) { skey[tid] = DBL_MAX; sidx[tid] = 0x7fffffff; }
    if (tid == 0) taken_s = 0u;
    __syncthreads();
    for (int64_t i = tid; i < N; i += 1024)
        if (val[i] < vk) {
            const unsigned int slot = atomicAdd(&taken_s, 1u);
            if (slot < 256u) { skey[slot] = val[i]; sidx[slot] = (int)i; }
        }
    __syncthreads();
    unsigned int tie_base = 0u;
    for (int64_t i0 = 0; i0 < N && tie_base < n_tie; i0 += 1024) {
        const int64_t i = i0 + tid;
        const unsigned int flag = (i < N && val[i] == vk && vk < __builtin_inf()) ? 1u : 0u;
        scan_s[tid] = flag;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {          // inclusive scan
            const unsigned int v = (tid >= off) ? scan_s[tid - off] : 0u;
            __syncthreads();
            scan_s[tid] += v;
            __syncthreads();
        }
        const unsigned int pos = tie_base + scan_s[tid] - flag;
        if (flag && pos < n_tie) { skey[n_less + pos] = vk; sidx[n_less + pos] = (int)i; }
        const unsigned int total = scan_s[1023];
        __syncthreads();
        tie_base += total;
    }
    __syncthreads();
    // 4. order by (distance, id) and write
    bitonic_sort_pairs(skey, sidx, 256);
    for (int j = tid; j < K; j += 1024) {
        int64_t c = -1;
        double d2 = SNK_VERY_BIG * SNK_VERY_BIG, d = SNK_VERY_BIG;
        if (skey[j] < __builtin_inf() && sidx[j] != 0x7fffffff) { c = (int64_t)sidx[j] + id_offset; d2 = skey[j]; d = __dsqrt_rn(d2); }
        if (cand) cand[row * K + j] = c;
        if (dist) dist[row * K + j] = d;
        if (d2_out) d2_out[row * K + j] = d2;
    }
}

void launch_knn_exact_rows(const double *Fw, int Dpad, int D, int64_t N, const double *Qp, const int *rows,
                           int n_rows, int K, double *scratch, int64_t scratch_pitch, const int32_t *unit_class,
                           const int32_t *query_class, int64_t id_offset, int64_t *cand, double *dist,
                           double *d2_out, hipStream_t s)
{
    hipLaunchKernelGGL(knn_exact_rows_kernel, dim3((unsigned)n_rows), dim3(1024), 0, s, Fw, Dpad, D, N, Qp, rows, K,
                       scratch, scratch_pitch, unit_class, query_class, id_offset, cand, dist, d2_out);
}

// ---------------------------------------------------------------------------
// distances of GIVEN candidates (quinphone preselection, synth_halfphone.py:1343-1349):
// dist[t,k] = ||F[cand[t,k]] - q_t||, canonical order; a negative id indexes from the end of the
// database exactly like the reference's numpy fancy indexing (padding -1 -> last unit)
// ---------------------------------------------------------------------------
__global__ void candidate_dist_kernel(const double *__restrict__ Fw, int Dpad, int D, int64_t N,
                                      const double *__restrict__ Qp, const int64_t *__restrict__ cand,
                                      int64_t T, int K, double *__restrict__ dist)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * K) return;
    const int64_t t = i / K;
    int64_t id = cand[i];
    if (id < 0) id += N;
    if (id < 0 || id >= N) { dist[i] = __builtin_nan(""); return; }
    const double *f = Fw + id * Dpad, *q = Qp + t * Dpad;
    double acc = 0.0;
    for (int c = 0; c < D; ++c) {
        const double d = __dsub_rn(f[c], q[c]);
        acc = __dadd_rn(acc, __dmul_rn(d, d));
    }
    dist[i] = __dsqrt_rn(acc);
}

void launch_candidate_dist(const double *Fw, int Dpad, int D, int64_t N, const double *Qp,
                           const int64_t *cand, int64_t T, int K, double *dist, hipStream_t s)
{
    const int64_t n = T * K;
    hipLaunchKernelGGL(candidate_dist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Fw, Dpad,
                       D, N, Qp, cand, T, K, dist);
}

// ---------------------------------------------------------------------------
// merge of G gathered per-shard top-K lists (multi-GPU exchange step)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
merge_topk_kernel(const double *__restrict__ d2, const int64_t *__restrict__ id, int G, int64_t T,
                  int K, int64_t *__restrict__ cand, double *__restrict__ dist)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t row = blockIdx.x;
    const int n = G * K;
    int P = 2;
    while (P < n) P <<= 1;
    double *key = reinterpret_cast<double *>(smem);
    int *idx = reinterpret_cast<int *>(smem + (size_t)P * sizeof(double));
    // unit ids travel through 32-bit sort keys: snk_set_shard refuses global_N >= 2^31
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        if (i < n) {
            const int gsh = i / K, j = i % K;
            const int64_t gi = id[((int64_t)gsh * T + row) * K + j];
            key[i] = (gi < 0) ? DBL_MAX : d2[((int64_t)gsh * T + row) * K + j];
            idx[i] = (gi < 0) ? 0x7fffffff : (int)gi;
        } else { key[i] = DBL_MAX; idx[i] = 0x7fffffff; }
    }
    __syncthreads();
    bitonic_sort_pairs(key, idx, P);
    for (int j = threadIdx.x; j < K; j += blockDim.x) {
        const bool ok = key[j] < DBL_MAX;
        cand[row * K + j] = ok ? (int64_t)idx[j] : -1;
        dist[row * K + j] = ok ? __dsqrt_rn(key[j]) : SNK_VERY_BIG;
    }
}

// The lists a shard sends are SORTED (knn_finalize_kernel writes a row's neighbours in the canonical order, padding last), so
// the merge is a tree of two-way merges of which only the first K outputs matter: one WAVEFRONT per row, every output position
// finds its element by a binary search along the merge path (7 steps at K = 100), log2(G) levels, no workgroup barrier.  The
// bitonic sort above took 0.84 ms per 19 200 rows at G = 8; a caller's lists that are NOT sorted (the entry point
// snk_merge_topk_dev does not promise it) are noticed while they are loaded and that row is sorted by its wavefront instead.
__device__ __forceinline__ void merge_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ void __launch_bounds__(256)
merge_topk_path_kernel(const double *__restrict__ d2, const int64_t *__restrict__ id, int G, int Gp, int P2, int64_t T,
                       int K, int64_t *__restrict__ cand, double *__restrict__ dist)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * wpb + wv;
    if (row >= T) return;
    const int nB = (Gp / 2) * K;
    // per wavefront: keys A (P2), keys B (nB), ids A (P2), ids B (nB)
    const size_t per_wave = ((size_t)(P2 + nB) * (sizeof(double) + sizeof(int)) + 15) & ~(size_t)15;
    double *keyA = reinterpret_cast<double *>(smem + (size_t)wv * per_wave);
    double *keyB = keyA + P2;
    int *idA = reinterpret_cast<int *>(keyB + nB);
    int *idB = idA + P2;
    const int n = G * K;
    for (int i = lane; i < P2; i += 64) {
        double k = DBL_MAX;
        int x = 0x7fffffff;
        if (i < n) {
            const int gsh = i / K, j = i - gsh * K;
            const int64_t gi = id[((int64_t)gsh * T + row) * K + j];
            if (gi >= 0) { k = d2[((int64_t)gsh * T + row) * K + j]; x = (int)gi; }
        }
        keyA[i] = k; idA[i] = x;
    }
    merge_wave_sync();
    bool bad = false;
    for (int i = lane; i < n; i += 64)
        if (i % K != 0 && pair_less(keyA[i], idA[i], keyA[i - 1], idA[i - 1])) bad = true;
    if (__any(bad)) {
        // not what a shard's re-rank writes: this row through a sort of its own (one wavefront, P2 entries)
        for (int k = 2; k <= P2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = lane; i < P2; i += 64) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const bool up = ((i & k) == 0);
                        const double ka = keyA[i], kb = keyA[ixj];
                        const int ia = idA[i], ib = idA[ixj];
                        const bool sw = up ? pair_less(kb, ib, ka, ia) : pair_less(ka, ia, kb, ib);
                        if (sw) { keyA[i] = kb; keyA[ixj] = ka; idA[i] = ib; idA[ixj] = ia; }
                    }
                }
                merge_wave_sync();
            }
    } else {
        double *ks = keyA, *kd = keyB;
        int *is = idA, *id_ = idB;
        for (int lists = Gp; lists > 1; lists >>= 1) {
            const int items = (lists >> 1) * K;
            for (int it = lane; it < items; it += 64) {
                const int m = it / K, o = it - m * K;
                const double *ak = ks + (size_t)(2 * m) * K, *bk = ak + K;
                const int *ai = is + (size_t)(2 * m) * K, *bi = ai + K;
                int lo = 0, hi = o;                 // elements of list a among the first o outputs
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (!pair_less(bk[o - 1 - mid], bi[o - 1 - mid], ak[mid], ai[mid])) lo = mid + 1;
                    else hi = mid;
                }
                const int i = lo, j = o - lo;
                const bool from_a = !pair_less(bk[j], bi[j], ak[i], ai[i]);
                kd[(size_t)m * K + o] = from_a ? ak[i] : bk[j];
                id_[(size_t)m * K + o] = from_a ? ai[i] : bi[j];
            }
            merge_wave_sync();
            double *tk = ks; ks = kd; kd = tk;
            int *ti = is; is = id_; id_ = ti;
        }
        if (ks != keyA) {                       // (an odd number of levels left the result in B)
            for (int j = lane; j < K; j += 64) { keyA[j] = ks[j]; idA[j] = is[j]; }
            merge_wave_sync();
        }
    }
    for (int j = lane; j < K; j += 64) {
        const bool ok = keyA[j] < DBL_MAX;
        cand[row * K + j] = ok ? (int64_t)idA[j] : -1;
        dist[row * K + j] = ok ? __dsqrt_rn(keyA[j]) : SNK_VERY_BIG;
    }
}

void launch_merge_topk(const double *d2, const int64_t *id, int G, int64_t T, int K, int64_t *cand,
                       double *dist, hipStream_t s)
{
    int P = 2;
    while (P < G * K) P <<= 1;
    {
        int Gp = 1;
        while (Gp < G) Gp <<= 1;
        int P2 = 2;
        while (P2 < Gp * K) P2 <<= 1;
        const size_t per_wave = ((size_t)(P2 + (Gp / 2) * K) * (sizeof(double) + sizeof(int)) + 15) & ~(size_t)15;
        int wpb = (int)((size_t)65536 / per_wave);
        if (wpb > 4) wpb = 4;
        if (wpb >= 1 && Gp >= 2) {
            const size_t shmem = per_wave * wpb;
            static size_t attr2[32] = {0};
            lds_attr_ensure(attr2, shmem, [&] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_topk_path_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); });
            hipLaunchKernelGGL(merge_topk_path_kernel, dim3((unsigned)((T + wpb - 1) / wpb)), dim3(64 * wpb), shmem, s, d2, id, G, Gp,
                               P2, T, K, cand, dist);
            return;
        }
    }
    const size_t shmem = (size_t)P * (sizeof(double) + sizeof(int));
    static size_t attr[32] = {0};
    lds_attr_ensure(attr, shmem, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_topk_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); });
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)T), dim3(256), shmem, s, d2, id, G, T, K,
                       cand, dist);
}

// ---------------------------------------------------------------------------
// database weighting: F = F_unw * wt (float64), row norms; JC = JC_unw * wj
// (speech_manip.py:209-213 applied by set_target_weights / set_join_weights)
// ---------------------------------------------------------------------------
__global__ void weight_target_kernel(const float *__restrict__ F_unw, int Fp, int64_t N, int Dt,
                                     const double *__restrict__ wt, double *__restrict__ Fw,
                                     double *__restrict__ fnorm, int64_t Nalloc, int Dpad)
{
    // one wave per row
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= Nalloc) return;
    double acc = 0.0;
    for (int c = lane; c < Dpad; c += 64) {
        double v = 0.0;
        if (row < N && c < Dt) v = __dmul_rn((double)F_unw[row * Fp + c], wt[c]);
        Fw[row * Dpad + c] = v;
        acc += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) fnorm[row] = (row < N) ? acc : __builtin_inf();
}

void launch_weight_target(const float *F_unw, int Fp, int64_t N, int Dt, const double *wt, double *Fw,
                          double *fnorm, int64_t Nalloc, int Dpad, const int32_t *, hipStream_t s)
{
    const int wpb = 4;
    hipLaunchKernelGGL(weight_target_kernel, dim3((unsigned)((Nalloc + wpb - 1) / wpb)),
                       dim3(64 * wpb), 0, s, F_unw, Fp, N, Dt, wt, Fw, fnorm, Nalloc, Dpad);
}

__global__ void weight_join_kernel(const float *__restrict__ JC_unw, int Jp, int64_t Njc, int Dj,
                                   const double *__restrict__ wj, double *__restrict__ JCw,
                                   int Djpad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = Njc * Djpad;
    if (i >= total) return;
    const int64_t row = i / Djpad;
    const int c = (int)(i % Djpad);
    JCw[i] = (c < Dj) ? __dmul_rn((double)JC_unw[row * Jp + c], wj[c]) : 0.0;
}

void launch_weight_join(const float *JC_unw, int Jp, int64_t Njc, int Dj, const double *wj, double *JCw,
                        int Djpad, hipStream_t s)
{
    const int64_t total = Njc * Djpad;
    hipLaunchKernelGGL(weight_join_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       JC_unw, Jp, Njc, Dj, wj, JCw, Djpad);
}

// ---------------------------------------------------------------------------
// MFMA mapping self test: C(16x16) = A(16x4) * B(4x16) with asymmetric integer data
// ---------------------------------------------------------------------------
__global__ void mfma_selftest_kernel(const double *A, const double *B, double *C)
{
    const int lane = threadIdx.x;
    const double a = A[(lane & 15) * 4 + (lane >> 4)];   // A[row = l&15][k = l>>4]
    const double b = B[(lane >> 4) * 16 + (lane & 15)];  // B[k = l>>4][col = l&15]
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[frag_row(lane, r) * 16 + (lane & 15)] = acc[r];
}

void launch_mfma_selftest(const double *A, const double *B, double *C, hipStream_t s)
{
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, s, A, B, C);
}

}  // namespace snk
