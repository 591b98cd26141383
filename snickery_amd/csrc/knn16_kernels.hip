// Single-precision PREFILTER for the K-NN preselection (gfx950): the same DB-stationary sweep as
// knn_kernels.hip, but on the f32 matrix pipe (v_mfma_f32_32x32x2_f32, exact f32 FMA chains at
// twice the f64 MFMA rate) over float32 copies of the weighted database and of -2q.
//
// Exactness is kept by construction, not by precision:
//   * key~ = fnorm - 2 q.f comes straight out of the accumulator (||f||^2 rides along in a spare
//     padding column against 1.0 in the query operand).
//   * |key~ - key| <= e(q,f) = c * (2 ||q|| ||f|| + ||f||^2): input rounding 2^-24 per operand plus
//     a (Dpad+1)-term f32 FMA chain; c = 2 (Dpad+3) 2^-24 (8e-6 at 64 columns).  (A half-precision
//     split (f16 hi+lo, 16x MFMA rate) was built first and rejected: with f32 accumulators its provable
//     bound is ~1e-4 relative, too coarse for databases whose neighbour distances differ by 1e-5.)
//   * thresholds are raised by eps_t = e(q_t, Fmax), the filter passes key~ <= thr + eps_t, and
//     knn_finalize re-ranks every survivor within 2 max_i e(q_t, f_i) (over the row's survivors) of
//     the K-th key with EXACT float64 canonical distances.  The approximate keys never reach the caller.
//   * anything that does not fit (no spare column, list overflow, > SEL_MAX near ties) falls back
//     to the f64 sweep.
//   * rows of DCH = 1..4 chunks of 64 columns (Dt <= 255): fragments [tile][chunk][j4][lane].
//
// Operand roles: A = database tile (32 units), B = query tile (32 frames); a lane of the 32x32
// result holds ONE query column and 16 database rows, so the threshold is one scalar per lane.
// The k index is permuted: within a chunk lane half h covers columns 32h..32h+31 (step kk -> column
// 32h+kk), so every lane reads 128 contiguous bytes per tile and chunk.
#include "snk_internal.h"
#include <stdlib.h>
#include <type_traits>
#include <float.h>

namespace snk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f16acc __attribute__((ext_vector_type(16)));

// PoolEntry's 16 bytes with the key as the float32 the matrix pipe produced (low half of PoolEntry's float64 key field, the high
// half unused): knn_bucket_kernel converts (keys_f32) -- a v_cvt_f64_f32 per tested result is an 8-cycle instruction on the issue
// port the survivors' path of these kernels is bound by, the bucket's eight per thread are not
struct __attribute__((aligned(16))) PoolEntry16 { float key; int unused; int idx; int row; };

// C/D layout of the 32x32 f32 MFMA results: lane l holds column (l & 31) and, in register r, row
//   (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
__device__ __forceinline__ int crow32(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------
// operand construction.  Fragment order: [tile][chunk = 0..DCH-1][j4 = 0..7][lane][4 floats] with
//   value = X[row(lane & 31)][column 64*chunk + 32*(lane >> 5) + 4*j4 + i]       (DCH = Dpad / 64)
// so each of the 8*DCH load instructions of a tile covers 1 KB of contiguous memory.
// ---------------------------------------------------------------------------
// database row held by row r of tile `tile`: identity for the full operand; for the sample operand
// wave w owns tiles [w*nt_a, (w+1)*nt_a), the 16*nt_a rows a lane half reduces together are spaced
// G sampled rows apart, and consecutive sampled rows fall into different groups
__device__ __forceinline__ int64_t tile_row16(int64_t tile, int r, int64_t sample_stride, int64_t G, int nt_a)
{
    if (sample_stride <= 0) return tile * 32 + r;
    const int64_t w = tile / nt_a;
    const int nt = (int)(tile % nt_a);
    const int hh = (r >> 2) & 1, reg = (r & 3) + 4 * (r >> 3);
    const int64_t m = (int64_t)nt * 16 + reg;
    return (m * G + (2 * w + hh)) * sample_stride;
}
// ... and the unit that stands at that POSITION of the operand: the database order, or the order the engine gave a voice whose
// own order says nothing about closeness (kmeans_kernels.hip; perm[position] = unit, positions >= N are padding either way)
__device__ __forceinline__ int64_t unit_at(const int32_t *__restrict__ perm, int64_t pos, int64_t N)
{
    return (perm && pos < N) ? (int64_t)perm[pos] : pos;
}

// class id of every tile row in tile order (-1 beyond the database): what the class-restricted sweep
// compares with the query's class
__global__ void build_class16_kernel(const int32_t *__restrict__ unit_class, int64_t N, int64_t n_tiles,
                                     int64_t sample_stride, int64_t G, int nt_a, int32_t *__restrict__ out,
                                     const int32_t *__restrict__ perm)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles * 32) return;
    const int64_t row = tile_row16(i >> 5, (int)(i & 31), sample_stride, G, nt_a);
    out[i] = (row < N) ? unit_class[unit_at(perm, row, N)] : -1;
}

void launch_build_class16(const int32_t *unit_class, int64_t N, int64_t n_tiles, int64_t sample_stride, int64_t G,
                          int nt_a, int32_t *out, hipStream_t s, const int32_t *perm)
{
    hipLaunchKernelGGL(build_class16_kernel, dim3((unsigned)((n_tiles * 32 + 255) / 256)), dim3(256), 0, s,
                       unit_class, N, n_tiles, sample_stride, G, nt_a, out, perm);
}

__global__ void build_db16_kernel(const double *__restrict__ Fw, const double *__restrict__ fnorm, int64_t N,
                                  int Dt, int Dpad, int64_t n_tiles, int64_t sample_stride, int64_t G,
                                  int nt_a, f32x4 *__restrict__ A32, const int32_t *__restrict__ perm)
{
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int per_tile = 8 * (Dpad / 64);
    if (item >= n_tiles * per_tile) return;
    const int64_t tile = item / per_tile;
    const int j4 = (int)(item % per_tile);            // chunk * 8 + j4
    const int r = lane & 31, h = lane >> 5;
    const int64_t pos = tile_row16(tile, r, sample_stride, G, nt_a);
    const int64_t row = unit_at(perm, pos, N);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 64 * (j4 >> 3) + 2 * (4 * (j4 & 7) + i) + h;       // k-step kk = 4 (j4 & 7) + i multiplies columns 2kk, 2kk+1:
                                                                       // padding columns gather in the last k-steps
        float x = 0.0f;
        if (pos < N) {
            if (c < Dt) x = (float)Fw[row * Dpad + c];
            else if (c == Dt) x = (float)fnorm[row];
        } else if (c == Dt) x = 3.0e38f;           // padding unit: key never passes
        v[i] = x;
    }
    A32[item * 64 + lane] = v;
}

__global__ void fmax_kernel(const double *__restrict__ fnorm, int64_t N, double *__restrict__ out)
{
    __shared__ double red[256];
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        m = fmax(m, fnorm[i]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(red[0]));
}

void launch_build_db16(const double *Fw, const double *fnorm, int64_t N, int Dt, int Dpad, int64_t n_tiles,
                       int64_t sample_stride, int64_t G, int nt_a, void *A32, hipStream_t s, const int32_t *perm)
{
    const int64_t items = n_tiles * 8 * (Dpad / 64);
    hipLaunchKernelGGL(build_db16_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, Fw, fnorm, N, Dt,
                       Dpad, n_tiles, sample_stride, G, nt_a, reinterpret_cast<f32x4 *>(A32), perm);
}

void launch_fmax(const double *fnorm, int64_t N, double *out, hipStream_t s)
{
    (void)hipMemsetAsync(out, 0, sizeof(double), s);
    hipLaunchKernelGGL(fmax_kernel, dim3(512), dim3(256), 0, s, fnorm, N, out);
}

// Query fragments (same fragment order): -2q in float32, 1.0 in the fnorm column; also eps_t.
__global__ void prepare_queries16_kernel(const double *__restrict__ Qp, const double *__restrict__ qnorm,
                                         int64_t T, int Dt, int Dpad, const double *__restrict__ fmax2,
                                         double eps_c, f32x4 *__restrict__ B32, double *__restrict__ eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_tiles = (T + 31) / 32;
    const int per_tile = 8 * (Dpad / 64);
    if (item >= n_tiles * per_tile) return;
    const int64_t tile = item / per_tile;
    const int j4 = (int)(item % per_tile);            // chunk * 8 + j4
    const int64_t row = tile * 32 + (lane & 31);
    const int h = lane >> 5;
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 64 * (j4 >> 3) + 2 * (4 * (j4 & 7) + i) + h;       // k-step kk = 4 (j4 & 7) + i multiplies columns 2kk, 2kk+1:
                                                                       // padding columns gather in the last k-steps
        float x = 0.0f;
        if (row < T) {
            if (c < Dt) x = (float)(-2.0 * Qp[row * Dpad + c]);
            else if (c == Dt) x = 1.0f;
        }
        v[i] = x;
    }
    B32[item * 64 + lane] = v;
    if (j4 == 0 && h == 0) {
        const double fm = sqrt(*fmax2);
        const double qn = (row < T) ? sqrt(qnorm[row]) : 0.0;
        eps[row] = eps_c * (2.0 * qn * fm + fm * fm) + 1e-30;
    }
}

void launch_prepare_queries16(const double *Qp, const double *qnorm, int64_t T, int Dt, int Dpad,
                              const double *fmax2, double eps_c, void *B32, double *eps, hipStream_t s)
{
    const int64_t items = ((T + 31) / 32) * 8 * (Dpad / 64);
    hipLaunchKernelGGL(prepare_queries16_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, Qp, qnorm,
                       T, Dt, Dpad, fmax2, eps_c, reinterpret_cast<f32x4 *>(B32), eps);
}

// ---------------------------------------------------------------------------
// the sweep
//   MODE 0: minima per (wave slab, lane half) group over the scattered sample -> gmin32[row][G]
//   MODE 1: filter over the whole database -> entry pool (same pool / bucket / finalize as f64)
// ---------------------------------------------------------------------------
// KU: MFMA k-steps actually issued per tile (<= 32 * DCH).  The operands are zero beyond column Dt (the
// norm column), so k-steps that would only multiply padding are skipped: 31 instead of 32 at Dt = 60, 61
// (3 % of the matrix work; adding 0 x 0 to an accumulator does not change it, the keys are the same bits).
template <int NT, int MODE, int DCH, bool CLS, int KU = 32 * DCH>
__global__ void __launch_bounds__(256, (NT <= 2 && DCH == 1 && !CLS) ? 2 : 1)
knn_sweep16(const f32x4 *__restrict__ A32, const f32x4 *__restrict__ B32,
            const int32_t *__restrict__ tile_class, const int32_t *__restrict__ query_class,
            const float *__restrict__ thr32, int nQT, int64_t n_slabs,
            unsigned int *__restrict__ slab_counter, int qsplit, int64_t n_main_slabs, int qsplit_tail,
            float *__restrict__ gmin32, int64_t G,
            PoolEntry16 *__restrict__ pool, unsigned int *__restrict__ pool_ctl, int *__restrict__ chunk_fill,
            int max_chunks, int pool_chunk)
{
    // per-wave staging area of survivors (a few per 32x32 tile); flushed to the wave's pool chunk one
    // tile later.  A burst that would not fit is flushed on the spot (SPOT: 768 entries = 49 KB of LDS per
    // workgroup, checked per group of four results) or provided for up front (room for every result of
    // a step, checked once per step).  The on-the-spot check sits inside the MFMA shadow code; the
    // three- and four-chunk variants have no registers left for it (1.1 KB of scratch, 4x slower).
#ifndef SNK_SWEEP16_SPOT
#define SNK_SWEEP16_SPOT(NT_, DCH_, CLS_) ((DCH_) <= 2)
#endif
    constexpr bool SPOT = SNK_SWEEP16_SPOT(NT, DCH, CLS);
    constexpr int STAGE_CAP = SPOT ? 768 : 64 * 16 * ((NT >= 2) ? 2 : 1) + 256;
    __shared__ PoolEntry16 stage[(MODE == 1) ? 4 : 1][(MODE == 1) ? STAGE_CAP : 1];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int qcol = lane & 31;

    auto grab = [&]() -> int64_t {
        unsigned int v = 0;
        if (lane == 0) v = atomicAdd(slab_counter, 1u);
        return (int64_t)__builtin_amdgcn_readfirstlane(v);
    };
    int chunk_id = -1, cused = pool_chunk, lcount = 0;
    auto new_chunk = [&]() {
        if (chunk_id >= 0 && lane == 0) chunk_fill[chunk_id] = cused;
        unsigned int c = 0;
        if (lane == 0) c = atomicAdd(&pool_ctl[0], 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        if ((int)c >= max_chunks) { if (lane == 0) pool_ctl[1] = 1u; chunk_id = -1; }
        else chunk_id = (int)c;
        cused = 0;
    };
    auto flush_stage = [&]() {
        if (cused + lcount > pool_chunk) new_chunk();
        if (chunk_id >= 0)
            for (int e = lane; e < lcount; e += 64)
                pool[(int64_t)chunk_id * pool_chunk + cused + e] = stage[wv][e];
        cused += lcount;
        lcount = 0;
    };

    // work items: whole rounds of (slab, part) items first; the slabs of the last, partial round
    // are cut into more parts so that the tail of the persistent sweep stays short
    const int64_t n_main_items = n_main_slabs * qsplit;
    const int64_t n_items = n_main_items + (n_slabs - n_main_slabs) * qsplit_tail;
    int64_t item = grab();
    while (item < n_items) {
        const int64_t item_next = grab();
        const bool tail = item >= n_main_items;
        const int qs = tail ? qsplit_tail : qsplit;
        const int64_t rel = tail ? item - n_main_items : item;
        const int64_t w = (tail ? n_main_slabs : 0) + rel / qs;
        const int part = (int)(rel % qs);
        const int qt_lo = (int)(((int64_t)nQT * part) / qs);
        const int qt_hi = (int)(((int64_t)nQT * (part + 1)) / qs);

        // database fragments of this slab: resident in registers
        constexpr int KS = 32 * DCH;                  // MFMA k-steps (and floats per lane) per 32-row tile
        float af[NT][KS];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j4 = 0; j4 < 8 * DCH; ++j4) {
                const f32x4 v = A32[((w * NT + nt) * (8 * DCH) + j4) * 64 + lane];
                af[nt][4 * j4] = v[0]; af[nt][4 * j4 + 1] = v[1]; af[nt][4 * j4 + 2] = v[2]; af[nt][4 * j4 + 3] = v[3];
            }

        // class-restricted search: the class of each of this lane's 16 rows per tile; a result counts
        // only when it equals the class of the lane's query column
        int ucls[CLS ? NT : 1][CLS ? 16 : 1];
        if (CLS) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) ucls[nt][r] = tile_class[(w * NT + nt) * 32 + crow32(lane, r)];
        }
        int qc_cur = -2, qc_nxt = -2, qc_prev = -2;

        float b0[KS], b1[KS];
        float th_cur = 0.f, th_nxt = 0.f, th_prev = -FLT_MAX;
        auto load_q = [&](int t, float (&x)[KS]) {
#pragma unroll
            for (int j4 = 0; j4 < 8 * DCH; ++j4) {
                const f32x4 v = B32[((int64_t)t * (8 * DCH) + j4) * 64 + lane];
                x[4 * j4] = v[0]; x[4 * j4 + 1] = v[1]; x[4 * j4 + 2] = v[2]; x[4 * j4 + 3] = v[3];
            }
            if (MODE == 1) th_nxt = thr32[t * 32 + qcol];
            if (CLS) qc_nxt = query_class[t * 32 + qcol];
        };
        int qt = qt_lo + (int)((w * 3) % (qt_hi - qt_lo));
        int qt_prev = qt;
        load_q(qt, b0);

        // software pipeline: while the 32-MFMA chain of one 32x32 tile issues, the 16 results of the
        // PREVIOUS tile are tested (and the few that pass staged in LDS) in the MFMA shadows
        constexpr int CH = (NT >= 2) ? 2 : 1;          // database tiles per step: independent MFMA chains
        constexpr int NSTEP = NT / CH;
        f16acc pacc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[j][r] = 0.0f;
        float gm = FLT_MAX;

        auto tile_body = [&](float (&x)[KS], float (&nx)[KS], int it) {
            // the tile's own operands must have landed (they were requested a tile ago)
#pragma unroll
            for (int k = 0; k < KS; ++k) asm volatile("" : "+v"(x[k]));
            if (MODE == 1) asm volatile("" : "+v"(th_nxt));
            if (CLS) asm volatile("" : "+v"(qc_nxt));
            th_cur = th_nxt;
            qc_cur = qc_nxt;
            if (MODE == 1 && lcount) flush_stage();      // staged entries leave a whole tile early
            const int qt_next = (qt + 1 == qt_hi) ? qt_lo : qt + 1;
            load_q(qt_next, nx);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                const int pnt = ((st > 0) ? st - 1 : NSTEP - 1) * CH;
                const float pth = (st > 0) ? th_cur : th_prev;
                const int pqt = (st > 0) ? qt : qt_prev;
                const int pqc = (st > 0) ? qc_cur : qc_prev;
                if (MODE == 1 && !SPOT && lcount > STAGE_CAP - 64 * 16 * CH) flush_stage();
                f16acc acc[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
                for (int kk = 0; kk < KU; ++kk) {
#pragma unroll
                    for (int j = 0; j < CH; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[st * CH + j][kk], x[kk], acc[j], 0, 0, 0);
                    // the pending results are tested during the first 32 k-steps (one chunk); with skipped
                    // trailing k-steps the schedule moves up by their number (at most 3)
                    static_assert(KS - KU >= 0 && KS - KU <= 3, "at most three skipped k-steps");
                    const int k = kk + (KS - KU);
                    if (k >= 32) continue;
                    // the 16*CH pending results are tested four at a time in the MFMA shadows: one
                    // min3/min + compare per group; the per-result ballots run only when some lane
                    // of the group passes (a few entries per 32x32 tile do)
                    if ((k & 3) == 3) {
                        const int e0 = (CH == 2) ? (k - 3) : ((k - 7) >> 1);      // first result of the group
                        if (e0 + 3 < 16 * CH && (CH == 2 || (k & 7) == 7)) {
                            const int j = e0 / 16, r0 = e0 % 16;
                            float v4[4] = {pacc[j][r0], pacc[j][r0 + 1], pacc[j][r0 + 2], pacc[j][r0 + 3]};
                            if (CLS) {                  // results of other classes never count (inf > any threshold)
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    v4[q] = (ucls[pnt + j][r0 + q] != pqc) ? __builtin_inff() : v4[q];
                            }
                            const float m4 = fminf(__builtin_fminf(__builtin_fminf(v4[0], v4[1]), v4[2]), v4[3]);
                            if (MODE == 0) gm = fminf(gm, m4);
                            else if (__any(m4 <= pth)) {
                                if (SPOT && lcount > STAGE_CAP - 256) flush_stage();
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int r = r0 + q;
                                    const float key = v4[q];
                                    const bool pass = key <= pth;
                                    const unsigned long long m = __ballot(pass);
                                    if (pass) {
                                        const int rank = __builtin_amdgcn_mbcnt_hi(
                                            (unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                        PoolEntry16 en;
                                        en.key = key; en.unused = 0;
                                        en.idx = (int)((w * NT + pnt + j) * 32 + crow32(lane, r));
                                        en.row = pqt * 32 + qcol;
                                        stage[wv][lcount + rank] = en;
                                    }
                                    lcount += __popcll(m);
                                }
                            }
                        }
                    }
                }
                if (MODE == 0 && st == 0) {
                    // the previous query tile's group minimum is complete now
                    if (it > 0) gmin32[((int64_t)qt_prev * 32 + qcol) * G + 2 * w + (lane >> 5)] = gm;
                    gm = FLT_MAX;
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    pacc[j] = acc[j];
                }
            }
            th_prev = th_cur;
            qc_prev = qc_cur;
            qt_prev = qt;
            qt = qt_next;
        };
        const int n_t = qt_hi - qt_lo;
        for (int it = 0; it < n_t; it += 2) {
            tile_body(b0, b1, it);
            if (it + 1 < n_t) tile_body(b1, b0, it + 1);
        }
        // drain the last pending step of this work item
        if (MODE == 1 && !SPOT && lcount > STAGE_CAP - 64 * 16 * CH) flush_stage();
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float key = (CLS && ucls[(NSTEP - 1) * CH + j][r] != qc_prev) ? __builtin_inff() : pacc[j][r];
                if (MODE == 0) gm = fminf(gm, key);
                else {
                    if (SPOT && lcount > STAGE_CAP - 64) flush_stage();
                    const bool pass = key <= th_prev;
                    const unsigned long long m = __ballot(pass);
                    if (pass) {
                        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        PoolEntry16 en;
                        en.key = key; en.unused = 0;
                        en.idx = (int)((w * NT + (NSTEP - 1) * CH + j) * 32 + crow32(lane, r));
                        en.row = qt_prev * 32 + qcol;
                        stage[wv][lcount + rank] = en;
                    }
                    lcount += __popcll(m);
                }
            }
        if (MODE == 0) gmin32[((int64_t)qt_prev * 32 + qcol) * G + 2 * w + (lane >> 5)] = gm;
        item = item_next;
    }
    if (MODE == 1) {
        if (lcount) flush_stage();
        if (chunk_id >= 0 && lane == 0) chunk_fill[chunk_id] = cused;
    }
}

template <int NT, int DCH, bool CLS, int KU = 32 * DCH>
static void launch16_t(int mode, int blocks, hipStream_t s, const void *A32, const void *B32,
                       const int32_t *tile_class, const int32_t *query_class,
                       const float *thr32, int nQT, int64_t n_slabs, unsigned int *ctr,
                       int qsplit, int64_t n_main, int qtail, float *gmin32, int64_t G, void *pool,
                       unsigned int *pool_ctl, int *chunk_fill, int max_chunks, int pool_chunk)
{
#define SNK_L16(MODE_)                                                                                    \
    hipLaunchKernelGGL((knn_sweep16<NT, MODE_, DCH, CLS, KU>), dim3(blocks), dim3(256), 0, s, (const f32x4 *)A32, \
                       (const f32x4 *)B32, tile_class, query_class, thr32, nQT, n_slabs, ctr, qsplit, n_main, qtail, \
                       gmin32, G, (PoolEntry16 *)pool, pool_ctl, chunk_fill, max_chunks, pool_chunk)
    if (mode == 0) SNK_L16(0);
    else SNK_L16(1);
#undef SNK_L16
}

// nt: tiles (32 units) per wave, dch: 64-column chunks per row; k_steps: ceil((Dt + 1) / 2) k-steps carry
// data (only 31 at dch = 1 has a variant of its own); tile_class / query_class non-null:
// class-restricted search.  Returns false when the shape is not instantiated.
bool launch_knn_sweep16(int mode, int nt, int dch, int k_steps, int grid_cus, const void *A32, const void *B32,
                        const int32_t *tile_class, const int32_t *query_class,
                        const float *thr32, int64_t T32, int64_t n_slabs, unsigned int *ctr,
                        float *gmin32, int64_t G, void *pool, unsigned int *pool_ctl, int *chunk_fill,
                        int max_chunks, int pool_chunk, hipStream_t s)
{
    const bool cls = tile_class != nullptr;
    const int nQT = (int)(T32 / 32);
    const int64_t max_blocks = (int64_t)grid_cus * ((nt <= 2 && dch == 1 && !cls) ? 2 : 1);
    int qsplit = 1;
    while (n_slabs * qsplit < 2 * 4 * max_blocks && qsplit * 2 <= nQT && qsplit < 8) qsplit *= 2;
    int64_t blocks = (n_slabs * qsplit + 3) / 4;
    if (blocks > max_blocks) blocks = max_blocks;
    int64_t n_main = n_slabs;
    int qtail = qsplit;
    sweep_tail_split(n_slabs, qsplit, blocks * 4, nQT, &n_main, &qtail);
#define SNK_NT16(NT_, DCH_, CLS_)                                                                          \
    if (nt == NT_ && dch == DCH_ && cls == CLS_) {                                                         \
        launch16_t<NT_, DCH_, CLS_>(mode, (int)blocks, s, A32, B32, tile_class, query_class, thr32, nQT, n_slabs, ctr, \
                                    qsplit, n_main, qtail, gmin32, G, pool, pool_ctl, chunk_fill, max_chunks, pool_chunk); \
        return true;                                                                                       \
    }
    // shapes whose last k-steps are all padding (the reference's own widths): Dt = 60, 61 (magphase-60
    // epoch targets, 31 of 32 k-steps) and Dt = 184, 185 (three-point halfphone targets, 93 of 96)
#define SNK_KU16(NT_, DCH_, CLS_, KU_)                                                                     \
    if (nt == NT_ && dch == DCH_ && cls == CLS_ && k_steps == KU_) {                                       \
        launch16_t<NT_, DCH_, CLS_, KU_>(mode, (int)blocks, s, A32, B32, tile_class, query_class, thr32, nQT, n_slabs, \
                                         ctr, qsplit, n_main, qtail, gmin32, G, pool, pool_ctl, chunk_fill, max_chunks, \
                                         pool_chunk);                                                      \
        return true;                                                                                       \
    }
    SNK_KU16(4, 1, false, 31) SNK_KU16(1, 3, false, 93) SNK_KU16(1, 3, true, 93)
#undef SNK_KU16
    SNK_NT16(4, 1, false) SNK_NT16(2, 1, false) SNK_NT16(8, 1, false) SNK_NT16(2, 2, false) SNK_NT16(1, 3, false)
    SNK_NT16(1, 4, false)
    SNK_NT16(2, 1, true) SNK_NT16(2, 2, true) SNK_NT16(1, 3, true) SNK_NT16(1, 4, true)
#undef SNK_NT16
    return false;
}

// ===========================================================================================================
// bf16-split prefilter: the same sweep on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the f32 MFMA rate).
// Every operand value x is carried as two bf16 pieces, x = hi + lo + r with |lo| <= 2^-8 |x| and |r| <= 2^-17 |x|
// (hi = bf16(x), lo = bf16(x - hi), both round-to-nearest; the subtraction is exact in float64), and a k-block of
// 16 columns costs THREE MFMAs (hi.hi + hi.lo + lo.hi; lo.lo <= 2^-16 |x||y| is dropped like the residuals: the
// sweep runs at the rate the matrix pipe issues, so a quarter fewer MFMAs is a quarter less time) or FOUR (with
// lo.lo: a third less key error, for data whose near ties make the exact re-rank the larger cost), accumulating in float32.  ||f||^2 rides in THREE
// spare columns as three bf16 pieces against 1.0 (24 bits), which is why the variant needs Dpad - Dt >= 3.
//     |key~ - key| <= cq ||f|| + c_acc (2 ||q|| ||f|| + ||f||^2)
//     cq: what the split drops, from the norms of the dropped pieces themselves (prepare_queries16b_kernel; worst case
//         3 2^-16 (2 ||q||) with three terms, 2 2^-16 with four; measured on the data a third of that);
//     c_acc = 1.02 (2^-20 (MFMAs per CHUNK + 1) + 2 2^-24 chunks + 2^-24): accumulation chains are one chunk of 64 columns
//         long, the chunks of wider rows are added in float32; every MFMA is ASSUMED off by at most 2^-20 of the sum of its
//         |products| and |C| -- the unit's internal order is not documented, so it was probed (snk_probe_mfma_bf16): it
//         aligns the sixteen products to the largest exponent and cuts each two bits below the float32 unit of that
//         exponent (one product of 1 beside fifteen of 0.97 2^-24 comes back 6.5 2^-24 short), i.e. < 17 2^-25 + 2^-24
//         = 0.6 2^-20 of the largest term by that model, 0.55 2^-20 observed on patterns built for it; a first version
//         of this bound assumed 2^-22 and the probe refuted it -- and the 24 bits of the three norm pieces.
// tests/test_gpu_prefilter.py measures the real deviation against float64 keys: it must stay below half of that
// (measured: 7-12 % of it -- rounding errors do not line up).
// The k index inside a k-block follows the instruction's operand map: lane l (r = l & 31, h = l >> 5) holds columns
// 16 kb + 8 h + j, j = 0..7, of row r.  Operand buffers: [tile][kb][piece][lane] x 16 bytes -- 8 KB per tile and
// 64 columns, the float32 operand's size.
// ===========================================================================================================
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned int bf16_rne_bits(float x)          // bits of bf16(x), round to nearest even
{
    unsigned int u = __builtin_bit_cast(unsigned int, x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_bits_to_float(unsigned int b) { return __builtin_bit_cast(float, b << 16); }

// pieces of one value: hi, lo (as bf16 bit patterns)
__device__ __forceinline__ void bf16_split(double x, unsigned int &hi, unsigned int &lo)
{
    hi = bf16_rne_bits((float)x);
    lo = bf16_rne_bits((float)(x - (double)bf16_bits_to_float(hi)));
}

// value of column c of database row `row` in the augmented operand: features, then three pieces of ||f||^2
__device__ __forceinline__ void db16b_column(const double *__restrict__ Fw, const double *__restrict__ fnorm, int64_t N,
                                             int Dt, int Dpad, int64_t row, int c, unsigned int &hi, unsigned int &lo)
{
    hi = 0u; lo = 0u;
    if (row < N) {
        if (c < Dt) bf16_split(Fw[row * Dpad + c], hi, lo);
        else if (c < Dt + 3) {
            // ||f||^2 = n0 + n1 + n2 (+ 2^-27 relative): each piece exact in bf16, all three in the hi operand
            const double n = fnorm[row];
            const unsigned int b0 = bf16_rne_bits((float)n);
            const double r1 = n - (double)bf16_bits_to_float(b0);
            const unsigned int b1 = bf16_rne_bits((float)r1);
            const double r2 = r1 - (double)bf16_bits_to_float(b1);
            hi = (c == Dt) ? b0 : (c == Dt + 1) ? b1 : bf16_rne_bits((float)r2);
        }
    } else if (c == Dt) hi = 0x7f7fu;                      // padding unit: the largest finite bf16, its key never passes
}

__global__ void build_db16b_kernel(const double *__restrict__ Fw, const double *__restrict__ fnorm, int64_t N,
                                   int Dt, int Dpad, int64_t n_tiles, int64_t sample_stride, int64_t G, int nt_a,
                                   u32x4 *__restrict__ A16, const int32_t *__restrict__ perm)
{
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // (tile, kb)
    const int KB = Dpad / 16;
    if (item >= n_tiles * KB) return;
    const int64_t tile = item / KB;
    const int kb = (int)(item % KB);
    const int r = lane & 31, h = lane >> 5;
    const int64_t pos = tile_row16(tile, r, sample_stride, G, nt_a);
    const int64_t row = pos < N ? unit_at(perm, pos, N) : pos;          // (a position past the database stays one: padding unit)
    unsigned int hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) db16b_column(Fw, fnorm, N, Dt, Dpad, row, 16 * kb + 8 * h + j, hi[j], lo[j]);
    u32x4 vh, vl;
#pragma unroll
    for (int j = 0; j < 4; ++j) { vh[j] = hi[2 * j] | (hi[2 * j + 1] << 16); vl[j] = lo[2 * j] | (lo[2 * j + 1] << 16); }
    A16[(item * 2 + 0) * 64 + lane] = vh;
    A16[(item * 2 + 1) * 64 + lane] = vl;
}

void launch_build_db16b(const double *Fw, const double *fnorm, int64_t N, int Dt, int Dpad, int64_t n_tiles,
                        int64_t sample_stride, int64_t G, int nt_a, void *A16, hipStream_t s, const int32_t *perm)
{
    const int64_t items = n_tiles * (Dpad / 16);
    hipLaunchKernelGGL(build_db16b_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, Fw, fnorm, N, Dt, Dpad,
                       n_tiles, sample_stride, G, nt_a, reinterpret_cast<u32x4 *>(A16), perm);
}

// What the split drops, measured on the data instead of assumed at its worst: with f = fh + fl + rf per element,
// rho[0] = max_i ||fl_i||^2 / ||f_i||^2 and rho[1] = max_i ||rf_i||^2 / ||f_i||^2 over the rows of the operand
// (worst case 2^-16 and 2^-32; rounding errors of 61 columns do not line up, so a third of that is typical).
__global__ void db16b_ratio_kernel(const double *__restrict__ Fw, int64_t N, int Dt, int Dpad,
                                   unsigned long long *__restrict__ rho)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double rl = 0.0, rr = 0.0;
    if (row < N) {
        double sf = 0.0, sl = 0.0, sr = 0.0;
        for (int c = 0; c < Dt; ++c) {
            const double x = Fw[row * Dpad + c];
            unsigned int hb, lb;
            bf16_split(x, hb, lb);
            const double l = (double)bf16_bits_to_float(lb);
            const double r = x - (double)bf16_bits_to_float(hb) - l;          // exact: three float64 with < 53 bits between them
            sf += x * x; sl += l * l; sr += r * r;
        }
        if (sf > 0.0) { rl = sl / sf; rr = sr / sf; }
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = rl; red[1][threadIdx.x] = rr;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] = fmax(red[0][threadIdx.x], red[0][threadIdx.x + off]);
            red[1][threadIdx.x] = fmax(red[1][threadIdx.x], red[1][threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {           // non-negative doubles order like their bit patterns
        atomicMax(&rho[0], (unsigned long long)__double_as_longlong(red[0][0]));
        atomicMax(&rho[1], (unsigned long long)__double_as_longlong(red[1][0]));
    }
}

void launch_db16b_ratios(const double *Fw, int64_t N, int Dt, int Dpad, double *rho, hipStream_t s, bool accumulate)
{
    if (!accumulate) (void)hipMemsetAsync(rho, 0, 2 * sizeof(double), s);
    hipLaunchKernelGGL(db16b_ratio_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, Fw, N, Dt, Dpad,
                       reinterpret_cast<unsigned long long *>(rho));
}

// query operand: a = -2q in two pieces (a = ah + al + ra), 1.0 against the three norm pieces.  Per row also the
// coefficient of the representation error and the key bound the filter works with:
//     a.f - (ah.fh + ah.fl + al.fh) = ah.rf + al.fl + al.rf + ra.f,   |.| <= ||f|| cq,
//     cq = (||ah|| + ||al||) rho_R + ||al|| rho_L + ||ra||          (norms of THIS row's pieces, rho of the operand)
//     eps = cq Fmax + c_acc (2 ||q|| Fmax + Fmax^2)                  (c_acc: accumulation + the norm pieces' 2^-24)
__global__ void prepare_queries16b_kernel(const double *__restrict__ Qp, const double *__restrict__ qnorm, int64_t T,
                                          int Dt, int Dpad, const double *__restrict__ fmax2, const double *__restrict__ rho,
                                          double c_acc, u32x4 *__restrict__ B16, double *__restrict__ eps, double *__restrict__ cq,
                                          double c_coarse, double *__restrict__ e1)
{
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_tiles = (T + 31) / 32;
    const int KB = Dpad / 16;
    if (item >= n_tiles * KB) return;
    const int64_t tile = item / KB;
    const int kb = (int)(item % KB);
    const int64_t row = tile * 32 + (lane & 31);
    const int h = lane >> 5;
    unsigned int hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 16 * kb + 8 * h + j;
        hi[j] = 0u; lo[j] = 0u;
        if (row < T) {
            if (c < Dt) bf16_split(-2.0 * Qp[row * Dpad + c], hi[j], lo[j]);
            else if (c < Dt + 3) hi[j] = 0x3f80u;            // 1.0
        }
    }
    u32x4 vh, vl;
#pragma unroll
    for (int j = 0; j < 4; ++j) { vh[j] = hi[2 * j] | (hi[2 * j + 1] << 16); vl[j] = lo[2 * j] | (lo[2 * j + 1] << 16); }
    B16[(item * 2 + 0) * 64 + lane] = vh;
    B16[(item * 2 + 1) * 64 + lane] = vl;
    if (kb == 0 && h == 0) {
        double sh = 0.0, sl = 0.0, sr = 0.0;
        if (row < T)
            for (int c = 0; c < Dt; ++c) {
                const double a = -2.0 * Qp[row * Dpad + c];
                unsigned int hb, lb;
                bf16_split(a, hb, lb);
                const double ah = (double)bf16_bits_to_float(hb), al = (double)bf16_bits_to_float(lb);
                const double ra = a - ah - al;
                sh += ah * ah; sl += al * al; sr += ra * ra;
            }
        const double up = 1.0 + 1e-12;                      // the square roots and sums above round
        const double nah = sqrt(sh) * up, nal = sqrt(sl) * up, nra = sqrt(sr) * up;
        const double c = ((nah + nal) * sqrt(rho[1]) + nal * sqrt(rho[0])) * up + nra;
        const double fm = sqrt(*fmax2) * up;
        const double qn = (row < T) ? sqrt(qnorm[row]) * up : 0.0;
        eps[row] = c * fm + c_acc * (2.0 * qn * fm + fm * fm) + 1e-30;
        cq[row] = c;
        // coarse pass (knn_coarse16b: the hi.hi term alone): its key differs from the three-term key by the two cross
        // terms, |ah.fl + al.fh| <= (||ah|| rho_L + ||al|| (1 + 2^-8)) ||f||, and by the two accumulation errors
        if (e1) e1[row] = (nah * sqrt(rho[0]) * up + nal * 1.00390625) * fm * up + c_coarse * (2.0 * qn * fm + fm * fm) + 1e-30;
    }
}

void launch_prepare_queries16b(const double *Qp, const double *qnorm, int64_t T, int Dt, int Dpad, const double *fmax2,
                               const double *rho, double c_acc, void *B16, double *eps, double *cq, hipStream_t s,
                               double c_coarse, double *e1)
{
    const int64_t items = ((T + 31) / 32) * (Dpad / 16);
    hipLaunchKernelGGL(prepare_queries16b_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, Qp, qnorm, T, Dt,
                       Dpad, fmax2, rho, c_acc, reinterpret_cast<u32x4 *>(B16), eps, cq, c_coarse, e1);
}

// The sweep: structure of knn_sweep16 (DB-stationary, persistent wavefronts, results of the previous step tested
// in the shadow of the current step's MFMAs), 16 MFMAs per 32x32 tile and 64 columns instead of 31-32.
template <int NT, int MODE, int KB, int TERMS>
__global__ void __launch_bounds__(256, 1)
knn_sweep16b(const u32x4 *__restrict__ A16, const u32x4 *__restrict__ B16, const float *__restrict__ thr32, int nQT,
             int64_t n_slabs, unsigned int *__restrict__ slab_counter, int qsplit, int64_t n_main_slabs, int qsplit_tail,
             float *__restrict__ gmin32, int64_t G, PoolEntry16 *__restrict__ pool, unsigned int *__restrict__ pool_ctl,
             int *__restrict__ chunk_fill, int max_chunks, int pool_chunk)
{
    constexpr int STAGE_CAP = 768;
    __shared__ PoolEntry16 stage[(MODE == 1) ? 4 : 1][(MODE == 1) ? STAGE_CAP : 1];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int qcol = lane & 31;

    auto grab = [&]() -> int64_t {
        unsigned int v = 0;
        if (lane == 0) v = atomicAdd(slab_counter, 1u);
        return (int64_t)__builtin_amdgcn_readfirstlane(v);
    };
    int chunk_id = -1, cused = pool_chunk, lcount = 0;
    auto new_chunk = [&]() {
        if (chunk_id >= 0 && lane == 0) chunk_fill[chunk_id] = cused;
        unsigned int c = 0;
        if (lane == 0) c = atomicAdd(&pool_ctl[0], 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        if ((int)c >= max_chunks) { if (lane == 0) pool_ctl[1] = 1u; chunk_id = -1; }
        else chunk_id = (int)c;
        cused = 0;
    };
    auto flush_stage = [&]() {
        if (cused + lcount > pool_chunk) new_chunk();
        if (chunk_id >= 0)
            for (int e = lane; e < lcount; e += 64)
                pool[(int64_t)chunk_id * pool_chunk + cused + e] = stage[wv][e];
        cused += lcount;
        lcount = 0;
    };
    auto mfma = [](const u32x4 &a, const u32x4 &b, f16acc c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };

    const int64_t n_main_items = n_main_slabs * qsplit;
    const int64_t n_items = n_main_items + (n_slabs - n_main_slabs) * qsplit_tail;
    int64_t item = grab();
    while (item < n_items) {
        const int64_t item_next = grab();
        const bool tail = item >= n_main_items;
        const int qs = tail ? qsplit_tail : qsplit;
        const int64_t rel = tail ? item - n_main_items : item;
        const int64_t w = (tail ? n_main_slabs : 0) + rel / qs;
        const int part = (int)(rel % qs);
        const int qt_lo = (int)(((int64_t)nQT * part) / qs);
        const int qt_hi = (int)(((int64_t)nQT * (part + 1)) / qs);

        // database fragments of this slab: resident in registers (hi and lo pieces)
        u32x4 ah[NT][KB], al[NT][KB];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                ah[nt][kb] = A16[(((w * NT + nt) * KB + kb) * 2 + 0) * 64 + lane];
                al[nt][kb] = A16[(((w * NT + nt) * KB + kb) * 2 + 1) * 64 + lane];
            }
        u32x4 bh0[KB], bl0[KB], bh1[KB], bl1[KB];
        float th_cur = 0.f, th_nxt = 0.f, th_prev = -FLT_MAX;
        auto load_q = [&](int t, u32x4 (&xh)[KB], u32x4 (&xl)[KB]) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                xh[kb] = B16[(((int64_t)t * KB + kb) * 2 + 0) * 64 + lane];
                xl[kb] = B16[(((int64_t)t * KB + kb) * 2 + 1) * 64 + lane];
            }
            if (MODE == 1) th_nxt = thr32[t * 32 + qcol];
        };
        int qt = qt_lo + (int)((w * 3) % (qt_hi - qt_lo));
        int qt_prev = qt;
        load_q(qt, bh0, bl0);

        constexpr int CH = (NT >= 2) ? 2 : 1;          // database tiles per step: independent MFMA chains
        constexpr int NSTEP = NT / CH;
        static_assert(NSTEP <= 2, "the two accumulator sets alternate per step");
        // two accumulator sets: the MFMAs of a step fill one while the results of the step before are tested out of
        // the other -- no copies.  NSTEP == 2: set = step of the query tile; NSTEP == 1: set = parity of the tile.
        f16acc S0[CH], S1[CH];
        const int idx_lane = 4 * (lane >> 5), idx_wave = (int)(w * NT * 32);
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { S0[j][r] = 0.0f; S1[j][r] = 0.0f; }
        float gm = FLT_MAX;

        // one group of four pending results: v4 of database tile pt, result registers r0..r0+3, of query tile pq
        auto test4 = [&](const f16acc &res, int pt, int r0, float pth, int pq) {
            const float v0 = res[r0], v1 = res[r0 + 1], v2 = res[r0 + 2], v3 = res[r0 + 3];
            if (MODE == 0) {
                // (builtins, not inline asm: the matrix results now live in architected registers -- Makefile,
                // -amdgpu-mfma-vgpr-form -- and an asm statement reading them gets none of the wait states the compiler
                // inserts between a matrix instruction and the first vector read of its result: the drain at the end of a
                // work item read unfinished results and neighbours went missing, tests/test_gpu_fullsize.py long utterance)
                gm = __builtin_fminf(__builtin_fminf(gm, v0), v1);
                gm = __builtin_fminf(__builtin_fminf(gm, v2), v3);
            } else {
                float m4;
                m4 = __builtin_fminf(__builtin_fminf(v0, v1), v2);
                m4 = __builtin_fminf(m4, v3);
                if (__any(m4 <= pth)) {
                    if (lcount > STAGE_CAP - 256) flush_stage();
                    const float v4[4] = {v0, v1, v2, v3};
                    // the ids are put together here, not hoisted per (tile, register) out of the loop
                    int ib = idx_lane, wb = idx_wave;
                    asm volatile("" : "+v"(ib));
                    asm volatile("" : "+s"(wb));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float key = v4[q];
                        const bool pass = key <= pth;
                        const unsigned long long mm = __ballot(pass);
                        // (a uniform branch per result: where the units stand in no order a third of the groups hold a key under the
                        // threshold, but only one result in ten does -- the other three of such a group cost a compare and a branch)
                        if (mm) {
                            if (pass) {
                                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mm, 0u));
                                PoolEntry16 en;
                                en.key = key; en.unused = 0;
                                en.idx = wb + ib + (pt * 32 + crow32(0, r0 + q));
                                en.row = pq * 32 + qcol;
                                stage[wv][lcount + rank] = en;
                            }
                            lcount += __popcll(mm);
                        }
                    }
                }
            }
        };
        // one step: CH database tiles (first one: t0) against the query tile in (xh, xl) into `cur`, while the CH
        // results in `prev` (database tiles from pt0 of query tile pq, threshold pth) are tested in the MFMA shadows
        auto step = [&](f16acc (&cur)[CH], const f16acc (&prev)[CH], int t0, u32x4 (&xh)[KB], u32x4 (&xl)[KB], int pt0, float pth,
                        int pq) {
            constexpr int NM = TERMS * KB;                 // MFMA slots per tile (hi.hi, hi.lo, lo.hi [, lo.lo] per k-block)
            constexpr int NGRP = 4 * CH;                   // groups of four pending results
            static_assert(NM >= NGRP, "at most one group per slot");
            f16acc part[CH];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int kb = m / TERMS, term = m % TERMS;   // term 0: hi.hi, 1: hi(db).lo(query), 2: lo(db).hi(query), 3: lo.lo
                // accumulation chains are one chunk of 64 columns long (4 TERMS MFMAs through C): a longer chain's error
                // bound grows with its length (every MFMA's error is relative to the whole running sum), so the chunks of
                // wider rows are accumulated apart and added in float32 -- c_acc is that of one chunk whatever Dpad
                constexpr int CM = 4 * TERMS;                  // MFMA slots per chunk
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const u32x4 &a = (term & 2) ? al[t0 + j][kb] : ah[t0 + j][kb];
                    const u32x4 &b = (term & 1) ? xl[kb] : xh[kb];
                    if (m % CM == 0) {
                        f16acc z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                        part[j] = mfma(a, b, z);
                    } else part[j] = mfma(a, b, part[j]);
                    if (m % CM == CM - 1 || m == NM - 1) {     // the chunk's chain is complete
                        if (m < CM) cur[j] = part[j];
                        else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) cur[j][r] += part[j][r];
                        }
                    }
                }
                // group g is tested after slot (g + 1) NM / NGRP - 1: spread evenly over the NM slots
                {
                    const int g = ((m + 1) * NGRP + NM - 1) / NM - 1;
                    if (((g + 1) * NM) / NGRP - 1 == m) test4(prev[(4 * g) / 16], pt0 + (4 * g) / 16, (4 * g) % 16, pth, pq);
                }
            }
        };

        auto tile_body = [&](u32x4 (&xh)[KB], u32x4 (&xl)[KB], u32x4 (&nh)[KB], u32x4 (&nl)[KB], int it, auto PAR) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) { asm volatile("" : "+v"(xh[kb])); asm volatile("" : "+v"(xl[kb])); }
            if (MODE == 1) asm volatile("" : "+v"(th_nxt));
            th_cur = th_nxt;
            if (MODE == 1 && lcount) flush_stage();      // staged entries leave a whole tile early
            const int qt_next = (qt + 1 == qt_hi) ? qt_lo : qt + 1;
            load_q(qt_next, nh, nl);
            if (NSTEP == 2) {
                // step 0 tests the second half of the previous query tile, step 1 the first half of this one
                step(S0, S1, 0, xh, xl, CH, th_prev, qt_prev);
                if (MODE == 0) {
                    if (it > 0) gmin32[((int64_t)qt_prev * 32 + qcol) * G + 2 * w + (lane >> 5)] = gm;
                    gm = FLT_MAX;
                }
                step(S1, S0, CH, xh, xl, 0, th_cur, qt);
            } else {
                if (decltype(PAR)::value == 0) step(S0, S1, 0, xh, xl, 0, th_prev, qt_prev);
                else step(S1, S0, 0, xh, xl, 0, th_prev, qt_prev);
                if (MODE == 0) {
                    if (it > 0) gmin32[((int64_t)qt_prev * 32 + qcol) * G + 2 * w + (lane >> 5)] = gm;
                    gm = FLT_MAX;
                }
            }
            th_prev = th_cur;
            qt_prev = qt;
            qt = qt_next;
        };
        const int n_t = qt_hi - qt_lo;
        for (int it = 0; it < n_t; it += 2) {
            tile_body(bh0, bl0, bh1, bl1, it, std::integral_constant<int, 0>());
            if (it + 1 < n_t) tile_body(bh1, bl1, bh0, bl0, it + 1, std::integral_constant<int, 1>());
        }
        // drain the last pending step of this work item
        auto drain = [&](const f16acc (&res)[CH]) {
#pragma unroll
            for (int j = 0; j < CH; ++j)
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 4) test4(res[j], (NSTEP - 1) * CH + j, r0, th_prev, qt_prev);
        };
        if (NSTEP == 2 || (n_t & 1) == 0) drain(S1);
        else drain(S0);
        if (MODE == 0) gmin32[((int64_t)qt_prev * 32 + qcol) * G + 2 * w + (lane >> 5)] = gm;
        item = item_next;
    }
    if (MODE == 1) {
        if (lcount) flush_stage();
        if (chunk_id >= 0 && lane == 0) chunk_fill[chunk_id] = cused;
    }
}

// nt / dch as launch_knn_sweep16; instantiated for the reference's own widths: one chunk (Dt <= 61: one-point / epoch
// targets) with four tiles per wavefront, two chunks (Dt <= 125: two-point halfphone targets) with two, three chunks
// (Dt <= 189: three-point) with one.  Returns false for any other shape (f32 prefilter then).
bool knn_sweep16b_supported(int nt, int dch, int Dt, int Dpad, bool cls)
{
    return !cls && Dpad - Dt >= 3 && ((nt == 4 && dch == 1) || (nt == 2 && dch == 2) || (nt == 1 && dch == 3));
}

bool launch_knn_sweep16b(int mode, int terms, int nt, int dch, int grid_cus, const void *A16, const void *B16, const float *thr32,
                         int64_t T32, int64_t n_slabs, unsigned int *ctr, float *gmin32, int64_t G, void *pool,
                         unsigned int *pool_ctl, int *chunk_fill, int max_chunks, int pool_chunk, hipStream_t s)
{
    const int nQT = (int)(T32 / 32);
    const int64_t max_blocks = grid_cus;
    int qsplit = 1;
    while (n_slabs * qsplit < 2 * 4 * max_blocks && qsplit * 2 <= nQT && qsplit < 8) qsplit *= 2;
    int64_t blocks = (n_slabs * qsplit + 3) / 4;
    if (blocks > max_blocks) blocks = max_blocks;
    int64_t n_main = n_slabs;
    int qtail = qsplit;
    sweep_tail_split(n_slabs, qsplit, blocks * 4, nQT, &n_main, &qtail);
#define SNK_L16T(NT_, KB_, MODE_, TERMS_)                                                                    \
    hipLaunchKernelGGL((knn_sweep16b<NT_, MODE_, KB_, TERMS_>), dim3((unsigned)blocks), dim3(256), 0, s, (const u32x4 *)A16, \
                       (const u32x4 *)B16, thr32, nQT, n_slabs, ctr, qsplit, n_main, qtail, gmin32, G,       \
                       (PoolEntry16 *)pool, pool_ctl, chunk_fill, max_chunks, pool_chunk)
#define SNK_L16B(NT_, KB_, MODE_) do { if (terms == 4) SNK_L16T(NT_, KB_, MODE_, 4); else SNK_L16T(NT_, KB_, MODE_, 3); } while (0)
    if (nt == 4 && dch == 1) { if (mode == 0) SNK_L16B(4, 4, 0); else SNK_L16B(4, 4, 1); return true; }
    if (nt == 2 && dch == 2) { if (mode == 0) SNK_L16B(2, 8, 0); else SNK_L16B(2, 8, 1); return true; }
    if (nt == 1 && dch == 3) { if (mode == 0) SNK_L16B(1, 12, 0); else SNK_L16B(1, 12, 1); return true; }
#undef SNK_L16B
#undef SNK_L16T
    return false;
}


// ===========================================================================================================
// Two-pass filter on the bf16-split operands (prefilter 3, the default where the shape has a variant).
//
// Of the three MFMA terms of a product the hi.hi term alone already tells, for almost every (32 database rows) x
// (32 query rows) tile, that nothing in it can pass: the cross terms move a key by at most e1 = (||ah|| rho_L +
// ||al||) ||f|| (+ the accumulation terms) -- a few 10^-3 of ||q|| ||f||, far less than what separates an
// average unit from the K-th nearest one.  So:
//   pass 1, knn_coarse16b: the sweep of knn_sweep16b with the hi pieces only -- a third of the MFMAs, half the
//     query bytes, the database's hi pieces of EIGHT tiles resident per wavefront (the registers the lo pieces took) --
//     tests min(tile) <= thr32 + e1 per query column and emits the (database tile, query tile) pairs that pass:
//     8 bytes per pair instead of 16 per surviving unit;
//   pass 2, knn_refine16b: the three-term keys of the listed tile pairs only, in knn_sweep16b's own MFMA order
//     (the probed accumulation bound applies unchanged), tested against thr32; survivors go to the entry pool.
// The set of survivors is the one knn_sweep16b<mode 1> writes: whatever passes there passes the coarse test.
// ===========================================================================================================
struct CoarsePair { unsigned int tile, qtile; };

// WPS: wavefronts per SIMD the instance is built for.  Measured (profiles/r03_*): kernel cycles = matrix-busy cycles
// + 4 x vector instructions, with one wavefront per SIMD (eight database tiles each) and with two (four each, twice the
// per-query-tile overhead): the vector instructions of the tests do not hide behind the matrix pipe either way, so the
// instance with fewer of them per MFMA is the default (SNK_COARSE_WPS=2 selects the other one).
template <int NTC, int KB, int WPS>
__global__ void __launch_bounds__(256, WPS)
knn_coarse16b(const u32x4 *__restrict__ A16, const u32x4 *__restrict__ B16, const float *__restrict__ thr1, int nQT,
              int64_t n_tiles, int64_t n_slabs, unsigned int *__restrict__ slab_counter, int qsplit, int64_t n_main_slabs,
              int qsplit_tail, CoarsePair *__restrict__ pairs, unsigned int *__restrict__ pair_ctl, unsigned int pair_cap)
{
    // independent MFMA chains per step, two accumulator sets that alternate by step parity.  Register homes are forced
    // (Makefile: -amdgpu-mfma-vgpr-form for this file; the "+a" pins below): the resident database pieces in the
    // accumulation registers, which the matrix instruction reads its A operand from directly, the results in the
    // architected registers, where the vector unit tests them.  Left to itself the allocator did the opposite and paid
    // sixteen v_accvgpr_read per tested tile: as many vector instructions as the matrix pipe was busy cycles.
    constexpr int CH = (NTC >= 8) ? 4 : 2;
    constexpr int NSTEP = NTC / CH;
    static_assert(NTC % CH == 0 && NSTEP % 2 == 0, "the accumulator sets alternate by step parity, a tile has an even number of steps");
    __shared__ CoarsePair pstage[4][64];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int qcol = lane & 31;
    int pcount = 0;
    auto grab = [&]() -> int64_t {
        unsigned int v = 0;
        if (lane == 0) v = atomicAdd(slab_counter, 1u);
        return (int64_t)__builtin_amdgcn_readfirstlane(v);
    };
    auto flush_pairs = [&]() {
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(&pair_ctl[0], (unsigned int)pcount);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < pcount) {
            if (base + (unsigned int)lane < pair_cap) pairs[base + lane] = pstage[wv][lane];
            else pair_ctl[1] = 1u;                     // the list is full: the caller's other path serves the call
        }
        pcount = 0;
    };
    auto emit = [&](unsigned int tile, unsigned int qtile) {
        if (lane == 0) pstage[wv][pcount] = CoarsePair{tile, qtile};
        if (++pcount == 64) flush_pairs();
    };
    auto mfma = [](const u32x4 &a, const u32x4 &b, f16acc c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };

    const int64_t n_main_items = n_main_slabs * qsplit;
    const int64_t n_items = n_main_items + (n_slabs - n_main_slabs) * qsplit_tail;
    int64_t item = grab();
    while (item < n_items) {
        const int64_t item_next = grab();
        const bool tail = item >= n_main_items;
        const int qs = tail ? qsplit_tail : qsplit;
        const int64_t rel = tail ? item - n_main_items : item;
        const int64_t w = (tail ? n_main_slabs : 0) + rel / qs;
        const int part = (int)(rel % qs);
        const int qt_lo = (int)(((int64_t)nQT * part) / qs);
        const int qt_hi = (int)(((int64_t)nQT * (part + 1)) / qs);
        const int n_t = qt_hi - qt_lo;

        // hi pieces of this slab's database tiles: resident in registers (tiles past the operand repeat the last one
        // and are never reported)
        u32x4 ah[NTC][KB];
        const int64_t tile0 = w * NTC;
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            const int64_t t = tile0 + nt < n_tiles ? tile0 + nt : n_tiles - 1;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) ah[nt][kb] = A16[((t * KB + kb) * 2 + 0) * 64 + lane];
        }
        // the resident database pieces live in the ACCUMULATION registers (the matrix instruction reads its A operand from
        // there as well) so that the results, which the vector unit tests, can stay in the architected ones: left to itself
        // the allocator did the opposite and paid sixteen v_accvgpr_read per tested tile
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) asm volatile("" : "+a"(ah[nt][kb]));
        const int n_valid = (int)(n_tiles - tile0 < NTC ? n_tiles - tile0 : NTC);

        // query tiles: hi pieces and the coarse threshold of this lane's query column, three buffers (two tiles ahead)
        u32x4 xq[3][KB];
        float thq[3];
        auto load_q = [&](int t, u32x4 (&x)[KB], float &th) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) x[kb] = B16[(((int64_t)t * KB + kb) * 2 + 0) * 64 + lane];
            th = thr1[t * 32 + qcol];
        };
        auto next_q = [&](int t) { return t + 1 == qt_hi ? qt_lo : t + 1; };
        int qt = qt_lo + (int)((w * 3) % n_t);
        int qt1 = next_q(qt), qt2 = next_q(qt1);
        load_q(qt, xq[0], thq[0]);
        load_q(qt1, xq[1], thq[1]);

        f16acc S0[CH], S1[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { S0[j][r] = FLT_MAX; S1[j][r] = FLT_MAX; }
        float th_prev = -FLT_MAX;
        int qt_prev = qt;

        // one pending result: the smallest of the 16 rows this lane holds of database tile pt against its query column
        auto test = [&](const f16acc &res, int pt, float pth, int pq) {
            float m;
            // (builtins: the compiler pairs them into v_min3_f32 and knows the wait states behind a matrix instruction)
            m = __builtin_fminf(__builtin_fminf(res[0], res[1]), res[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) m = __builtin_fminf(__builtin_fminf(m, res[r]), res[r + 1]);
            m = __builtin_fminf(m, res[15]);
            if (__any(m <= pth) && pt < n_valid) emit((unsigned int)(tile0 + pt), (unsigned int)pq);
        };
        // one step: CH database tiles (first: t0) against the query tile in x into `cur`; the CH results in `prev`
        // (database tiles from pt0 of query tile pq, threshold pth) are tested in the MFMA shadows
        auto step = [&](f16acc (&cur)[CH], const f16acc (&prev)[CH], int t0, const u32x4 (&x)[KB], int pt0, float pth, int pq) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    if (kb == 0) {
                        f16acc z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                        cur[j] = mfma(ah[t0 + j][kb], x[kb], z);
                    } else cur[j] = mfma(ah[t0 + j][kb], x[kb], cur[j]);
                }
                if (kb < CH) test(prev[kb], pt0 + kb, pth, pq);
            }
            // the results are tested by the vector unit: architected registers
#pragma unroll
            for (int j = 0; j < CH; ++j) asm volatile("" : "+v"(cur[j]));
            if (KB < CH) {
#pragma unroll
                for (int j = KB; j < CH; ++j) test(prev[j], pt0 + j, pth, pq);
            }
        };
        auto tile_body = [&](u32x4 (&x)[KB], float th, u32x4 (&xload)[KB], float &thload, int t_load) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) asm volatile("" : "+v"(x[kb]));
            load_q(t_load, xload, thload);               // two tiles ahead
            // step st fills S[st & 1] and tests the step before it (the last step of the previous query tile for st = 0)
            step(S0, S1, 0, x, (NSTEP - 1) * CH, th_prev, qt_prev);
#pragma unroll
            for (int st = 1; st < NSTEP; ++st) {
                if (st & 1) step(S1, S0, st * CH, x, (st - 1) * CH, th, qt);
                else step(S0, S1, st * CH, x, (st - 1) * CH, th, qt);
            }
            th_prev = th;
            qt_prev = qt;
            qt = qt1; qt1 = qt2; qt2 = next_q(qt2);
        };
        for (int it = 0; it < n_t; it += 3) {
            tile_body(xq[0], thq[0], xq[2], thq[2], qt2);
            if (it + 1 < n_t) tile_body(xq[1], thq[1], xq[0], thq[0], qt2);
            if (it + 2 < n_t) tile_body(xq[2], thq[2], xq[1], thq[1], qt2);
        }
        // drain: the last step's results
#pragma unroll
        for (int j = 0; j < CH; ++j) test(S1[j], (NSTEP - 1) * CH + j, th_prev, qt_prev);
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) S1[j][r] = FLT_MAX;
        item = item_next;
    }
    if (pcount) flush_pairs();
}

// pass 2: the three-term keys of the listed tile pairs, tested against thr32; survivors to the entry pool
template <int KB, int TERMS>
__global__ void __launch_bounds__(256, (KB <= 4 ? 2 : 1))        // two operand sets of wider rows need the whole register file
knn_refine16b(const u32x4 *__restrict__ A16, const u32x4 *__restrict__ B16, const float *__restrict__ thr32,
              const CoarsePair *__restrict__ pairs, const unsigned int *__restrict__ pair_ctl, unsigned int pair_cap,
              PoolEntry16 *__restrict__ pool, unsigned int *__restrict__ pool_ctl, int *__restrict__ chunk_fill, int max_chunks,
              int pool_chunk)
{
    constexpr int STAGE_CAP = 1024 + 64;
    __shared__ PoolEntry16 stage[4][STAGE_CAP];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int qcol = lane & 31;
    unsigned int n_pairs = pair_ctl[0];
    if (n_pairs > pair_cap) { n_pairs = pair_cap; if (threadIdx.x == 0 && blockIdx.x == 0) pool_ctl[1] = 1u; }   // list overflow: status bit 4
    int chunk_id = -1, cused = pool_chunk, lcount = 0;
    auto new_chunk = [&]() {
        if (chunk_id >= 0 && lane == 0) chunk_fill[chunk_id] = cused;
        unsigned int c = 0;
        if (lane == 0) c = atomicAdd(&pool_ctl[0], 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        if ((int)c >= max_chunks) { if (lane == 0) pool_ctl[1] = 1u; chunk_id = -1; }
        else chunk_id = (int)c;
        cused = 0;
    };
    auto flush_stage = [&]() {
        if (cused + lcount > pool_chunk) new_chunk();
        if (chunk_id >= 0)
            for (int e = lane; e < lcount; e += 64)
                pool[(int64_t)chunk_id * pool_chunk + cused + e] = stage[wv][e];
        cused += lcount;
        lcount = 0;
    };
    auto mfma = [](const u32x4 &a, const u32x4 &b, f16acc c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    const unsigned int wave_id = blockIdx.x * 4u + (unsigned int)wv, wave_stride = gridDim.x * 4u;
    // operands of a pair: database tile (hi, lo), query tile (hi, lo).  The database tiles alternate between two register
    // sets (the next pair's is requested before this one's arithmetic); the QUERY tile is loaded only when it changes -- the
    // ball pass lists the pairs of one (stretch of the database, query tile) next to each other, so runs of pairs share it
    // (every pair re-reading both tiles made the stage L2-bound: 8 x the algorithmic bytes, profiles/r03_traffic_filter.json)
    // -- into the set that is not in use, so that its load too lies behind arithmetic.
    u32x4 a0[KB][2], a1[KB][2], qA[KB][2], qB[KB][2];
    float thA = 0.f, thB = 0.f;
    CoarsePair p0{0u, 0u}, p1{0u, 0u};
    auto load_a = [&](unsigned int i, CoarsePair &pr, u32x4 (&a)[KB][2]) {
        const unsigned int k = i < n_pairs ? i : n_pairs - 1u;
        pr = pairs[k];
        pr.tile = __builtin_amdgcn_readfirstlane(pr.tile); pr.qtile = __builtin_amdgcn_readfirstlane(pr.qtile);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) a[kb][pc] = A16[(((int64_t)pr.tile * KB + kb) * 2 + pc) * 64 + lane];
    };
    auto load_q = [&](unsigned int qtile, u32x4 (&b)[KB][2], float &th) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) b[kb][pc] = B16[(((int64_t)qtile * KB + kb) * 2 + pc) * 64 + lane];
        th = thr32[qtile * 32u + (unsigned int)qcol];
    };
    auto work = [&](const CoarsePair &pr, const u32x4 (&a)[KB][2], const u32x4 (&b)[KB][2], float pth) {
        // knn_sweep16b's order: per k-block hi.hi, hi(db).lo(query), lo(db).hi(query) [, lo.lo]; chains of one 64-column
        // chunk (4 TERMS MFMAs through C), the chunks' sums added in float32
        constexpr int CM = 4 * TERMS, NM = TERMS * KB;
        f16acc acc, part;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int kb = m / TERMS, term = m % TERMS;
            const u32x4 &av = (term & 2) ? a[kb][1] : a[kb][0];
            const u32x4 &bv = (term & 1) ? b[kb][1] : b[kb][0];
            if (m % CM == 0) {
                f16acc z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                part = mfma(av, bv, z);
            } else part = mfma(av, bv, part);
            if (m % CM == CM - 1 || m == NM - 1) {
                if (m < CM) acc = part;
                else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] += part[r];
                }
            }
        }
        if (lcount > STAGE_CAP - 1024) flush_stage();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float key = acc[r];
            const bool pass = key <= pth;
            const unsigned long long mm = __ballot(pass);
            if (mm) {
                if (pass) {
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mm, 0u));
                    PoolEntry16 en;
                    en.key = key; en.unused = 0;
                    en.idx = (int)(pr.tile * 32u) + crow32(lane, r);
                    en.row = (int)(pr.qtile * 32u) + qcol;
                    stage[wv][lcount + rank] = en;
                }
                lcount += __popcll(mm);
            }
        }
    };
    // a wavefront takes runs of BL consecutive pairs: the coarse sweep lists the pairs of one query tile (and of
    // neighbouring database tiles) next to each other, so a run re-reads operand tiles this compute unit has just had
    // (pairs dealt out one by one took 0.46 ms at B*, L2-bound: 16 KB of operands per twelve MFMAs)
    constexpr unsigned int BL = 8u;
    auto idx = [&](unsigned int j) { return (wave_id + (j / BL) * wave_stride) * BL + (j % BL); };
    if (idx(0u) < n_pairs) {
        load_a(idx(0u), p0, a0);
        load_q(p0.qtile, qA, thA);
        if constexpr (KB <= 4) {
            int cur = 0;                                       // which query set holds the tile of the pair in work (uniform)
            unsigned int cur_qt = p0.qtile;
            // one pair out of (a, p) while the next pair's database tile -- and its query tile, if another -- are requested
            auto pair_step = [&](const CoarsePair &p, const u32x4 (&a)[KB][2], unsigned int i_next, CoarsePair &pn, u32x4 (&an)[KB][2]) {
                load_a(i_next, pn, an);
                const bool sw = pn.qtile != cur_qt;            // uniform
                if (cur == 0) {
                    if (sw) load_q(pn.qtile, qB, thB);
                    work(p, a, qA, thA);
                } else {
                    if (sw) load_q(pn.qtile, qA, thA);
                    work(p, a, qB, thB);
                }
                if (sw) { cur ^= 1; cur_qt = pn.qtile; }
            };
            for (unsigned int j = 0u;; j += 2u) {
                const unsigned int i1 = idx(j + 1u);
                pair_step(p0, a0, i1, p1, a1);
                if (i1 >= n_pairs) break;
                const unsigned int i2 = idx(j + 2u);
                pair_step(p1, a1, i2, p0, a0);
                if (i2 >= n_pairs) break;
            }
        } else {
            // wider rows: the two-way choice of the query set costs registers these instances do not have (KB = 12 spilled
            // 161): every pair brings both tiles, into the set of its parity
            for (unsigned int j = 0u;; j += 2u) {
                const unsigned int i1 = idx(j + 1u);
                load_a(i1, p1, a1);
                load_q(p1.qtile, qB, thB);
                work(p0, a0, qA, thA);
                if (i1 >= n_pairs) break;
                const unsigned int i2 = idx(j + 2u);
                load_a(i2, p0, a0);
                load_q(p0.qtile, qA, thA);
                work(p1, a1, qB, thB);
                if (i2 >= n_pairs) break;
            }
        }
    }
    if (lcount) flush_stage();
    if (chunk_id >= 0 && lane == 0) chunk_fill[chunk_id] = cused;
}

// developer switch (SNK_COARSE_WPS=1|2, read once): wavefronts per SIMD of the one-chunk coarse sweep
static int coarse_wps()
{
    static int v = 0;
    if (!v) { const char *e = getenv("SNK_COARSE_WPS"); v = (e && atoi(e) == 2) ? 2 : 1; }
    return v;
}

// ===========================================================================================================
// Pass 0 of the filter, in front of the coarse sweep: a bound per (database tile, query row) from the tile's BALL.
// A tile is 32 consecutive units -- in a speech database consecutive frames, close to each other -- with centre c
// (mean of its rows, float64) and radius r = max_i ||f_i - c||.  For every unit f of the tile ||q - f|| >= ||q - c|| - r,
// so a tile can hold a unit inside the filter's threshold only if ||q - c|| <= sqrt(D2max(q)) + r, where D2max(q) =
// thr32 + eps + ||q||^2 bounds the squared distance of anything the three-term test would let through.  The centres
// are 1 / 32 of the database: their three-term keys (knn_sweep16b's operands and MFMA order, so its error bound eps
// applies; ||c|| <= Fmax) cost a thirtieth of the coarse sweep, and the test
//        key~(c) <= (tq + r)^2 - nq,     tq = sqrt(D2max(q)) rounded up,  nq = ||q||^2 - eps rounded down
// lists the (tile, query tile) pairs for knn_refine16b directly.  Either list is a superset of what the three-term test
// passes, so which pass writes it is a matter of speed only: the engine switches a voice to the coarse sweep once the
// ball pass has listed more than coarse_gate_fraction of all pairs (tiles that are not compact; api_knn.hip knn_device).
// ===========================================================================================================
__global__ void __launch_bounds__(256)
build_tile_balls_kernel(const double *__restrict__ Fw, int64_t N, int Dt, int Dpad, int64_t n_tiles, double *__restrict__ C,
                        double *__restrict__ cnorm, float *__restrict__ rad, const int32_t *__restrict__ perm)
{
    // one wavefront per tile: lane = column (strided), rows one after the other
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int64_t r0 = tile * 32;
    const int n = (int)(N - r0 < 32 ? (N - r0 > 0 ? N - r0 : 0) : 32);
    double cn = 0.0;
    for (int c = lane; c < Dpad; c += 64) {
        double m = 0.0;
        if (c < Dt && n > 0) {
            for (int i = 0; i < n; ++i) m += Fw[unit_at(perm, r0 + i, N) * Dpad + c];
            m /= (double)n;
        }
        C[tile * Dpad + c] = m;
        cn += m * m;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cn += __shfl_xor(cn, off, 64);
    double rmax2 = 0.0;
    for (int i = 0; i < n; ++i) {
        double d2 = 0.0;
        for (int c = lane; c < Dt; c += 64) {
            const double d = Fw[unit_at(perm, r0 + i, N) * Dpad + c] - C[tile * Dpad + c];        // (this lane's own columns: written above)
            d2 += d * d;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) d2 += __shfl_xor(d2, off, 64);
        rmax2 = d2 > rmax2 ? d2 : rmax2;
    }
    if (lane == 0) {
        cnorm[tile] = n > 0 ? cn : __builtin_inf();                            // an empty tile: its key never passes
        float r = (float)(sqrt(rmax2) * (1.0 + 1e-9));
        if ((double)r < sqrt(rmax2) * (1.0 + 1e-9)) r = nextafterf(r, FLT_MAX);
        rad[tile] = n > 0 ? r : 0.f;
    }
}

void launch_build_tile_balls(const double *Fw, int64_t N, int Dt, int Dpad, int64_t n_tiles, double *C, double *cnorm, float *rad,
                             hipStream_t s, const int32_t *perm)
{
    hipLaunchKernelGGL(build_tile_balls_kernel, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, s, Fw, N, Dt, Dpad, n_tiles, C,
                       cnorm, rad, perm);
}

// One level up: the ball of 32 consecutive tiles (1 024 units) -- centre C = mean of its units, radius <= max_t (||c_t - C|| + r_t).
// The ball pass tests these first (1 / 32 of its products) and visits a (centre tile, query tile) block only where the super
// ball passes for some row of the query tile: the same inequality, the same operands and key bound one level up.
__global__ void __launch_bounds__(256)
build_super_balls_kernel(const double *__restrict__ C, const float *__restrict__ rad, int64_t N, int64_t n_tiles, int Dt, int Dpad,
                         int64_t n_super, double *__restrict__ C2, double *__restrict__ cnorm2, float *__restrict__ rad2)
{
    const int lane = threadIdx.x & 63;
    const int64_t sidx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sidx >= n_super) return;
    const int64_t t0 = sidx * 32;
    const int nt = (int)(n_tiles - t0 < 32 ? (n_tiles - t0 > 0 ? n_tiles - t0 : 0) : 32);
    double wsum = 0.0;
    for (int i = 0; i < nt; ++i) { const int64_t u = N - (t0 + i) * 32; wsum += (double)(u < 32 ? (u > 0 ? u : 0) : 32); }
    double cn = 0.0;
    for (int c = lane; c < Dpad; c += 64) {
        double m = 0.0;
        if (c < Dt && wsum > 0.0) {
            for (int i = 0; i < nt; ++i) {
                const int64_t u = N - (t0 + i) * 32;
                m += C[(t0 + i) * Dpad + c] * (double)(u < 32 ? (u > 0 ? u : 0) : 32);
            }
            m /= wsum;
        }
        C2[sidx * Dpad + c] = m;
        cn += m * m;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cn += __shfl_xor(cn, off, 64);
    double rmax = 0.0;
    for (int i = 0; i < nt; ++i) {
        if (N - (t0 + i) * 32 <= 0) continue;
        double d2 = 0.0;
        for (int c = lane; c < Dt; c += 64) {
            const double d = C[(t0 + i) * Dpad + c] - C2[sidx * Dpad + c];       // (this lane's own columns: written above)
            d2 += d * d;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) d2 += __shfl_xor(d2, off, 64);
        const double r = sqrt(d2) * (1.0 + 1e-9) + (double)rad[t0 + i];
        rmax = r > rmax ? r : rmax;
    }
    if (lane == 0) {
        cnorm2[sidx] = wsum > 0.0 ? cn : __builtin_inf();
        float r = (float)(rmax * (1.0 + 1e-9));
        if ((double)r < rmax * (1.0 + 1e-9)) r = nextafterf(r, FLT_MAX);
        rad2[sidx] = wsum > 0.0 ? r : 0.f;
    }
}

void launch_build_super_balls(const double *C, const float *rad, int64_t N, int64_t n_tiles, int Dt, int Dpad, int64_t n_super, double *C2,
                              double *cnorm2, float *rad2, hipStream_t s)
{
    hipLaunchKernelGGL(build_super_balls_kernel, dim3((unsigned)((n_super + 3) / 4)), dim3(256), 0, s, C, rad, N, n_tiles, Dt, Dpad, n_super,
                       C2, cnorm2, rad2);
}

// per query row: tq and nq of the ball test (see above) from the row's filter threshold, key bound and norm
__global__ void ball_query_terms_kernel(const float *__restrict__ thr32, const double *__restrict__ eps, const double *__restrict__ qnorm,
                                        int64_t T, int64_t T32, float *__restrict__ tq, float *__restrict__ nq)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= T32) return;
    float t = 0.f, n = FLT_MAX;                                               // padding rows: nothing passes
    if (row < T) {
        const float th = thr32[row];
        if (th >= FLT_MAX) { t = FLT_MAX; n = 0.f; }                          // no bound for this row: every tile
        else if (th > -FLT_MAX) {
            const double d2max = (double)th + eps[row] + qnorm[row];
            if (d2max >= 0.0) {
                const double td = sqrt(d2max) * (1.0 + 1e-9);
                t = (float)td;
                if ((double)t < td) t = nextafterf(t, FLT_MAX);
                const double nd = qnorm[row] - eps[row];
                n = (float)nd;
                if ((double)n > nd) n = nextafterf(n, -FLT_MAX);
            }
        }
    }
    tq[row] = t; nq[row] = n;
}

void launch_ball_query_terms(const float *thr32, const double *eps, const double *qnorm, int64_t T, int64_t T32, float *tq, float *nq,
                             hipStream_t s)
{
    hipLaunchKernelGGL(ball_query_terms_kernel, dim3((unsigned)((T32 + 255) / 256)), dim3(256), 0, s, thr32, eps, qnorm, T, T32, tq, nq);
}

// centres (operand C16: tiles of 32 centres, hi / lo pieces like the database) against all query tiles
// BITS: the super-ball pass -- instead of pairs, a bit per (row of the centre operand = super ball, query tile) in `mask`;
// the tile-level pass is handed that mask as `visit` and skips the (centre tile, query tile) blocks whose bit is clear
template <int KB, int TERMS, bool BITS>
__global__ void __launch_bounds__(256, 2)
knn_balls16b(const u32x4 *__restrict__ C16, const u32x4 *__restrict__ B16, const float *__restrict__ rad, const float *__restrict__ tq,
             const float *__restrict__ nq, int nQT, int64_t n_tiles, int64_t n_ctiles, int qsplit, CoarsePair *__restrict__ pairs,
             unsigned int *__restrict__ pair_ctl, unsigned int pair_cap, unsigned int *__restrict__ mask, const unsigned int *__restrict__ visit)
{
    const int nQTw = (nQT + 31) >> 5;
    __shared__ CoarsePair pstage[4][96];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int qcol = lane & 31;
    int pcount = 0;
    auto flush_pairs = [&]() {
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(&pair_ctl[0], (unsigned int)pcount);
        base = __builtin_amdgcn_readfirstlane(base);
        for (int e = lane; e < pcount; e += 64) {
            if (base + (unsigned int)e < pair_cap) pairs[base + e] = pstage[wv][e];
            else pair_ctl[1] = 1u;
        }
        pcount = 0;
    };
    auto mfma = [](const u32x4 &a, const u32x4 &b, f16acc c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    const int64_t n_items = n_ctiles * qsplit;
    for (int64_t item = (int64_t)blockIdx.x * 4 + wv; item < n_items; item += (int64_t)gridDim.x * 4) {
        const int64_t ct = item / qsplit;
        const int part = (int)(item % qsplit);
        const int qt_lo = (int)(((int64_t)nQT * part) / qsplit), qt_hi = (int)(((int64_t)nQT * (part + 1)) / qsplit);
        // the query tiles of this block's range that are to be visited: all of them, or the set bits of the super pass's row
        const unsigned int *const vrow = (!BITS && visit) ? visit + ct * nQTw : nullptr;
        auto next_qt = [&](int qq) -> int {
            if (!vrow) return qq < qt_hi ? qq : qt_hi;
            while (qq < qt_hi) {
                const unsigned int w = __builtin_amdgcn_readfirstlane(vrow[qq >> 5]) >> (qq & 31);
                if (w) { qq += __builtin_ctz(w); return qq < qt_hi ? qq : qt_hi; }
                qq = (qq | 31) + 1;
            }
            return qt_hi;
        };
        int q0 = next_qt(qt_lo);
        if (q0 >= qt_hi) continue;
        u32x4 a[KB][2];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) a[kb][pc] = C16[((ct * KB + kb) * 2 + pc) * 64 + lane];
        float rv[16];                                          // radius of the 16 rows this lane holds results of
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t t = ct * 32 + crow32(lane, r);
            rv[r] = t < n_tiles ? rad[t] : __builtin_nanf("");
        }
        u32x4 b0[KB][2], b1[KB][2];
        float t0 = 0.f, n0 = 0.f, t1 = 0.f, n1 = 0.f;
        auto load_q = [&](int t, u32x4 (&b)[KB][2], float &tt, float &nn) {
            const int tc = t < qt_hi ? t : qt_hi - 1;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) b[kb][pc] = B16[(((int64_t)tc * KB + kb) * 2 + pc) * 64 + lane];
            tt = tq[tc * 32 + qcol]; nn = nq[tc * 32 + qcol];
        };
        auto work = [&](int qt, const u32x4 (&b)[KB][2], float tt, float nn) {
            constexpr int CM = 4 * TERMS, NM = TERMS * KB;
            f16acc acc, part;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int kb = m / TERMS, term = m % TERMS;
                const u32x4 &av = (term & 2) ? a[kb][1] : a[kb][0];
                const u32x4 &bv = (term & 1) ? b[kb][1] : b[kb][0];
                if (m % CM == 0) {
                    f16acc z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                    part = mfma(av, bv, z);
                } else part = mfma(av, bv, part);
                if (m % CM == CM - 1 || m == NM - 1) {
                    if (m < CM) acc = part;
                    else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] += part[r];
                    }
                }
            }
            if (pcount > 96 - 32) flush_pairs();
            // key~(c) <= (tq + r)^2 - nq, the right side rounded up: (tq + r)^2 (1 + 2^-21) - nq + 2^-21 |nq| + 1e-30 (a relative
            // 2^-21 of both terms covers the three float32 roundings); the column's part is taken once per query tile.  A row
            // past the database carries a NaN radius: its comparison is false.
            const float kq = __builtin_fmaf(4.76837158203125e-07f, __builtin_fabsf(nn), -nn) + 1e-30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sr = rv[r] + tt;
                const bool pass = acc[r] <= __builtin_fmaf(sr * sr, 1.f + 4.76837158203125e-07f, kq);
                const unsigned long long mm = __ballot(pass);
                if (mm) {
                    // lanes 0..31 hold row crow32(0, r), lanes 32..63 row crow32(32, r): one pair per row with any passing column
                    const unsigned int lo = (unsigned int)mm, hi = (unsigned int)(mm >> 32);
                    if (BITS) {
                        if (lane == 0) {
                            if (lo) atomicOr(&mask[(ct * 32 + crow32(0, r)) * nQTw + (qt >> 5)], 1u << (qt & 31));
                            if (hi) atomicOr(&mask[(ct * 32 + crow32(32, r)) * nQTw + (qt >> 5)], 1u << (qt & 31));
                        }
                    } else {
                        if (lo) { if (lane == 0) pstage[wv][pcount] = CoarsePair{(unsigned int)(ct * 32 + crow32(0, r)), (unsigned int)qt}; ++pcount; }
                        if (hi) { if (lane == 0) pstage[wv][pcount] = CoarsePair{(unsigned int)(ct * 32 + crow32(32, r)), (unsigned int)qt}; ++pcount; }
                    }
                }
            }
        };
        load_q(q0, b0, t0, n0);
        for (;;) {
            const int q1 = next_qt(q0 + 1);
            load_q(q1, b1, t1, n1);                            // (clamped inside: a load past the range is never used)
            work(q0, b0, t0, n0);
            if (q1 >= qt_hi) break;
            const int q2 = next_qt(q1 + 1);
            load_q(q2, b0, t0, n0);
            work(q1, b1, t1, n1);
            if (q2 >= qt_hi) break;
            q0 = q2;
        }
    }
    if (!BITS && pcount) flush_pairs();
}

bool launch_knn_balls16b(int terms, int dch, int grid_cus, const void *C16, const void *B16, const float *rad, const float *tq,
                         const float *nq, int64_t T32, int64_t n_tiles, void *pairs, unsigned int *pair_ctl, unsigned int pair_cap,
                         hipStream_t s, unsigned int *mask_out, const unsigned int *visit)
{
    // mask_out: the super pass (bits instead of pairs; the caller zeroed ceil(T32 / 1024) words per row of the operand);
    // visit: the tile pass, restricted to the blocks the super pass marked
    const int nQT = (int)(T32 / 32);
    const int64_t n_ctiles = (n_tiles + 31) / 32;
    int qsplit = 1;
    while (n_ctiles * qsplit < 8 * (int64_t)grid_cus && qsplit * 2 <= nQT && qsplit < 64) qsplit *= 2;
    int64_t blocks = (n_ctiles * qsplit + 3) / 4;
    if (blocks > 2 * (int64_t)grid_cus) blocks = 2 * (int64_t)grid_cus;
#define SNK_B16(KB_, TERMS_)                                                                                         \
    {                                                                                                                \
        if (mask_out)                                                                                                \
            hipLaunchKernelGGL((knn_balls16b<KB_, TERMS_, true>), dim3((unsigned)blocks), dim3(256), 0, s, (const u32x4 *)C16, \
                               (const u32x4 *)B16, rad, tq, nq, nQT, n_tiles, n_ctiles, qsplit, (CoarsePair *)pairs, pair_ctl, pair_cap, mask_out, visit); \
        else                                                                                                         \
            hipLaunchKernelGGL((knn_balls16b<KB_, TERMS_, false>), dim3((unsigned)blocks), dim3(256), 0, s, (const u32x4 *)C16, \
                               (const u32x4 *)B16, rad, tq, nq, nQT, n_tiles, n_ctiles, qsplit, (CoarsePair *)pairs, pair_ctl, pair_cap, mask_out, visit); \
    }
    if (dch == 1) { if (terms == 4) SNK_B16(4, 4) else SNK_B16(4, 3) }
    else if (dch == 2) { if (terms == 4) SNK_B16(8, 4) else SNK_B16(8, 3) }
    else if (dch == 3) { if (terms == 4) SNK_B16(12, 4) else SNK_B16(12, 3) }
    else return false;
#undef SNK_B16
    return true;
}

// ===========================================================================================================
// Stage A' of the thresholds: the K-th smallest key among the units of the few tiles nearest to a query row (no reference
// counterpart: the reference's KD-tree needs no threshold, script/synth_halfphone.py:1364).
// Stage A bounds the K-th nearest key by the K-th smallest key of a 1/16 sample, which lets about sixteen times K units
// through the filter; the bound the tiles' balls give, (||q - c|| + r)^2 - ||q||^2 for the ceil(K / 32)-th tile, was built
// and measured no better (its slack 2 r ||q - c|| is wider than the key window that holds 17 K units: lists 1785 -> 1575).
// Any K distinct units bound the K-th nearest key by their K-th smallest key -- so take units that are likely to BE the
// neighbours: in a speech database the neighbours of a frame are frames of the same stretch of signal, i.e. of the tiles
// (32 consecutive units) whose centres are nearest and of the tiles next to them.
//   S1 knn_scout16b       centres against query tiles (the ball pass's operands), query-stationary: per (query row, tile
//                         position mod 32, range of centre tiles) the smallest centre key with the centre tile's number in
//                         its low mantissa bits -- a hint, no bound: precision is irrelevant
//   S2 knn_scout_pick     per query tile: per group the best hint over its 32 rows (consecutive frames share their stretch of
//                         the database), the tiles of the SCOUT_SLOTS best groups
//   S3 knn_scout_keys16b  the three-term keys of those (tile, query tile) pairs -- knn_refine16b's arithmetic: the filter's
//                         own keys and error bound -- into gkeys[row][slot * 32 + unit]
// and the threshold kernel takes the K-th smallest of a row's keys + eps as a second bound (bound2).  Valid whatever the
// hint picks; where a row's tiles hold fewer than K units the row keeps stage A's bound.
// ===========================================================================================================
#define SCOUT_SLOTS 64

template <int KB, int TERMS>
__global__ void __launch_bounds__(256, 2)
knn_scout16b(const u32x4 *__restrict__ C16, const u32x4 *__restrict__ B16, int nQT, int64_t n_tiles, int64_t n_ctiles, int csplit,
             int ct_bits, float *__restrict__ smin)
{
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int qcol = lane & 31;
    auto mfma = [](const u32x4 &a, const u32x4 &b, f16acc c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    const int64_t n_items = (int64_t)nQT * csplit;
    const int Gb = csplit * 64;
    const unsigned int keep = ~((1u << ct_bits) - 1u);
    for (int64_t item = (int64_t)blockIdx.x * 4 + wv; item < n_items; item += (int64_t)gridDim.x * 4) {
        const int qt = (int)(item / csplit), part = (int)(item % csplit);
        // (ranges start at even centre tiles: the two operand buffers below then are the two parities)
        const int64_t c_lo = ((n_ctiles * part) / csplit) & ~(int64_t)1;
        const int64_t c_hi = part + 1 == csplit ? n_ctiles : (((n_ctiles * (part + 1)) / csplit) & ~(int64_t)1);
        u32x4 b[KB][2];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) b[kb][pc] = B16[(((int64_t)qt * KB + kb) * 2 + pc) * 64 + lane];
        // minima per (parity of the centre tile, result register): the tiles of one minimum are 64 tiles apart
        float mn0[16], mn1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { mn0[r] = FLT_MAX; mn1[r] = FLT_MAX; }
        u32x4 a0[KB][2], a1[KB][2];
        auto load_c = [&](int64_t ct, u32x4 (&a)[KB][2]) {
            const int64_t cc = ct < c_hi ? ct : c_hi - 1;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) a[kb][pc] = C16[((cc * KB + kb) * 2 + pc) * 64 + lane];
        };
        auto work = [&](int64_t ct, const u32x4 (&a)[KB][2], float (&mn)[16]) {
            constexpr int NM = TERMS * KB;
            f16acc acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int kb = m / TERMS, term = m % TERMS;
                acc = mfma((term & 2) ? a[kb][1] : a[kb][0], (term & 1) ? b[kb][1] : b[kb][0], acc);
            }
            const unsigned int tag = (unsigned int)ct;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = ct * 32 + crow32(lane, r) < n_tiles && __builtin_fabsf(acc[r]) < FLT_MAX;
                const float v = __uint_as_float((__float_as_uint(acc[r]) & keep) | tag);
                mn[r] = __builtin_fminf(mn[r], ok ? v : FLT_MAX);
            }
        };
        if (c_lo < c_hi) {
            load_c(c_lo, a0);
            for (int64_t ct = c_lo; ct < c_hi; ct += 2) {
                load_c(ct + 1, a1);
                work(ct, a0, mn0);
                if (ct + 1 < c_hi) {
                    load_c(ct + 2, a0);
                    work(ct + 1, a1, mn1);
                }
            }
        }
        float *const out = smin + ((int64_t)qt * 32 + qcol) * Gb + part * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) { out[crow32(lane, r)] = mn0[r]; out[32 + crow32(lane, r)] = mn1[r]; }
    }
}

// one wavefront per query tile: per group the smallest hint over the tile's rows, then the SCOUT_SLOTS best groups' tiles
// (groups are disjoint sets of tiles: the tiles of a list are distinct)
__global__ void __launch_bounds__(64)
knn_scout_pick_kernel(const float *__restrict__ smin, int Gb, int ct_bits, int64_t T, unsigned int *__restrict__ list)
{
    const int lane = threadIdx.x;
    const int64_t qt = blockIdx.x;
    const unsigned int tagm = (1u << ct_bits) - 1u;
    constexpr int PER = 8;                                    // Gb <= 512 groups: eight per lane
    // a group's score: the smallest EXCESS of its hint over the row's best hint, over the tile's rows -- every row has its
    // own key offset (-||q||^2) and its own landscape; the excess treats them alike
    float v[PER];
    unsigned int tl[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { v[k] = FLT_MAX; tl[k] = 0xffffffffu; }
    for (int i = 0; i < 32; ++i) {
        const int64_t row = qt * 32 + i;
        if (row >= T) break;                                  // uniform
        float x[PER], rm = FLT_MAX;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int g = lane + 64 * k;
            x[k] = g < Gb ? smin[row * Gb + g] : FLT_MAX;
            rm = __builtin_fminf(rm, x[k]);
        }
#pragma unroll
        for (int o = 1; o <= 32; o <<= 1) rm = __builtin_fminf(rm, __shfl_xor(rm, o, 64));
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float e = x[k] < FLT_MAX ? x[k] - rm : FLT_MAX;
            if (e < v[k]) { v[k] = e; tl[k] = (__float_as_uint(x[k]) & tagm) * 32u + (unsigned int)((lane + 64 * k) & 31); }
        }
    }
    for (int s = 0; s < SCOUT_SLOTS; ++s) {
        float m = v[0];
#pragma unroll
        for (int k = 1; k < PER; ++k) m = __builtin_fminf(m, v[k]);
#pragma unroll
        for (int o = 1; o <= 32; o <<= 1) m = __builtin_fminf(m, __shfl_xor(m, o, 64));
        // one holder of the minimum hands its tile over and leaves
        bool mine = false;
#pragma unroll
        for (int k = 0; k < PER; ++k) mine = mine || v[k] == m;
        const unsigned long long has = __ballot(mine && m < FLT_MAX);
        unsigned int tile = 0xffffffffu;
        if (has) {
            const int src = __builtin_ctzll(has);
            unsigned int t = 0xffffffffu;
            if (lane == src) {
                bool done = false;
#pragma unroll
                for (int k = 0; k < PER; ++k)
                    if (!done && v[k] == m) { t = tl[k]; v[k] = FLT_MAX; done = true; }
            }
            tile = (unsigned int)__shfl((int)t, src, 64);
        }
        if (lane == 0) list[qt * SCOUT_SLOTS + s] = tile;
    }
}

// one wavefront per (query tile, slot): the three-term keys of the pair, knn_refine16b's arithmetic
template <int KB, int TERMS>
__global__ void __launch_bounds__(256, 2)
knn_scout_keys16b(const u32x4 *__restrict__ A16, const u32x4 *__restrict__ B16, const unsigned int *__restrict__ list, int nQT,
                  float *__restrict__ gkeys)
{
    const int lane = threadIdx.x & 63, qcol = lane & 31;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (int64_t)nQT * SCOUT_SLOTS) return;
    const int64_t qt = item / SCOUT_SLOTS;
    const int slot = (int)(item % SCOUT_SLOTS);
    const unsigned int tile = __builtin_amdgcn_readfirstlane(list[item]);
    float *const out = gkeys + (qt * 32 + qcol) * (int64_t)(SCOUT_SLOTS * 32) + slot * 32;
    if (tile == 0xffffffffu) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[crow32(lane, r)] = FLT_MAX;
        return;
    }
    auto mfma = [](const u32x4 &a, const u32x4 &b, f16acc c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    u32x4 a[KB][2], b[KB][2];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            a[kb][pc] = A16[(((int64_t)tile * KB + kb) * 2 + pc) * 64 + lane];
            b[kb][pc] = B16[((qt * KB + kb) * 2 + pc) * 64 + lane];
        }
    // knn_sweep16b's order: per k-block hi.hi, hi(db).lo(query), lo(db).hi(query) [, lo.lo]; chains of one 64-column chunk,
    // the chunks' sums added in float32 (the key bound eps is for this order)
    constexpr int CM = 4 * TERMS, NM = TERMS * KB;
    f16acc acc, part;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int kb = m / TERMS, term = m % TERMS;
        const u32x4 &av = (term & 2) ? a[kb][1] : a[kb][0];
        const u32x4 &bv = (term & 1) ? b[kb][1] : b[kb][0];
        if (m % CM == 0) {
            f16acc z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
            part = mfma(av, bv, z);
        } else part = mfma(av, bv, part);
        if (m % CM == CM - 1 || m == NM - 1) {
            if (m < CM) acc = part;
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += part[r];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[crow32(lane, r)] = acc[r] == acc[r] ? acc[r] : FLT_MAX;
}

int knn_scout_groups(int64_t T32, int64_t n_tiles)
{
    const int nQT = (int)(T32 / 32);
    const int64_t n_ctiles = (n_tiles + 31) / 32;
    int csplit = 1;
    while ((int64_t)nQT * csplit < 2048 && csplit < 8 && 4 * csplit <= n_ctiles) csplit *= 2;
    return csplit * 64;
}
int knn_scout_keys_per_row() { return SCOUT_SLOTS * 32; }
size_t knn_scout_list_bytes(int64_t T32) { return (size_t)(T32 / 32) * SCOUT_SLOTS * sizeof(unsigned int); }

// smin: T32 x knn_scout_groups floats; list: knn_scout_list_bytes; gkeys: T32 x knn_scout_keys_per_row floats
bool launch_knn_scout16b(int terms, int dch, int grid_cus, const void *C16, const void *A16, const void *B16, int64_t T, int64_t T32,
                         int64_t n_tiles, float *smin, unsigned int *list, float *gkeys, hipStream_t s)
{
    const int nQT = (int)(T32 / 32);
    const int64_t n_ctiles = (n_tiles + 31) / 32;
    const int Gb = knn_scout_groups(T32, n_tiles), csplit = Gb / 64;
    int ct_bits = 1;
    while (((int64_t)1 << ct_bits) < n_ctiles) ++ct_bits;
    if (ct_bits > 20 || dch < 1 || dch > 3) return false;
    int64_t blocks = ((int64_t)nQT * csplit + 3) / 4;
    if (blocks > 2 * (int64_t)grid_cus) blocks = 2 * (int64_t)grid_cus;
    const unsigned kblocks = (unsigned)(((int64_t)nQT * SCOUT_SLOTS + 3) / 4);
#define SNK_SC16(KB_, TERMS_)                                                                                        \
    {                                                                                                                \
        hipLaunchKernelGGL((knn_scout16b<KB_, TERMS_>), dim3((unsigned)blocks), dim3(256), 0, s, (const u32x4 *)C16, \
                           (const u32x4 *)B16, nQT, n_tiles, n_ctiles, csplit, ct_bits, smin);                       \
        hipLaunchKernelGGL(knn_scout_pick_kernel, dim3((unsigned)nQT), dim3(64), 0, s, smin, Gb, ct_bits, T, list); \
        hipLaunchKernelGGL((knn_scout_keys16b<KB_, TERMS_>), dim3(kblocks), dim3(256), 0, s, (const u32x4 *)A16,     \
                           (const u32x4 *)B16, list, nQT, gkeys);                                                    \
    }
    if (dch == 1) { if (terms == 4) SNK_SC16(4, 4) else SNK_SC16(4, 3) }
    else if (dch == 2) { if (terms == 4) SNK_SC16(8, 4) else SNK_SC16(8, 3) }
    else { if (terms == 4) SNK_SC16(12, 4) else SNK_SC16(12, 3) }
#undef SNK_SC16
    return true;
}

bool knn_coarse16b_supported(int nt, int dch) { return (nt == 4 && dch == 1) || (nt == 2 && dch == 2) || (nt == 1 && dch == 3); }
size_t knn_coarse_pair_bytes() { return sizeof(CoarsePair); }

// pass 1 + pass 2 on stream s.  n_tiles: 32-row tiles of the database operand; pair_ctl: two device words (count, overflow),
// zeroed by the caller (knn_reset)
bool launch_knn_filter16c(int terms, int dch, int grid_cus, const void *A16, const void *B16, const float *thr32, const float *thr1,
                          int64_t T32, int64_t n_tiles, unsigned int *ctr, void *pairs, unsigned int *pair_ctl, unsigned int pair_cap,
                          void *pool, unsigned int *pool_ctl, int *chunk_fill, int max_chunks, int pool_chunk, hipStream_t s, bool run_coarse,
                          bool run_refine)
{
    const int nQT = (int)(T32 / 32);
    const int wps = (dch == 1 && coarse_wps() == 2) ? 2 : 1;
    const int ntc = dch == 1 ? (wps == 2 ? 4 : 8) : 4;        // hi pieces of 16 / 32 (dch 3: 48) k-blocks resident per wavefront
    const int64_t n_slabs = (n_tiles + ntc - 1) / ntc;
    const int64_t max_blocks = (int64_t)grid_cus * wps;
    int qsplit = 1;
    while (n_slabs * qsplit < 2 * 4 * max_blocks && qsplit * 2 <= nQT && qsplit < 8) qsplit *= 2;
    int64_t blocks = (n_slabs * qsplit + 3) / 4;
    if (blocks > max_blocks) blocks = max_blocks;
    int64_t n_main = n_slabs;
    int qtail = qsplit;
    sweep_tail_split(n_slabs, qsplit, blocks * 4, nQT, &n_main, &qtail);
#define SNK_C16(NTC_, KB_, WPS_)                                                                                    \
    hipLaunchKernelGGL((knn_coarse16b<NTC_, KB_, WPS_>), dim3((unsigned)blocks), dim3(256), 0, s, (const u32x4 *)A16, \
                       (const u32x4 *)B16, thr1, nQT, n_tiles, n_slabs, ctr, qsplit, n_main, qtail,                 \
                       (CoarsePair *)pairs, pair_ctl, pair_cap)
#define SNK_R16(KB_, TERMS_)                                                                                        \
    hipLaunchKernelGGL((knn_refine16b<KB_, TERMS_>), dim3((unsigned)((KB_ <= 4 ? 2 : 1) * grid_cus)), dim3(256), 0, s, \
                       (const u32x4 *)A16, (const u32x4 *)B16, thr32, (const CoarsePair *)pairs, pair_ctl, pair_cap, \
                       (PoolEntry16 *)pool, pool_ctl, chunk_fill, max_chunks, pool_chunk)
    // run_coarse false: the pair list is already there (the ball pass wrote it)
    // run_refine false: the coarse sweep alone, as a probe that only COUNTS the pairs it would list (pair_cap 0)
    if (dch == 1) { if (run_coarse) { if (wps == 2) SNK_C16(4, 4, 2); else SNK_C16(8, 4, 1); } if (!run_refine) return true; if (terms == 4) SNK_R16(4, 4); else SNK_R16(4, 3); return true; }
    if (dch == 2) { if (run_coarse) SNK_C16(4, 8, 1); if (!run_refine) return true; if (terms == 4) SNK_R16(8, 4); else SNK_R16(8, 3); return true; }
    if (dch == 3) { if (run_coarse) SNK_C16(4, 12, 1); if (!run_refine) return true; if (terms == 4) SNK_R16(12, 4); else SNK_R16(12, 3); return true; }
#undef SNK_C16
#undef SNK_R16
    return false;
}

// ===========================================================================================================
// Rows of 257 .. 512 columns (the doubled join rows of an epoch voice from train_halfphone, 2 x 151 columns, as a K-NN
// database: initialise_join_table_with_knn, script/active_learning_join.py:184-212; script/train_halfphone.py:263-266).
// A database tile's fragments of that many k-blocks do not stay in registers, so this is an ordinary blocked product:
// a workgroup of four wavefronts takes 4 database tiles x 4 query tiles; wavefront w keeps the accumulators of ITS database
// tile against the four query tiles (64 registers), loads its tile's pieces of a k-block from global memory and stages
// query tile w's pieces in LDS for everybody (double buffered, one barrier per k-block); 4 TERMS MFMAs per k-block and
// wavefront.  Same operands (build_db16b / prepare_queries16b: [tile][kb][piece][lane]), same MFMA order and chains (one
// 64-column chunk through C, the chunks' sums added in float32) as knn_sweep16b: its key bound eps applies unchanged.
// MODE 0: stage A on the sample operand (one group per (tile, lane half): gmin32[row][2 tile + half]);
// MODE 1: the filter -- keys under the row threshold go to the entry pool like knn_refine16b's.
// ===========================================================================================================
template <int MODE, int TERMS>
__global__ void __launch_bounds__(256, 2)
knn_wide16b(const u32x4 *__restrict__ A16, const u32x4 *__restrict__ B16, int KB, const float *__restrict__ thr32, int nQT,
            int64_t n_tiles, float *__restrict__ gmin32, int64_t G, PoolEntry16 *__restrict__ pool,
            unsigned int *__restrict__ pool_ctl, int *__restrict__ chunk_fill, int max_chunks, int pool_chunk)
{
    constexpr int STAGE_CAP = (MODE == 1) ? 1024 + 64 : 1;
    __shared__ PoolEntry16 stage[(MODE == 1) ? 4 : 1][STAGE_CAP];
    __shared__ u32x4 Qs[2][4][2][64];                          // [buffer][query tile of the group][hi, lo][lane]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qcol = lane & 31;
    int chunk_id = -1, cused = pool_chunk, lcount = 0;
    auto new_chunk = [&]() {
        if (chunk_id >= 0 && lane == 0) chunk_fill[chunk_id] = cused;
        unsigned int c = 0;
        if (lane == 0) c = atomicAdd(&pool_ctl[0], 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        if ((int)c >= max_chunks) { if (lane == 0) pool_ctl[1] = 1u; chunk_id = -1; }
        else chunk_id = (int)c;
        cused = 0;
    };
    auto flush_stage = [&]() {
        if (cused + lcount > pool_chunk) new_chunk();
        if (chunk_id >= 0)
            for (int e = lane; e < lcount; e += 64)
                pool[(int64_t)chunk_id * pool_chunk + cused + e] = stage[wv][e];
        cused += lcount;
        lcount = 0;
    };
    auto mfma = [](const u32x4 &a, const u32x4 &b, f16acc c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    const int64_t n_tg = (n_tiles + 3) / 4;
    const int n_qg = (nQT + 3) / 4;
    const int64_t n_items = n_tg * n_qg;
    // consecutive items of a workgroup share the database tiles (query groups innermost)
    for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int64_t tg = item / n_qg;
        const int qg = (int)(item % n_qg);
        int64_t tile = tg * 4 + wv;
        const bool tile_ok = tile < n_tiles;
        if (!tile_ok) tile = n_tiles - 1;                      // (loads stay in range; nothing of it is kept)
        int qt_mine = qg * 4 + wv;
        if (qt_mine >= nQT) qt_mine = nQT - 1;
        f16acc acc[4], part[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[j][r] = 0.0f; part[j][r] = 0.0f; }
        u32x4 ah = A16[((tile * KB + 0) * 2 + 0) * 64 + lane], al = A16[((tile * KB + 0) * 2 + 1) * 64 + lane];
        u32x4 qh = B16[(((int64_t)qt_mine * KB + 0) * 2 + 0) * 64 + lane], ql = B16[(((int64_t)qt_mine * KB + 0) * 2 + 1) * 64 + lane];
        for (int kb = 0; kb < KB; ++kb) {
            const u32x4 ahc = ah, alc = al;
            Qs[kb & 1][wv][0][lane] = qh;
            Qs[kb & 1][wv][1][lane] = ql;
            if (kb + 1 < KB) {
                ah = A16[((tile * KB + kb + 1) * 2 + 0) * 64 + lane]; al = A16[((tile * KB + kb + 1) * 2 + 1) * 64 + lane];
                qh = B16[(((int64_t)qt_mine * KB + kb + 1) * 2 + 0) * 64 + lane]; ql = B16[(((int64_t)qt_mine * KB + kb + 1) * 2 + 1) * 64 + lane];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 bh = Qs[kb & 1][j][0][lane], bl = Qs[kb & 1][j][1][lane];
                // per k-block: hi.hi, hi(db).lo(query), lo(db).hi(query) [, lo.lo]
                part[j] = mfma(ahc, bh, part[j]);
                part[j] = mfma(ahc, bl, part[j]);
                part[j] = mfma(alc, bh, part[j]);
                if (TERMS == 4) part[j] = mfma(alc, bl, part[j]);
            }
            if ((kb & 3) == 3 || kb == KB - 1) {               // a 64-column chunk is complete: its sum joins the total
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc[j][r] += part[j][r]; part[j][r] = 0.0f; }
            }
        }
        __syncthreads();                                       // (the next item writes Qs[0])
        if (!tile_ok) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qt = qg * 4 + j;
            if (qt >= nQT) break;                               // uniform
            if (MODE == 0) {
                float gm = FLT_MAX;
#pragma unroll
                for (int r = 0; r < 16; ++r) gm = fminf(gm, acc[j][r]);
                gmin32[((int64_t)qt * 32 + qcol) * G + 2 * tile + (lane >> 5)] = gm;
            } else {
                const float pth = thr32[qt * 32 + qcol];
                if (lcount > STAGE_CAP - 1024) flush_stage();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float key = acc[j][r];
                    const bool pass = key <= pth;
                    const unsigned long long mm = __ballot(pass);
                    if (mm) {
                        if (pass) {
                            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mm, 0u));
                            PoolEntry16 en;
                            en.key = key; en.unused = 0;
                            en.idx = (int)(tile * 32) + crow32(lane, r);
                            en.row = qt * 32 + qcol;
                            stage[wv][lcount + rank] = en;
                        }
                        lcount += __popcll(mm);
                    }
                }
            }
        }
    }
    if (MODE == 1) {
        if (lcount) flush_stage();
        if (chunk_id >= 0 && lane == 0) chunk_fill[chunk_id] = cused;
    }
}

bool knn_wide16b_supported(int Dt, int Dpad) { return Dpad > 256 && Dpad <= 512 && Dpad - Dt >= 3; }

void launch_knn_wide16b(int mode, int terms, int grid_cus, const void *A16, const void *B16, int Dpad, const float *thr32, int64_t T32,
                        int64_t n_tiles, float *gmin32, int64_t G, void *pool, unsigned int *pool_ctl, int *chunk_fill,
                        int max_chunks, int pool_chunk, hipStream_t s)
{
    const int nQT = (int)(T32 / 32), KB = Dpad / 16;
    const int64_t n_items = ((n_tiles + 3) / 4) * ((nQT + 3) / 4);
    int64_t blocks = 2 * (int64_t)grid_cus;
    if (blocks > n_items) blocks = n_items;
#define SNK_W16(MODE_, TERMS_)                                                                                        \
    hipLaunchKernelGGL((knn_wide16b<MODE_, TERMS_>), dim3((unsigned)blocks), dim3(256), 0, s, (const u32x4 *)A16,     \
                       (const u32x4 *)B16, KB, thr32, nQT, n_tiles, gmin32, G, (PoolEntry16 *)pool, pool_ctl, chunk_fill, \
                       max_chunks, pool_chunk)
    if (mode == 0) { if (terms == 4) SNK_W16(0, 4); else SNK_W16(0, 3); }
    else { if (terms == 4) SNK_W16(1, 4); else SNK_W16(1, 3); }
#undef SNK_W16
}

#define THR16_GROUPS 2048
// ---------------------------------------------------------------------------
// threshold from the f32 group minima: K-th smallest of G values + eps_t, as f32 rounded UP.
// bound = (K-th smallest group minimum) + eps is a true upper bound of the K-th nearest key of THIS
// database; thr = bound + eps so that the filter's approximate test keeps everything below it.
// Row-sharded databases: the smallest bound over all shards still bounds the K-th nearest key of
// the WHOLE database, so the ranks exchange their bounds (bound_out -> all-reduce MIN -> bound_in)
// and every shard filters against that instead of its own, looser one: G times fewer survivors.
// ---------------------------------------------------------------------------
// One WAVEFRONT per row (four rows per workgroup), no LDS, no barriers: the (folded) group minima of the row sit in the lanes'
// registers (P / 64 each) as order-preserving integer images and the K-th smallest is found bit by bit from the top -- 32 rounds of
// "how many values agree with the prefix so far and have a 0 here" (compare + count per value, one wavefront-wide sum per round).
// (Until round 5: a workgroup per row, four passes of a 256-bin LDS histogram with atomics and ten barriers -- 0.21 ms per 9 600
// rows inside a B* step, 5 % of its kernel time, for 39 MB of input.)
__global__ void __launch_bounds__(256)
knn_threshold16_kernel(const float *__restrict__ gmin32, int64_t G, int64_t T, int64_t T32, int K,
                       const double *__restrict__ eps, double *__restrict__ thr, float *__restrict__ thr32,
                       const double *__restrict__ bound_in, double *__restrict__ bound_out,
                       const double *__restrict__ e1, float *__restrict__ thr1, const double *__restrict__ bound2)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T32) return;
    if (row >= T) { if (lane == 0) { thr32[row] = -FLT_MAX; thr[row] = -DBL_MAX; if (thr1) thr1[row] = -FLT_MAX; } return; }
    double bound = DBL_MAX;
    if (bound_in) {
        bound = bound_in[row];
    } else {
        // more than THR16_GROUPS groups are folded (minimum over every P-th group): the K-th smallest minimum of ANY partition
        // of the sample into groups bounds the K-th nearest key from above
        int P = 2;
        while (P < G && P < THR16_GROUPS) P <<= 1;
        constexpr int NVMAX = THR16_GROUPS / 64;
        const int nv = P >= 64 ? P / 64 : 1;
        auto image = [](float f) -> unsigned int {
            const unsigned int b = __float_as_uint(f);
            return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        };
        unsigned int v[NVMAX];
#pragma unroll
        for (int j = 0; j < NVMAX; ++j) {
            v[j] = 0xffffffffu;                                  // beyond the row's values: above everything, never counted
            if (j < nv) {
                const int i = lane + 64 * j;
                float m = FLT_MAX;
                if (i < P) for (int64_t g = i; g < G; g += P) m = fminf(m, gmin32[row * G + g]);
                v[j] = (i < P) ? image(m) : 0xffffffffu;
            }
        }
        if (P >= K && G >= K) {
            unsigned int prefix = 0u, rank = (unsigned int)(K - 1);
            int alive = 64 * nv;                                  // values that agree with the prefix so far (padding included)
            int b = 31;
#pragma unroll 1
            for (; b >= 0 && alive > 1; --b) {
                const unsigned int hi_mask = b == 31 ? 0u : (0xffffffffu << (b + 1));
                int c0 = 0;
#pragma unroll
                for (int j = 0; j < NVMAX; ++j)
                    if (j < nv) c0 += ((v[j] & hi_mask) == prefix && !((v[j] >> b) & 1u)) ? 1 : 0;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) c0 += __shfl_xor(c0, off, 64);
                if (rank >= (unsigned int)c0) { rank -= (unsigned int)c0; prefix |= 1u << b; alive -= c0; }
                else alive = c0;
            }
            if (b >= 0) {
                // one value left under the prefix (rank is 0 then): it is the K-th smallest -- its low bits need no more rounds
                // (1 024 minima part after some 18 of the 32 bits; rows with equal minima at the K-th place run all rounds)
                const unsigned int hi_mask = 0xffffffffu << (b + 1);
                unsigned int found = 0u;
#pragma unroll
                for (int j = 0; j < NVMAX; ++j)
                    if (j < nv && (v[j] & hi_mask) == prefix) found = v[j];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) found |= __shfl_xor(found, off, 64);
                prefix = found;
            }
            const unsigned int bits = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;
            const float kth = __uint_as_float(bits);
            // group minima are approximate: + eps makes the K-th smallest a true upper bound
            if (kth < FLT_MAX) bound = (double)kth + eps[row];
        }
    }
    // a second upper bound of the K-th nearest key (the tiles' balls, stage A'): the smaller one serves
    if (bound2 && bound2[row] < bound) bound = bound2[row];
    if (lane == 0) {
        double vv = DBL_MAX;
        float v32 = FLT_MAX;
        if (bound < 0.5 * DBL_MAX) {
            vv = bound + eps[row];
            if (vv < (double)FLT_MAX) {
                v32 = (float)vv;
                if ((double)v32 < vv) v32 = nextafterf(v32, FLT_MAX);
            }
        }
        thr[row] = vv;
        thr32[row] = v32;
        if (thr1) {
            // the coarse pass's threshold: whatever the three-term key would pass (key3 <= thr32) passes here
            float t1 = FLT_MAX;
            if (v32 < FLT_MAX) {
                const double w = (double)v32 + e1[row];
                if (w < (double)FLT_MAX) { t1 = (float)w; if ((double)t1 < w) t1 = nextafterf(t1, FLT_MAX); }
            }
            thr1[row] = t1;
        }
        if (bound_out) bound_out[row] = bound;
    }
}

void launch_knn_threshold16(const float *gmin32, int64_t G, int64_t T, int64_t T32, int K, const double *eps,
                            double *thr, float *thr32, const double *bound_in, double *bound_out, hipStream_t s,
                            const double *e1, float *thr1, const double *bound2)
{
    hipLaunchKernelGGL(knn_threshold16_kernel, dim3((unsigned)((T32 + 3) / 4)), dim3(256), 0, s,
                       gmin32, G, T, T32, K, eps, thr, thr32, bound_in, bound_out, e1, thr1, bound2);
}

// ---------------------------------------------------------------------------
// self test of the f32 32x32x2 MFMA operand / result maps: C(32x32) = A(32x2) * B(2x32)
// ---------------------------------------------------------------------------
__global__ void mfma16_selftest_kernel(const float *A, const float *B, float *C)
{
    const int lane = threadIdx.x;
    const float a = A[(lane & 31) * 2 + (lane >> 5)];      // A[row = l&31][k = l>>5]
    const float b = B[(lane >> 5) * 32 + (lane & 31)];     // B[k = l>>5][col = l&31]
    f16acc acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[crow32(lane, r) * 32 + (lane & 31)] = acc[r];
}

// one v_mfma_f32_32x32x16_bf16 on caller-chosen bit patterns: D = A (32 x 16) B (16 x 32) + C, row-major arrays;
// what the accumulation term of the bf16-split bound (c_acc) is checked against (tests/test_gpu_prefilter.py)
__global__ void mfma_bf16_probe_kernel(const unsigned short *A, const unsigned short *B, const float *C, float *D)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    u32x4 a, b;
    for (int j = 0; j < 4; ++j) {
        a[j] = (unsigned)A[r * 16 + 8 * h + 2 * j] | ((unsigned)A[r * 16 + 8 * h + 2 * j + 1] << 16);      // A[row r][k = 8 h + ..]
        b[j] = (unsigned)B[(8 * h + 2 * j) * 32 + r] | ((unsigned)B[(8 * h + 2 * j + 1) * 32 + r] << 16);  // B[k][col r]
    }
    f16acc acc;
    for (int i = 0; i < 16; ++i) acc[i] = C[crow32(lane, i) * 32 + r];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[crow32(lane, i) * 32 + r] = acc[i];
}

void launch_mfma_bf16_probe(const unsigned short *A, const unsigned short *B, const float *C, float *D, hipStream_t s)
{
    hipLaunchKernelGGL(mfma_bf16_probe_kernel, dim3(1), dim3(64), 0, s, A, B, C, D);
}

void launch_mfma16_selftest(const float *A, const float *B, float *C, hipStream_t s)
{
    hipLaunchKernelGGL(mfma16_selftest_kernel, dim3(1), dim3(64), 0, s, A, B, C);
}

}  // namespace snk
