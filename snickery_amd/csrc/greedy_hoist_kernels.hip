// Greedy joint search: the target term of ALL steps of an utterance as one matrix product.
//
// greedy_joint_search (script/synth_simple.py:458-503) asks, at step s, for the window i that minimises
//     ||prev - S'[i]||^2 + ||Q[s] - Fwin[i]||^2 .
// The second term does not depend on the path: W[s][i] = ||Q[s] - Fwin[i]||^2 for every step and window is
//     W[s][i] = sum_k ||w f_(i+ep_k)||^2  -  2 sum_k < f_(i+ep_k) , w q_(s,k) >  +  ||Q[s]||^2 ,
// an (nsteps x nep Dt) by (nep Dt x Nwin) product whose right operand is the database itself: a window's
// row is nep consecutive (or first + last) rows of the unweighted float32 matrix, so the overlapping rows the
// reference materialises (segment_axis, synth_simple.py:201-214: 4.1 GB at 1.5 M units) are LDS reads at a
// row offset here.  SURVEY 8(d) prices the greedy step on exactly this: (Dj + 1) 4 N bytes per step -- the join
// columns and ONE target value per window.
//
// The product runs on the FLOAT64 matrix pipe (v_mfma_f64_16x16x4_f64).  Its values only PREFILTER: the scan
// (greedy32_kernels.hip, HOIST instance) adds them, rounded to float32, to its float32 join totals, and every
// window that could still be the float64 minimum is re-evaluated in the canonical float64 order, as before.  What
// the scan needs is a bound on a stored value W~ against the canonical float64 target term W:
//     |W~ - W| <= u W + hoist_c (||Q[s]|| + ||Fwin[i]||)^2 ,   u = 2^-24 (the float32 store),
//     hoist_c = 4 (n + 8) 2^-53,  n = padded product length
//   - left operand fl(w q) (2^-53 relative), right operand exact (the database IS float32), an n-term float64
//     multiply-add chain in any order: <= (n + 1) 2^-53 ||q|| ||w f||   (Cauchy-Schwarz); the norms in float64;
//   - the canonical value itself is a float64 evaluation (n + 2 roundings of its own): the same size again.
//   The float32 matrix pipe was tried first (v_mfma_f32_32x32x2_f32, proven bound 2 (n + 4) 2^-24 of the squared
//   norms = 0.01 absolute at B3's weights): 50 - 160 windows per step fell inside the bound and had to be decided
//   exactly (330 us per step at 65 536 units against 28).  The expansion ||f||^2 - 2 f.q + ||q||^2 cancels; in
//   float64 that costs nothing, and 100 steps x 1.5 M windows x 384 columns are 1.5 ms of the float64 pipe.
//
// Layout.  Workgroup = 128 windows (4 wavefronts x 2 tiles of 16): their 128 + me - 1 unit rows are staged in LDS
// once as float32 (pitch 64 nch + 4 floats) and every wavefront walks all steps in blocks of 16.  One MFMA consumes
// k = 4: lane quarter h takes columns [16 h, 16 h + 16) of a 64-column chunk, so both operands are 16-byte reads
// of consecutive elements (the order of the k terms is free: both sides use the same one).  The result tile's
// register-to-row map is not assumed: two probe MFMAs (row index / column index as operands) report it.
#include "greedy_common.h"
#include <stdlib.h>

namespace snk {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define GH_WIN 128
#define GH_SB 16            // steps per result tile

struct GhEp { int nep; int ep[GR_MAX_EP]; };

// nw[i] = sum_k ||w f_(i+ep_k)||^2 (float64) and the largest of them
__global__ void __launch_bounds__(256)
hoist_window_norms_kernel(const double *__restrict__ fnorm, int64_t Nwin, int64_t Wp, GhEp e, double *__restrict__ nw,
                          unsigned long long *__restrict__ max_bits)
{
    __shared__ double red[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (i < Nwin) {
        for (int k = 0; k < e.nep; ++k) v += fnorm[i + e.ep[k]];
        nw[i] = v;
    } else if (i < Wp) nw[i] = 0.0;
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    // non-negative doubles order like their bit patterns
    if (threadIdx.x == 0) atomicMax(max_bits, (unsigned long long)__double_as_longlong(red[0]));
}

// Left operand of one utterance: Aq[s][(k nch + ch) 64 + c] = fl(w_c q_(s,k,c)) (zero beyond Dt and beyond the last
// step: the rows are padded to a multiple of GH_SB), qn2[s] = ||Q[s]||^2
__global__ void __launch_bounds__(256)
hoist_prepare_kernel(const double *__restrict__ Q, int64_t q_off, int64_t nsteps, int me, int Dt, int nch, GhEp e,
                     const double *__restrict__ wt, double *__restrict__ Aq, double *__restrict__ qn2)
{
    __shared__ double red[256];
    const int64_t s = blockIdx.x;
    const int KA = e.nep * nch * 64;
    double acc = 0.0;
    for (int j = threadIdx.x; j < KA; j += 256) {
        const int k = j / (nch * 64), c = j - k * (nch * 64);
        double v = 0.0;
        if (s < nsteps && c < Dt) {
            const double q = Q[(q_off + s * me + e.ep[k]) * Dt + c];
            acc += q * q;
            v = wt[c] * q;
        }
        Aq[s * KA + j] = v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) qn2[s] = red[0];                       // qn2 has the padded number of rows
}

// W[s][i] for all steps of one utterance and the workgroup's 128 windows
__global__ void __launch_bounds__(256)
hoist_product_kernel(const float *__restrict__ F_unw, int Fp, int64_t n_f_rows, GhEp e, int nch, const double *__restrict__ Aq,
                     int64_t nsteps, const double *__restrict__ qn2, const double *__restrict__ nw, float *__restrict__ W,
                     int64_t Wp)
{
    extern __shared__ __align__(16) float Fs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 4, c16 = lane & 15;
    const int P = nch * 64 + 4, q4n = nch * 16;
    const int64_t w0 = (int64_t)blockIdx.x * GH_WIN;
    const int nrows = GH_WIN + e.ep[e.nep - 1];
    for (int idx = tid; idx < nrows * q4n; idx += 256) {
        const int r = idx / q4n, col = (idx - r * q4n) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (w0 + r < n_f_rows && col < Fp) v = *reinterpret_cast<const f32x4 *>(F_unw + (w0 + r) * Fp + col);
        *reinterpret_cast<f32x4 *>(Fs + (size_t)r * P + col) = v;
    }
    __syncthreads();
    // where the result registers live: D = A B with A[i][k] = (k == 0) i, B[k][j] = (k == 0)  ->  D[i][j] = i, and
    // with A[i][k] = (k == 0), B[k][j] = (k == 0) j  ->  D[i][j] = j
    int rrow[4], rcol[4];
    {
        const f64x4 z = {0.0, 0.0, 0.0, 0.0};
        const f64x4 ri = __builtin_amdgcn_mfma_f64_16x16x4f64(h == 0 ? (double)c16 : 0.0, h == 0 ? 1.0 : 0.0, z, 0, 0, 0);
        const f64x4 ci = __builtin_amdgcn_mfma_f64_16x16x4f64(h == 0 ? 1.0 : 0.0, h == 0 ? (double)c16 : 0.0, z, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { rrow[i] = (int)ri[i]; rcol[i] = (int)ci[i]; }
    }
    const int KA = e.nep * nch * 64;
    const int64_t nsb = (nsteps + GH_SB - 1) / GH_SB;
    const int64_t wbase = w0 + wave * 32;
    for (int64_t sb = 0; sb < nsb; ++sb) {
        f64x4 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
        const double *arow = Aq + (sb * GH_SB + c16) * KA + 16 * h;
        for (int k = 0; k < e.nep; ++k) {
            const float *brow = Fs + (size_t)(wave * 32 + c16 + e.ep[k]) * P + 16 * h;
            for (int ch = 0; ch < nch; ++ch) {
                double a[16];
                const double *ap = arow + (k * nch + ch) * 64;
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const f64x2 v = *reinterpret_cast<const f64x2 *>(ap + j);
                    a[j] = v[0]; a[j + 1] = v[1];
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 b[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4 *>(brow + (size_t)t * 16 * P + ch * 64 + 4 * j);
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], (double)b[j >> 2][j & 3], acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t s = sb * GH_SB + rrow[i], wcol = wbase + t * 16 + rcol[i];
                if (s < nsteps) {
                    const double v = (qn2[s] + nw[wcol]) - 2.0 * acc[t][i];
                    __builtin_nontemporal_store((float)(v > 0.0 ? v : 0.0), W + s * Wp + wcol);      // W >= 0: clamping only helps
                }
            }
    }
}

// ---- the same product on the bf16 matrix pipe, for scans of float16 join tiles (greedy32_kernels.hip, F16 instances) ----
// With float16 tiles the scan's own bound is 2 sqrt(d) D + D^2, D = 2^-11 max ||w o S'[i]|| -- 4e-3 absolute at B3's weights,
// fifteen windows per step -- so the target values need not be better than ~1e-4, and the float64 pipe (2.4 ms per 100 steps at
// 1.5 M units, a sixth of the step) is the wrong tool.  Both operands (the query side after rounding fl(w q) to float32) are split
// EXACTLY into three bf16 pieces, x = h + m + l (h = bf16(x), m = bf16(x - h), l = x - h - m: 8 + 8 + 8 bits), and a k-block
// of 16 columns is six v_mfma_f32_32x32x16_bf16: h.h into a FRESH accumulator g1, added to the epoch's partial sum in float32
// on the vector unit; h.m + m.h + m.m + h.l + l.h into an accumulator g2 of the epoch (dropped: m.l + l.m + l.l <= 1.02 2^-25
// of |f a|); fl32(partial + g2) of the epochs are added in float32.  Fresh accumulators matter: an MFMA is off by at most 2^-20
// of (|products| + |C|) (the probed property of the instruction, snk_probe_mfma_bf16), so a chain through one accumulator pays
// 2^-20 of the running sum at every link, a fresh one 2^-20 of its own block.  (The first version -- two pieces, one accumulator
// per block, float32 chain over all blocks: cG = 1.6e-5 -- doubled the scan's window: 36 windows per step instead of 15, and gave
// back in exact decisions what the product saved; float64 additions per block were 3/4 of the second version's time, float64
// sums per epoch cost the registers of the second wavefront per SIMD.)  Bound, with P = sum_c |f_c a_c| <= ||w f|| ||q||
// (elementwise: f_c (w_c q_c) = (w_c f_c) q_c, then Cauchy-Schwarz), kpe k-blocks per epoch, nep epochs:
//     |G~ - G| <= cG P,  cG = 1.02 [ 1.01 2^-20 (g1) + 5 kpe 2^-20 0.006 (g2's MFMAs: C is 2^-8, the products 2^-9 of the epoch)
//                                   + (kpe + nep + 2) 2^-24 (the float32 chains: blocks of an epoch, partial + g2, epochs)
//                                   + 1.02 2^-25 (dropped) ] + 2^-24 (fl32 of the query side) = 1.9e-6 at 61 columns, 6 epochs,
//     |W~ - W| <= 2 cG ||w f|| ||q|| + the float64 terms <= hoist_c16 (||Q[s]|| + ||Fwin[i]||)^2,  hoist_c16 = 0.505 cG + hoist_c:
// 1.9e-4 absolute at B3's weights, 5 % of the scan's window.
// Left operand: fragment order, A16[(sb n_kb + kb) 64 + lane] = (8 h | 8 m | 8 l pieces: 48 bytes) of step 32 sb + (lane & 31),
// columns 16 kb + 8 (lane >> 5) + 0..7 (coalesced 3 KB reads, from L2); right operand: the workgroup's window rows in LDS as
// pieces, row pitch 384 nch + 16 bytes (conflict-free 16-byte reads).  Four wavefronts per workgroup, 32 windows each, all step
// blocks of 32 one after the other: 53 KB of LDS and 152 registers leave room for three workgroups per compute unit, which is what
// hides the staging of the rows (64 windows per wavefront and two step phases per workgroup -- half the left-operand traffic, one
// workgroup per compute unit -- were 8 % slower).  0.83 ms per 100 steps at 1.5 M units against 2.36 ms on the float64 pipe:
// 1.07 PFLOP/s of bf16 MFMAs = 0.43 of the dense peak, beside 3.7 vector instructions per MFMA.
typedef __bf16 gh_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gh_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gh_f32x2 __attribute__((ext_vector_type(2)));
typedef float gh_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int gh_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gh_u32x2 __attribute__((ext_vector_type(2)));

// two float32 -> their three bf16 pieces each (packed pairs)
__device__ __forceinline__ void gh_split3(gh_f32x2 y, unsigned int &h, unsigned int &m, unsigned int &l)
{
    h = __builtin_bit_cast(unsigned int, __builtin_convertvector(y, gh_bf16x2));
    const gh_f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
    const gh_f32x2 r1 = y - hf;                                 // exact
    m = __builtin_bit_cast(unsigned int, __builtin_convertvector(r1, gh_bf16x2));
    const gh_f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
    l = __builtin_bit_cast(unsigned int, __builtin_convertvector(r1 - mf, gh_bf16x2));      // exact, and 8 bits at most
}

// grid: steps padded to 32; A16 in fragment order, qn2[s] = ||Q[s]||^2 (float64; rows beyond the last step: 0)
__global__ void __launch_bounds__(256)
hoist_prepare16_kernel(const double *__restrict__ Q, int64_t q_off, int64_t nsteps, int me, int Dt, int nch, GhEp e,
                       const double *__restrict__ wt, unsigned short *__restrict__ A16, double *__restrict__ qn2)
{
    __shared__ double red[256];
    const int64_t s = blockIdx.x;
    const int KA = e.nep * nch * 64, n_kb = KA / 16;
    const int64_t sb = s >> 5;
    const int r = (int)(s & 31);
    double acc = 0.0;
    for (int j = threadIdx.x; j < KA; j += 256) {
        const int k = j / (nch * 64), c = j - k * (nch * 64);
        float v = 0.f;
        if (s < nsteps && c < Dt) {
            const double q = Q[(q_off + s * me + e.ep[k]) * Dt + c];
            acc += q * q;
            v = (float)(wt[c] * q);
        }
        unsigned int h, m, l;
        gh_split3(gh_f32x2{v, 0.f}, h, m, l);
        const int kb = j >> 4, kk = j & 15, lane = r + 32 * (kk >> 3), el = kk & 7;
        unsigned short *dst = A16 + (((size_t)sb * n_kb + kb) * 64 + lane) * 24;
        dst[el] = (unsigned short)h;
        dst[8 + el] = (unsigned short)m;
        dst[16 + el] = (unsigned short)l;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) qn2[s] = red[0];
}

__global__ void __launch_bounds__(256)
hoist_product16_kernel(const float *__restrict__ F_unw, int Fp, int64_t n_f_rows, GhEp e, int nch, const gh_u32x4 *__restrict__ A16,
                       int64_t nsteps, const double *__restrict__ qn2, const double *__restrict__ nw, float *__restrict__ W,
                       int64_t Wp)
{
    extern __shared__ __align__(16) char Bs[];
    constexpr int T = 1, WGW = 128 * T;                     // windows per workgroup: four wavefronts x 32 T
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wg = wave, nthr = 256;
    const int pitch = nch * 384 + 16, pc = nch * 128, q4n = nch * 16;
    const int64_t w0 = (int64_t)blockIdx.x * WGW;
    const int nrows = WGW + e.ep[e.nep - 1];
    // (four requests in flight per thread: one workgroup per compute unit, nobody else hides the round trips to HBM)
    for (int base = tid; base < nrows * q4n; base += nthr * 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * nthr, r = idx / q4n, col = (idx - r * q4n) * 4;
            const bool ok = idx < nrows * q4n && w0 + r < n_f_rows && col < Fp;
            const f32x4 x = *reinterpret_cast<const f32x4 *>(F_unw + (ok ? (w0 + r) * Fp + col : 0));
            v[u] = ok ? x : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * nthr, r = idx / q4n, col = (idx - r * q4n) * 4;
            if (idx >= nrows * q4n) break;
            gh_u32x2 ph, pm, pl;
            unsigned int a, b, c;
            gh_split3(gh_f32x2{v[u][0], v[u][1]}, a, b, c); ph[0] = a; pm[0] = b; pl[0] = c;
            gh_split3(gh_f32x2{v[u][2], v[u][3]}, a, b, c); ph[1] = a; pm[1] = b; pl[1] = c;
            char *dst = Bs + (size_t)r * pitch + col * 2;
            *reinterpret_cast<gh_u32x2 *>(dst) = ph;
            *reinterpret_cast<gh_u32x2 *>(dst + pc) = pm;
            *reinterpret_cast<gh_u32x2 *>(dst + 2 * pc) = pl;
        }
    }
    __syncthreads();
    const int n_kb = e.nep * nch * 4, kpe = nch * 4;          // k-blocks, and k-blocks per epoch
    const int64_t wbase = w0 + wg * (32 * T);
    const gh_f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mfma = [](gh_u32x4 a, gh_u32x4 b, gh_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gh_bf16x8, a), __builtin_bit_cast(gh_bf16x8, b), c, 0, 0, 0);
    };
    const int64_t nsb = (nsteps + 31) / 32;
    const char *const bbase = Bs + (size_t)(wg * 32 * T + l31) * pitch + 16 * half;
    for (int64_t sb = 0; sb < nsb; ++sb) {
        gh_f32x16 sum[T];
#pragma unroll
        for (int t = 0; t < T; ++t) sum[t] = zero;
        const gh_u32x4 *ap = A16 + ((size_t)sb * n_kb * 64 + lane) * 3;
        // software pipeline, two register sets: the left operands (L2) are requested two k-blocks ahead, the right ones
        // (LDS) one k-block ahead -- a wavefront has one partner on its SIMD at most, nothing else hides the latencies
        gh_u32x4 A[2][3], B[2][T][3];
        auto a_load = [&](int kb, gh_u32x4 (&a)[3]) {
            const gh_u32x4 *p = ap + (size_t)(kb < n_kb ? kb : n_kb - 1) * 192;
            a[0] = p[0]; a[1] = p[1]; a[2] = p[2];
        };
        int nk = 0, ncb = 0;                                   // epoch and block in it of the NEXT right operands
        auto b_load = [&](gh_u32x4 (&b)[T][3]) {
            const char *q = bbase + (size_t)e.ep[nk] * pitch + ncb * 32;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                b[t][0] = *reinterpret_cast<const gh_u32x4 *>(q + (size_t)t * 32 * pitch);
                b[t][1] = *reinterpret_cast<const gh_u32x4 *>(q + (size_t)t * 32 * pitch + pc);
                b[t][2] = *reinterpret_cast<const gh_u32x4 *>(q + (size_t)t * 32 * pitch + 2 * pc);
            }
            if (++ncb == kpe) { ncb = 0; if (nk + 1 < e.nep) ++nk; }
        };
        a_load(0, A[0]); a_load(1, A[1]); b_load(B[0]);
        gh_f32x16 part[T], g2[T];                              // the epoch's h.h blocks (float32 chain) and its small products
#pragma unroll
        for (int t = 0; t < T; ++t) { part[t] = zero; g2[t] = zero; }
        int cb = 0;
        for (int kb = 0; kb < n_kb; kb += 2) {                  // (kpe is even: an epoch ends behind an odd block)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b_load(B[j ^ 1]);
                const gh_u32x4 ch = A[j][0], cm = A[j][1], cl = A[j][2];
                a_load(kb + j + 2, A[j]);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const gh_f32x16 g1 = mfma(ch, B[j][t][0], zero);
                    g2[t] = mfma(cl, B[j][t][0], g2[t]);
                    g2[t] = mfma(ch, B[j][t][2], g2[t]);
                    g2[t] = mfma(cm, B[j][t][1], g2[t]);
                    g2[t] = mfma(cm, B[j][t][0], g2[t]);
                    g2[t] = mfma(ch, B[j][t][1], g2[t]);
                    part[t] += g1;
                }
            }
            cb += 2;
            if (cb == kpe) {
                cb = 0;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    sum[t] += part[t] + g2[t];
                    part[t] = zero; g2[t] = zero;
                }
            }
        }
        // epilogue, branch-free: the 16 norms of the lane's steps are requested together (qn2 has the padded rows); stores
        // through a buffer descriptor of the step block's rows -- rows beyond the last step and windows beyond the pitch
        // fall outside it (a conditional store per element was a chain of 32 round trips: 2/3 of the kernel's time)
        const int64_t s0 = sb * 32;
        const int64_t rows_here = nsteps - s0 < 32 ? nsteps - s0 : 32;
        const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(W + s0 * Wp, 0, (int)(rows_here * Wp * 4), 0x00020000);
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += 8) {
            double q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = qn2[s0 + ((i0 + i) & 3) + 8 * ((i0 + i) >> 2) + 4 * half];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int64_t wcol = wbase + t * 32 + l31;
                const bool ok = wcol < Wp;
                const double nwv = ok ? nw[wcol] : 0.0;
                const unsigned int vbase = ok ? (unsigned int)((4 * half * Wp + wcol) * 4) : 0x7ffffffcu;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double v = (q[i] + nwv) - 2.0 * (double)sum[t][i0 + i];
                    const float o = (float)(v > 0.0 ? v : 0.0);
                    const unsigned int voff = ok ? vbase + (unsigned int)((((i0 + i) & 3) + 8 * ((i0 + i) >> 2)) * Wp * 4) : 0x7ffffffcu;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, o), wres, (int)voff, 0, 2);
                }
            }
        }
    }
}

int64_t greedy_hoist_rows(int64_t nsteps) { return (nsteps + 31) / 32 * 32; }      // (the bf16 product's step blocks; two of the float64 one's)
int64_t greedy_hoist_pitch(const GreedyLayout &g) { return (g.Nwin + GH_WIN - 1) / GH_WIN * GH_WIN; }
int greedy_hoist_k(const GreedyLayout &g, int Dt)
{
    const int nep = (g.last_frame_as_target && g.me > 1) ? 2 : g.me;
    return nep * ((Dt + 63) / 64) * 64;
}
// the hoisted form needs three join chunks per tile at least (the scan requests three chunks ahead and keeps the W values
// of two tiles: the one in work and the next)
// and the staged rows in LDS
bool greedy_hoist_supported(const GreedyLayout &g, int Dt)
{
    const int nch = (Dt + 63) / 64;
    return (g.jdim + GR_CC - 1) / GR_CC >= 3 && (size_t)(GH_WIN + g.me - 1) * (nch * 64 + 4) * 4 <= (size_t)(160 * 1024);
}
double greedy_hoist_c(const GreedyLayout &g, int Dt) { return 4.0 * (double)(greedy_hoist_k(g, Dt) + 8) * 1.1102230246251565e-16; }
// the bf16 product's piece rows of 128 windows must fit LDS
static bool gh16_fits(const GreedyLayout &g, int Dt)
{
    return (size_t)(128 + g.me - 1) * ((Dt + 63) / 64 * 384 + 16) <= (size_t)(158 * 1024);
}
bool greedy_hoist16_supported(const GreedyLayout &g, int Dt)
{
    return greedy_hoist_supported(g, Dt) && gh16_fits(g, Dt) && g.Nwin <= ((int64_t)1 << 24);      // (a step block's rows: one 2 GB descriptor)
}
double greedy_hoist_c16(const GreedyLayout &g, int Dt)
{
    const double kpe = (double)((Dt + 63) / 64 * 4), nep = (g.last_frame_as_target && g.me > 1) ? 2.0 : (double)g.me;
    const double u20 = 9.5367431640625e-07, u24 = 5.9604644775390625e-08;
    const double cg = 1.02 * (1.01 * u20 + 5.0 * kpe * u20 * 0.006 + (kpe + nep + 2.0) * u24 + 1.02 * 0.5 * u24) + u24;
    return 0.505 * cg + greedy_hoist_c(g, Dt);
}

static GhEp gh_epochs(const GreedyLayout &g)
{
    GhEp e{};
    if (g.last_frame_as_target && g.me > 1) { e.nep = 2; e.ep[0] = 0; e.ep[1] = g.me - 1; }
    else { e.nep = g.me; for (int k = 0; k < g.me; ++k) e.ep[k] = k; }
    return e;
}

// once per (database, layout, weights): nw (Wp floats) and *max_bits = bits of max_i ||Fwin[i]||^2 (float64)
void launch_hoist_window_norms(const GreedyLayout &g, const double *fnorm, double *nw, unsigned long long *max_bits, hipStream_t s)
{
    const int64_t Wp = greedy_hoist_pitch(g);
    (void)hipMemsetAsync(max_bits, 0, sizeof(unsigned long long), s);
    hipLaunchKernelGGL(hoist_window_norms_kernel, dim3((unsigned)((Wp + 255) / 256)), dim3(256), 0, s, fnorm, g.Nwin, Wp, gh_epochs(g), nw, max_bits);
}

// one utterance: Aq ((nsteps rounded up to 16) x K doubles), qn2 (the same rows), W (nsteps x Wp floats)
void launch_hoist_product(const GreedyLayout &g, const float *F_unw, int Fp, int64_t n_f_rows, int Dt, const double *wt, const double *Q,
                          int64_t q_off, int64_t nsteps, const double *nw, double *Aq, double *qn2, float *W, hipStream_t s)
{
    if (nsteps <= 0) return;
    const GhEp e = gh_epochs(g);
    const int nch = (Dt + 63) / 64;
    const int64_t rows = greedy_hoist_rows(nsteps), Wp = greedy_hoist_pitch(g);
    hipLaunchKernelGGL(hoist_prepare_kernel, dim3((unsigned)rows), dim3(256), 0, s, Q, q_off, nsteps, g.me, Dt, nch, e, wt, Aq, qn2);
    const size_t lds = (size_t)(GH_WIN + g.me - 1) * (nch * 64 + 4) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hoist_product_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(hoist_product_kernel, dim3((unsigned)(Wp / GH_WIN)), dim3(256), lds, s, F_unw, Fp, n_f_rows, e, nch, Aq, nsteps,
                       qn2, nw, W, Wp);
}

// the same on the bf16 pipe (scans of float16 join tiles): Aq holds the fragment-ordered pieces (48 bytes per lane, step
// block of 32 and k-block: never more than the float64 operand's bytes)
void launch_hoist_product16(const GreedyLayout &g, const float *F_unw, int Fp, int64_t n_f_rows, int Dt, const double *wt, const double *Q,
                            int64_t q_off, int64_t nsteps, const double *nw, double *Aq, double *qn2, float *W, hipStream_t s)
{
    if (nsteps <= 0) return;
    const GhEp e = gh_epochs(g);
    const int nch = (Dt + 63) / 64;
    const int64_t rows = (nsteps + 31) / 32 * 32, Wp = greedy_hoist_pitch(g);
    hipLaunchKernelGGL(hoist_prepare16_kernel, dim3((unsigned)rows), dim3(256), 0, s, Q, q_off, nsteps, g.me, Dt, nch, e, wt,
                       reinterpret_cast<unsigned short *>(Aq), qn2);
    const size_t lds = (size_t)(128 + g.me - 1) * (nch * 384 + 16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hoist_product16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(hoist_product16_kernel, dim3((unsigned)((g.Nwin + 127) / 128)), dim3(256), lds, s, F_unw, Fp, n_f_rows, e, nch,
                       reinterpret_cast<const gh_u32x4 *>(Aq), nsteps, qn2, nw, W, Wp);
}

}  // namespace snk
