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

int64_t greedy_hoist_rows(int64_t nsteps) { return (nsteps + GH_SB - 1) / GH_SB * GH_SB; }
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

}  // namespace snk
