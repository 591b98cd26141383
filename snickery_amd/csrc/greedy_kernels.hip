// Greedy joint search for gfx950: one exact nearest-neighbour scan of the windowed unit
// database per step.
//
// Replaces the per-step `self.joint_tree.query(both, k=1, eps=...)` of
// greedy_joint_search (script/synth_simple.py:458-503 == script/synth_halfphone.py:1900-1945)
// and the data layout of get_tree_for_greedy_search (script/synth_simple.py:190-229):
//   combined[i] = [ prev_join_rep[i] , F[i], F[i+1], ..., F[i+me-1] ]      (517-D at me=6)
//   d2(i) = || prev - prev_join_rep[i] ||^2 + || q_s - Fwin[i] ||^2 ,  i* = argmin, lowest id on ties
//
// The windowed database is never materialised (the reference hstacks an (N-me+1) x 517 float64
// copy, 4.1 GB at N = 1 M): window i reads rows i..i+me-1 of the UNWEIGHTED float32 feature
// matrix and row i of the unweighted join matrix exactly as the HDF5 file stores them, and
// applies the float64 stream weights on the fly (fl64(f32 * w) is bit-identical to the
// reference's speech_manip.weight()).  Every step is therefore a pure HBM-bandwidth-bound
// stream over (Dj + Dt) * 4 bytes per unit.  Squared distances are accumulated in the canonical
// oracle order (column by column, separately rounded sub/mul/add), so the argmin is bit-exact.
//
// Per step: greedy_scan_kernel (one workgroup per R consecutive windows; the R rows form ONE
// contiguous span of the row-major matrix, copied to LDS with 16-byte loads; thread t then
// walks row t, odd row pitch => conflict-free) and greedy_pick_kernel (cross-block argmin,
// appends to the path, fetches the winner's `current_join_rep` row as the next `prev`).
#include "snk_internal.h"
#include <float.h>

namespace snk {

#define GR_R 128          // windows per workgroup
#define GR_MAX_EP 16      // max multiepoch

struct GreedyArgs {
    const float *JC_unw; int Dj; const double *wj;
    const float *F_unw; int Dt; const double *wt;
    int me, nep; int ep[GR_MAX_EP];       // epochs of the window that enter the target term
    int prev_col0, cur_col0, jdim;
    int64_t prev_row0, cur_row0, Nwin;
    int R;                                // windows per workgroup (<= GR_R, fits LDS)
    const double *Q;                      // (T, Dt) weighted targets, row-major
    const double *prev_vec;               // (jdim)
};

__global__ void __launch_bounds__(GR_R)
greedy_scan_kernel(GreedyArgs a, int64_t step, double *__restrict__ blk_min,
                   int64_t *__restrict__ blk_arg)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * a.R;
    const int rows = (int)((a.Nwin - i0 < a.R) ? (a.Nwin - i0) : a.R);
    const int frows = rows + a.me - 1;

    float *jt = reinterpret_cast<float *>(smem);                    // [GR_R][Dj]
    float *ft = jt + ((a.R * a.Dj + 3) & ~3);                       // [R+me-1][Dt]
    double *wjs = reinterpret_cast<double *>(ft + (((a.R + a.me - 1) * a.Dt + 3) & ~3));
    double *wts = wjs + a.Dj;
    double *prevs = wts + a.Dt;
    double *qs = prevs + a.jdim;                                     // [nep][Dt]
    __shared__ double red_v[GR_R];
    __shared__ int64_t red_i[GR_R];

    // contiguous spans -> LDS
    {
        const float *src = a.JC_unw + (a.prev_row0 + i0) * a.Dj;
        const int n = rows * a.Dj;
        for (int e = tid; e < n; e += GR_R) jt[e] = src[e];
        const float *fsrc = a.F_unw + i0 * a.Dt;
        const int nf = frows * a.Dt;
        for (int e = tid; e < nf; e += GR_R) ft[e] = fsrc[e];
        for (int e = tid; e < a.Dj; e += GR_R) wjs[e] = a.wj[e];
        for (int e = tid; e < a.Dt; e += GR_R) wts[e] = a.wt[e];
        for (int e = tid; e < a.jdim; e += GR_R) prevs[e] = a.prev_vec[e];
        for (int e = tid; e < a.nep * a.Dt; e += GR_R) {
            const int k = e / a.Dt, c = e % a.Dt;
            qs[e] = a.Q[(step * a.me + a.ep[k]) * a.Dt + c];
        }
    }
    __syncthreads();

    double best = DBL_MAX;
    int64_t arg = INT64_MAX;
    if (tid < rows) {
        double accj = 0.0;
        const float *jr = jt + tid * a.Dj + a.prev_col0;
        const double *wjr = wjs + a.prev_col0;
        for (int c = 0; c < a.jdim; ++c) {
            const double v = __dmul_rn((double)jr[c], wjr[c]);
            const double d = __dsub_rn(v, prevs[c]);
            accj = __dadd_rn(accj, __dmul_rn(d, d));
        }
        double acct = 0.0;
        for (int k = 0; k < a.nep; ++k) {
            const float *fr = ft + (tid + a.ep[k]) * a.Dt;
            const double *q = qs + k * a.Dt;
            for (int c = 0; c < a.Dt; ++c) {
                const double v = __dmul_rn((double)fr[c], wts[c]);
                const double d = __dsub_rn(v, q[c]);
                acct = __dadd_rn(acct, __dmul_rn(d, d));
            }
        }
        best = __dadd_rn(accj, acct);
        arg = i0 + tid;
    }
    red_v[tid] = best;
    red_i[tid] = arg;
    __syncthreads();
    for (int off = GR_R / 2; off > 0; off >>= 1) {
        if (tid < off) {
            const double v2 = red_v[tid + off];
            const int64_t i2 = red_i[tid + off];
            if (v2 < red_v[tid] || (v2 == red_v[tid] && i2 < red_i[tid])) { red_v[tid] = v2; red_i[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) { blk_min[blockIdx.x] = red_v[0]; blk_arg[blockIdx.x] = red_i[0]; }
}

__global__ void __launch_bounds__(256)
greedy_pick_kernel(GreedyArgs a, int64_t step, const double *__restrict__ blk_min,
                   const int64_t *__restrict__ blk_arg, int nblk, double *__restrict__ prev_vec,
                   int64_t *__restrict__ path, double *__restrict__ dist)
{
    __shared__ double red_v[256];
    __shared__ int64_t red_i[256];
    const int tid = threadIdx.x;
    double best = DBL_MAX;
    int64_t arg = INT64_MAX;
    for (int b = tid; b < nblk; b += 256) {
        const double v = blk_min[b];
        const int64_t i = blk_arg[b];
        if (v < best || (v == best && i < arg)) { best = v; arg = i; }
    }
    red_v[tid] = best; red_i[tid] = arg;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            const double v2 = red_v[tid + off];
            const int64_t i2 = red_i[tid + off];
            if (v2 < red_v[tid] || (v2 == red_v[tid] && i2 < red_i[tid])) { red_v[tid] = v2; red_i[tid] = i2; }
        }
        __syncthreads();
    }
    const int64_t ix = red_i[0];
    if (tid == 0) { path[step] = ix; if (dist) dist[step] = __dsqrt_rn(red_v[0]); }
    // prev_join_vector = current_join_rep[ix]   (synth_simple.py:501)
    const float *src = a.JC_unw + (a.cur_row0 + ix) * a.Dj + a.cur_col0;
    for (int c = tid; c < a.jdim; c += 256)
        prev_vec[c] = __dmul_rn((double)src[c], a.wj[a.cur_col0 + c]);
}

__global__ void greedy_init_prev_kernel(GreedyArgs a, int64_t start_state, double *prev_vec)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.jdim) return;
    if (start_state < 0) { prev_vec[c] = 0.0; return; }     // np.zeros((n,)) synth_simple.py:467-468
    const float *src = a.JC_unw + (a.prev_row0 + start_state) * a.Dj + a.prev_col0;
    prev_vec[c] = __dmul_rn((double)src[c], a.wj[a.prev_col0 + c]);
}

static int greedy_rows(const GreedyLayout &g, int Dt, int Dj)
{
    const int nep = (g.last_frame_as_target && g.me > 1) ? 2 : g.me;
    const size_t fixed = (size_t)(Dj + Dt + g.jdim + nep * Dt) * sizeof(double) + (size_t)(g.me + 2) * Dt * 4 + 64;
    const size_t budget = 144 * 1024;
    if (fixed >= budget) return 0;
    size_t r = (budget - fixed) / ((size_t)(Dj + Dt) * sizeof(float));
    if (r > GR_R) r = GR_R;
    if (r >= 64) r = (r / 32) * 32;
    return (int)r;
}

static size_t greedy_shmem_for(const GreedyLayout &g, int Dt, int Dj, int R)
{
    const int nep = (g.last_frame_as_target && g.me > 1) ? 2 : g.me;
    return (size_t)(((R * Dj + 3) & ~3) + (((R + g.me - 1) * Dt + 3) & ~3)) * sizeof(float)
           + (size_t)(Dj + Dt + g.jdim + nep * Dt) * sizeof(double);
}

void launch_greedy(const GreedyLayout &g, const float *F_unw, int Dt, const double *wt,
                   const float *JC_unw, int Dj, const double *wj, const double *Q,
                   int64_t nsteps, int64_t start_state, double *prev_vec, double *blk_min,
                   int64_t *blk_arg, int nblk, int64_t *path, double *dist, hipStream_t s)
{
    GreedyArgs a{};
    a.JC_unw = JC_unw; a.Dj = Dj; a.wj = wj;
    a.F_unw = F_unw; a.Dt = Dt; a.wt = wt;
    a.me = g.me;
    if (g.last_frame_as_target && g.me > 1) { a.nep = 2; a.ep[0] = 0; a.ep[1] = g.me - 1; }
    else { a.nep = g.me; for (int e = 0; e < g.me; ++e) a.ep[e] = e; }
    a.prev_col0 = g.prev_col0; a.cur_col0 = g.cur_col0; a.jdim = g.jdim;
    a.prev_row0 = g.prev_row0; a.cur_row0 = g.cur_row0; a.Nwin = g.Nwin;
    a.Q = Q; a.prev_vec = prev_vec;
    a.R = greedy_rows(g, Dt, Dj);
    const size_t shmem = greedy_shmem_for(g, Dt, Dj, a.R);
    static size_t attr = 0;
    if (shmem > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&greedy_scan_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        attr = shmem;
    }
    hipLaunchKernelGGL(greedy_init_prev_kernel, dim3((g.jdim + 255) / 256), dim3(256), 0, s, a,
                       start_state, prev_vec);
    for (int64_t st = 0; st < nsteps; ++st) {
        hipLaunchKernelGGL(greedy_scan_kernel, dim3(nblk), dim3(GR_R), shmem, s, a, st, blk_min, blk_arg);
        hipLaunchKernelGGL(greedy_pick_kernel, dim3(1), dim3(256), 0, s, a, st, blk_min, blk_arg, nblk,
                           prev_vec, path, dist);
    }
}

size_t greedy_shmem_bytes(const GreedyLayout &g, int Dt, int Dj)
{
    const int R = greedy_rows(g, Dt, Dj);
    if (R < 8) return (size_t)1 << 30;      // does not fit
    return greedy_shmem_for(g, Dt, Dj, R);
}

int greedy_blocks(const GreedyLayout &g, int Dt, int Dj)
{
    const int R = greedy_rows(g, Dt, Dj);
    return (int)((g.Nwin + R - 1) / R);
}

// ---------------------------------------------------------------------------
// per-column squared errors along a path (get_target_scores_per_stream /
// get_join_scores_per_stream, script/synth_halfphone.py:1964-1981)
// ---------------------------------------------------------------------------
__global__ void path_scores_kernel(GreedyArgs a, int mode, const int64_t *__restrict__ path,
                                   int64_t L, int jcols, int64_t e_row_shift,
                                   double *__restrict__ tsq, double *__restrict__ jsq)
{
    const int64_t l = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t p = path[l];
    const int tcols = a.nep * a.Dt;
    for (int e = tid; e < tcols; e += blockDim.x) {
        const int k = e / a.Dt, c = e % a.Dt;
        const double f = __dmul_rn((double)a.F_unw[(p + a.ep[k]) * a.Dt + c], a.wt[c]);
        const double q = a.Q[(l * a.me + a.ep[k]) * a.Dt + c];
        const double d = __dsub_rn(f, q);
        tsq[l * tcols + e] = __dmul_rn(d, d);
    }
    if (l + 1 < L) {
        const int64_t pn = path[l + 1];
        for (int c = tid; c < jcols; c += blockDim.x) {
            double x, y;
            if (mode == 0) {   // (unit_end_data[p[:-1]] - unit_start_data[p[1:]])**2
                x = __dmul_rn((double)a.JC_unw[(p + 1) * a.Dj + c], a.wj[c]);
                y = __dmul_rn((double)a.JC_unw[pn * a.Dj + c], a.wj[c]);
                const double d = __dsub_rn(x, y);
                jsq[l * jcols + c] = __dmul_rn(d, d);
            } else {           // (prev_join_rep[p[1:]] - current_join_rep[p[:-1]])**2
                x = __dmul_rn((double)a.JC_unw[(a.prev_row0 + pn) * a.Dj + a.prev_col0 + c], a.wj[a.prev_col0 + c]);
                y = __dmul_rn((double)a.JC_unw[(a.cur_row0 + p) * a.Dj + a.cur_col0 + c], a.wj[a.cur_col0 + c]);
                const double d = __dsub_rn(x, y);
                jsq[l * jcols + c] = __dmul_rn(d, d);
            }
        }
    }
}

void launch_path_scores(const GreedyLayout &g, int mode, const float *F_unw, int Dt, const double *wt,
                        const float *JC_unw, int Dj, const double *wj, const double *Q,
                        const int64_t *path, int64_t L, double *tsq, double *jsq, int jcols,
                        hipStream_t s)
{
    GreedyArgs a{};
    a.JC_unw = JC_unw; a.Dj = Dj; a.wj = wj;
    a.F_unw = F_unw; a.Dt = Dt; a.wt = wt;
    if (mode == 0) { a.me = 1; a.nep = 1; a.ep[0] = 0; }
    else {
        a.me = g.me;
        if (g.last_frame_as_target && g.me > 1) { a.nep = 2; a.ep[0] = 0; a.ep[1] = g.me - 1; }
        else { a.nep = g.me; for (int e = 0; e < g.me; ++e) a.ep[e] = e; }
    }
    a.prev_col0 = g.prev_col0; a.cur_col0 = g.cur_col0; a.jdim = g.jdim;
    a.prev_row0 = g.prev_row0; a.cur_row0 = g.cur_row0; a.Nwin = g.Nwin;
    a.Q = Q;
    hipLaunchKernelGGL(path_scores_kernel, dim3((unsigned)L), dim3(256), 0, s, a, mode, path, L, jcols,
                       (int64_t)0, tsq, jsq);
}

}  // namespace snk
