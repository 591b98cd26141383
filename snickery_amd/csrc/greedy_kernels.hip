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
// Per step ONE launch of greedy_step_kernel (see the comment on the kernel).
#include "snk_internal.h"
#include <float.h>

namespace snk {

#define GR_R 256          // windows per workgroup (one thread per window)
#define GR_CC 32          // columns per staged chunk
#define GR_MAX_EP 16      // max multiepoch

struct GreedyArgs {
    const float *JC_unw; int Dj; const double *wj;
    const float *F_unw; int Dt; const double *wt;
    int me, nep; int ep[GR_MAX_EP];       // epochs of the window that enter the target term
    int prev_col0, cur_col0, jdim;
    int64_t prev_row0, cur_row0, Nwin;
    int64_t n_jc_rows, n_f_rows;          // matrix heights (clamp for the ragged last workgroup)
    const double *Q;                      // (T, Dt) weighted targets, row-major
    double *prev_vec;                     // (jdim): read by every block, rewritten by the last one
};

// One step of the greedy search = ONE launch:
//   every workgroup scans GR_R consecutive windows (thread t owns window i0+t): the join columns
//   and then, epoch by epoch, the target columns stream through LDS in 32-column chunks
//   (coalesced loads: a wave instruction reads 32 consecutive floats of two rows), register-staged
//   one chunk ahead; thread t accumulates its window's squared distance in the canonical column
//   order.  The workgroup's (min, argmin) goes to global memory; the workgroup that arrives LAST
//   (agent-scope release/acquire around an arrival counter) reduces all partial results, appends
//   the winner to the path and fetches its `current_join_rep` row as the next step's `prev`.
__global__ void __launch_bounds__(GR_R)
greedy_step_kernel(GreedyArgs a, int64_t step, double *__restrict__ blk_min, int64_t *__restrict__ blk_arg,
                   unsigned int *__restrict__ arrive, int64_t *__restrict__ path, double *__restrict__ dist)
{
    __shared__ float buf[2][GR_R][GR_CC + 1];
    __shared__ double ref[GR_CC], wgt[GR_CC];
    __shared__ double red_v[GR_R];
    __shared__ int64_t red_i[GR_R];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * GR_R;

    // chunk schedule: part 0 = join columns, parts 1..nep = target columns of epoch ep[k]
    const int jch = (a.jdim + GR_CC - 1) / GR_CC, tch = (a.Dt + GR_CC - 1) / GR_CC;
    const int n_chunks = jch + a.nep * tch;
    auto chunk_info = [&](int c, const float *&base, int &pitch, int64_t &row0, int64_t &nrows, int &col0,
                          int &ncols, const double *&w, const double *&rf) {
        if (c < jch) {
            base = a.JC_unw; pitch = a.Dj; row0 = a.prev_row0 + i0; nrows = a.n_jc_rows;
            col0 = a.prev_col0 + c * GR_CC; ncols = min(GR_CC, a.jdim - c * GR_CC);
            w = a.wj + col0; rf = a.prev_vec + c * GR_CC;
        } else {
            const int k = (c - jch) / tch, cc = (c - jch) % tch;
            base = a.F_unw; pitch = a.Dt; row0 = i0 + a.ep[k]; nrows = a.n_f_rows;
            col0 = cc * GR_CC; ncols = min(GR_CC, a.Dt - cc * GR_CC);
            w = a.wt + col0; rf = a.Q + (step * a.me + a.ep[k]) * a.Dt + col0;
        }
    };
    float stage[GR_CC];      // this thread's share of the next chunk
    auto fetch = [&](int c) {
        const float *base; int pitch, col0, ncols; int64_t row0, nrows; const double *w, *rf;
        chunk_info(c, base, pitch, row0, nrows, col0, ncols, w, rf);
#pragma unroll
        for (int j = 0; j < GR_CC; ++j) {
            const int e = tid + j * GR_R;             // lanes -> consecutive columns of a row
            const int r = e / GR_CC, cc = e % GR_CC;
            int64_t row = row0 + r;
            if (row >= nrows) row = nrows - 1;
            stage[j] = (cc < ncols) ? base[row * pitch + col0 + cc] : 0.0f;
        }
    };
    double acc_j = 0.0, acc_t = 0.0;
    fetch(0);
    for (int c = 0; c < n_chunks; ++c) {
        const float *base; int pitch, col0, ncols; int64_t row0, nrows; const double *w, *rf;
        chunk_info(c, base, pitch, row0, nrows, col0, ncols, w, rf);
        float (*B)[GR_CC + 1] = buf[c & 1];
#pragma unroll
        for (int j = 0; j < GR_CC; ++j) {
            const int e = tid + j * GR_R;
            B[e / GR_CC][e % GR_CC] = stage[j];
        }
        if (tid < GR_CC) {
            ref[tid] = (tid < ncols) ? rf[tid] : 0.0;
            wgt[tid] = (tid < ncols) ? w[tid] : 0.0;
        }
        __syncthreads();
        if (c + 1 < n_chunks) fetch(c + 1);           // in flight while this chunk is accumulated
        double acc = (c < jch) ? acc_j : acc_t;
        for (int cc = 0; cc < ncols; ++cc) {
            const double v = __dmul_rn((double)B[tid][cc], wgt[cc]);
            const double d = __dsub_rn(v, ref[cc]);
            acc = __dadd_rn(acc, __dmul_rn(d, d));
        }
        if (c < jch) acc_j = acc; else acc_t = acc;
        __syncthreads();
    }
    double best = DBL_MAX;
    int64_t arg = INT64_MAX;
    if (i0 + tid < a.Nwin) { best = __dadd_rn(acc_j, acc_t); arg = i0 + tid; }
    red_v[tid] = best; red_i[tid] = arg;
    __syncthreads();
    for (int off = GR_R / 2; off > 0; off >>= 1) {
        if (tid < off) {
            const double v2 = red_v[tid + off];
            const int64_t i2 = red_i[tid + off];
            if (v2 < red_v[tid] || (v2 == red_v[tid] && i2 < red_i[tid])) { red_v[tid] = v2; red_i[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        blk_min[blockIdx.x] = red_v[0];
        blk_arg[blockIdx.x] = red_i[0];
        // publish, then arrive (MI355X guide G16: agent-scope release before the counter)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == gridDim.x - 1);
        if (is_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!is_last) return;
    // ---- last workgroup: global argmin (lowest index on exact ties), path, next prev ----
    best = DBL_MAX; arg = INT64_MAX;
    for (int b = tid; b < (int)gridDim.x; b += GR_R) {
        const double v = __hip_atomic_load(&blk_min[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t i = __hip_atomic_load(&blk_arg[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v < best || (v == best && i < arg)) { best = v; arg = i; }
    }
    red_v[tid] = best; red_i[tid] = arg;
    __syncthreads();
    for (int off = GR_R / 2; off > 0; off >>= 1) {
        if (tid < off) {
            const double v2 = red_v[tid + off];
            const int64_t i2 = red_i[tid + off];
            if (v2 < red_v[tid] || (v2 == red_v[tid] && i2 < red_i[tid])) { red_v[tid] = v2; red_i[tid] = i2; }
        }
        __syncthreads();
    }
    const int64_t ix = red_i[0];
    if (tid == 0) {
        path[step] = ix;
        if (dist) dist[step] = __dsqrt_rn(red_v[0]);
        *arrive = 0;                                   // re-armed for the next step (next launch)
    }
    // prev_join_vector = current_join_rep[ix]   (synth_simple.py:501)
    const float *src = a.JC_unw + (a.cur_row0 + ix) * a.Dj + a.cur_col0;
    for (int c = tid; c < a.jdim; c += GR_R)
        a.prev_vec[c] = __dmul_rn((double)src[c], a.wj[a.cur_col0 + c]);
}

__global__ void greedy_init_prev_kernel(GreedyArgs a, int64_t start_state, unsigned int *arrive)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0) *arrive = 0;
    if (c >= a.jdim) return;
    if (start_state < 0) { a.prev_vec[c] = 0.0; return; }     // np.zeros((n,)) synth_simple.py:467-468
    const float *src = a.JC_unw + (a.prev_row0 + start_state) * a.Dj + a.prev_col0;
    a.prev_vec[c] = __dmul_rn((double)src[c], a.wj[a.prev_col0 + c]);
}

static void fill_args(GreedyArgs &a, const GreedyLayout &g, const float *F_unw, int Dt, const double *wt,
                      const float *JC_unw, int Dj, const double *wj, const double *Q, double *prev_vec,
                      bool greedy_mode)
{
    a.JC_unw = JC_unw; a.Dj = Dj; a.wj = wj;
    a.F_unw = F_unw; a.Dt = Dt; a.wt = wt;
    if (!greedy_mode) { a.me = 1; a.nep = 1; a.ep[0] = 0; }
    else {
        a.me = g.me;
        if (g.last_frame_as_target && g.me > 1) { a.nep = 2; a.ep[0] = 0; a.ep[1] = g.me - 1; }
        else { a.nep = g.me; for (int e = 0; e < g.me; ++e) a.ep[e] = e; }
    }
    a.prev_col0 = g.prev_col0; a.cur_col0 = g.cur_col0; a.jdim = g.jdim;
    a.prev_row0 = g.prev_row0; a.cur_row0 = g.cur_row0; a.Nwin = g.Nwin;
    a.n_jc_rows = g.Nwin + g.me;          // join_contexts has N+1 = Nwin + me rows
    a.n_f_rows = g.Nwin + g.me - 1;       // N
    a.Q = Q; a.prev_vec = prev_vec;
}

void launch_greedy(const GreedyLayout &g, const float *F_unw, int Dt, const double *wt,
                   const float *JC_unw, int Dj, const double *wj, const double *Q,
                   int64_t nsteps, int64_t start_state, double *prev_vec, double *blk_min,
                   int64_t *blk_arg, int nblk, unsigned int *arrive, int64_t *path, double *dist, hipStream_t s)
{
    GreedyArgs a{};
    fill_args(a, g, F_unw, Dt, wt, JC_unw, Dj, wj, Q, prev_vec, true);
    hipLaunchKernelGGL(greedy_init_prev_kernel, dim3((g.jdim + 255) / 256), dim3(256), 0, s, a, start_state, arrive);
    for (int64_t st = 0; st < nsteps; ++st)
        hipLaunchKernelGGL(greedy_step_kernel, dim3(nblk), dim3(GR_R), 0, s, a, st, blk_min, blk_arg, arrive,
                           path, dist);
}

size_t greedy_shmem_bytes(const GreedyLayout &, int, int) { return 0; }    // static LDS only (fits any shape)

int greedy_blocks(const GreedyLayout &g, int, int) { return (int)((g.Nwin + GR_R - 1) / GR_R); }

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
    fill_args(a, g, F_unw, Dt, wt, JC_unw, Dj, wj, Q, nullptr, mode == 1);
    hipLaunchKernelGGL(path_scores_kernel, dim3((unsigned)L), dim3(256), 0, s, a, mode, path, L, jcols,
                       (int64_t)0, tsq, jsq);
}

}  // namespace snk
