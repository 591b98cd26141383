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
// matrix and row i of the unweighted join matrix as the HDF5 file stores them (rows padded to a
// multiple of 4 floats on the device, so every access is 16 bytes wide) and applies the float64
// stream weights on the fly (fl64(f32 * w) is bit-identical to the
// reference's speech_manip.weight()).  Every step is therefore a stream over (Dj + Dt) * 4
// bytes per unit (HBM-bound by design).  Squared distances are accumulated in the canonical
// oracle order (column by column, separately rounded sub/mul/add), so the argmin is bit-exact.
//
// Per step ONE launch of greedy_step_kernel (see the comment on the kernel).
#include "snk_internal.h"
#include <float.h>

namespace snk {

#define GR_R 256          // windows per workgroup (one thread per window)
#define GR_CC 32          // columns per staged chunk (8 float4 per row)
#define GR_NSTG 3         // chunks in flight in registers per thread
#define GR_NFL 18         // 16-byte loads per thread that fetch a resident target block
#define GR_LP 36          // LDS row pitch in floats: 16-byte aligned rows, conflict-free 128-bit access
#define GR_MAX_EP 16      // max multiepoch
#define GR_S1 256          // arrival counters of the first / second level (see greedy_finish_step)
#define GR_S2 16

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GreedyArgs {
    const float *JC_unw; int Jp, Dj; const double *wj;   // Jp / Fp: row pitch in floats (multiple of 4,
    const float *F_unw; int Fp, Dt; const double *wt;    // zero-filled padding columns)
    int me, nep; int ep[GR_MAX_EP];       // epochs of the window that enter the target term
    int prev_col0, cur_col0, jdim;
    int64_t prev_row0, cur_row0, Nwin;
    int64_t n_jc_rows, n_f_rows;          // matrix heights (clamp for the ragged last workgroup)
    const double *Q;                      // (T, Dt) weighted targets, row-major
};

__device__ __forceinline__ int greedy_join_chunks(const GreedyArgs &a) { return (a.jdim + GR_CC - 1) / GR_CC; }
__device__ __forceinline__ int greedy_target_chunks(const GreedyArgs &a) { return (a.Dt + GR_CC - 1) / GR_CC; }

// The (weight, reference) pair of every column of the scan in chunk order -- join columns against
// `prev`, then the target columns of each epoch against the query rows of this step -- padded with
// (0, 0) to whole chunks.  Built by one workgroup for the NEXT step; the scan stages it in LDS and reads it with
// 16-byte LDS broadcasts.
// prev_row < 0: prev = 0 (np.zeros, synth_simple.py:467-468).
__device__ void greedy_write_table(const GreedyArgs &a, int64_t step, int64_t prev_row, bool prev_is_current,
                                   double *__restrict__ tab, int tid, int nthreads)
{
    const int jch = greedy_join_chunks(a), tch = greedy_target_chunks(a);
    const int n = (jch + a.nep * tch) * GR_CC;
    for (int e = tid; e < n; e += nthreads) {
        const int c = e / GR_CC, cc = e % GR_CC;
        double w = 0.0, ref = 0.0;
        if (c < jch) {
            const int col = c * GR_CC + cc;
            if (col < a.jdim) {
                w = a.wj[a.prev_col0 + col];
                if (prev_row >= 0) {
                    // prev_join_vector = current_join_rep[ix] (synth_simple.py:501) or prev_join_rep[start]
                    const int col0 = prev_is_current ? a.cur_col0 : a.prev_col0;
                    const int64_t row0 = prev_is_current ? a.cur_row0 : a.prev_row0;
                    ref = __dmul_rn((double)a.JC_unw[(row0 + prev_row) * a.Jp + col0 + col], a.wj[col0 + col]);
                }
            }
        } else {
            const int k = (c - jch) / tch, col = ((c - jch) % tch) * GR_CC + cc;
            if (col < a.Dt) {
                w = a.wt[col];
                ref = a.Q[(step * a.me + a.ep[k]) * a.Dt + col];
            }
        }
        tab[2 * e] = w;
        tab[2 * e + 1] = ref;
    }
}

// acc += sum over `ngroups` x 8 consecutive columns of (fl64(x) * w - ref)^2, in the canonical column
// order with separately rounded operations.  `row` points at the window's columns in LDS, `tc` at
// their (w, ref) pairs in the step's table.  Eight columns are in flight at a time: their
// conversions, products and squares are independent, only the eight final additions form a chain.
__device__ __forceinline__ double greedy_accumulate_chunk(const float *row, const double *__restrict__ tc, double acc,
                                                          int ngroups = GR_CC / 8)
{
    // `tc`: the chunk's 32 (w, ref) pairs in GLOBAL memory at a wave-uniform address: scalar loads
    // into SGPRs (no LDS bandwidth, no vector registers), requested one group of eight columns ahead
    struct Group { float4 x0, x1; double w[8], r[8]; };
    auto load = [&](int g, Group &G) {
        G.x0 = *reinterpret_cast<const float4 *>(row + 8 * g);
        G.x1 = *reinterpret_cast<const float4 *>(row + 8 * g + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) { G.w[i] = tc[2 * (8 * g + i)]; G.r[i] = tc[2 * (8 * g + i) + 1]; }
    };
    auto consume = [&](const Group &G) {
        const float xs[8] = {G.x0.x, G.x0.y, G.x0.z, G.x0.w, G.x1.x, G.x1.y, G.x1.z, G.x1.w};
        double sq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double d = __dsub_rn(__dmul_rn((double)xs[i], G.w[i]), G.r[i]);
            sq[i] = __dmul_rn(d, d);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __dadd_rn(acc, sq[i]);     // padded columns: w = ref = 0 adds +0.0
    };
    // the loads of group g+1 are issued (and pinned there) before group g is consumed; `ngroups`
    // (even) groups of eight columns, contiguous in the row and in the table
    Group A, B;
    load(0, A);
    __builtin_amdgcn_sched_barrier(0);
    for (int g = 0; g < ngroups; g += 2) {
        load(g + 1, B);
        __builtin_amdgcn_sched_barrier(0);
        consume(A);
        __builtin_amdgcn_sched_barrier(0);
        load(g + 2 < ngroups ? g + 2 : g, A);           // the last trip re-reads its own group: never consumed
        __builtin_amdgcn_sched_barrier(0);
        consume(B);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// Tail of a step, shared by both scan kernels: workgroup (min, argmin) -> global memory; the
// workgroup that arrives LAST (sc1 stores drained before an arrival counter, sc1 loads after) reduces all
// partial results, appends the winner to the path and writes the next step's table (its
// `current_join_rep` row is the next `prev`).
__device__ void greedy_finish_step(const GreedyArgs &a, int64_t step, int64_t nsteps, double best, int64_t arg,
                                   double *__restrict__ tab_next, double *__restrict__ blk_min,
                                   int64_t *__restrict__ blk_arg, unsigned int *__restrict__ arrive,
                                   int64_t *__restrict__ path, double *__restrict__ dist,
                                   double *red_v, int64_t *red_i, int *is_last_p)
{
    const int tid = threadIdx.x;
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
        // publish, then arrive.  The 16 handed-off bytes are written with sc1 (agent-scope) stores,
        // drained before the counter add, and read back with sc1 loads by the last workgroup: no
        // release/acquire fence.  (A release fence per workgroup writes back the XCD's L2 and
        // serialises at ~1.2 us per workgroup and XCD: 1 ms per step at 5 860 workgroups, measured.)
        __hip_atomic_store(&blk_min[blockIdx.x], red_v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&blk_arg[blockIdx.x], red_i[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Arrival tree: 256 -> 16 -> 1 monotonic counters, each on its own 128-byte line (one counter
        // for all workgroups serialises at ~0.1 us per add: 0.6 ms per step at 5 860 workgroups,
        // measured).  A workgroup climbs a level only when its add completes the counter's quota.
        const unsigned int nb = gridDim.x, b = blockIdx.x, round = (unsigned int)step + 1u;
        const unsigned int S1 = nb < GR_S1 ? nb : GR_S1, s1 = b % S1, q1 = (nb - s1 + S1 - 1) / S1;
        bool last = false;
        if (__hip_atomic_fetch_add(arrive + 32 * s1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == round * q1 - 1) {
            const unsigned int S2 = S1 < GR_S2 ? S1 : GR_S2, s2 = s1 % S2, q2 = (S1 - s2 + S2 - 1) / S2;
            if (__hip_atomic_fetch_add(arrive + 32 * (GR_S1 + s2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == round * q2 - 1)
                last = __hip_atomic_fetch_add(arrive + 32 * (GR_S1 + GR_S2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       == round * S2 - 1;
        }
        *is_last_p = last;
    }
    __syncthreads();
    if (!*is_last_p) return;
    // ---- last workgroup: global argmin (lowest index on exact ties), path, next step's table ----
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
    }
    if (step + 1 < nsteps) greedy_write_table(a, step + 1, ix, true, tab_next, tid, GR_R);
}

// One step of the greedy search = ONE launch:
//   every workgroup scans GR_R consecutive windows (thread t owns window i0+t) with 16-byte global
//   loads, LDS writes and LDS reads (rows are padded to a multiple of 4 floats on the device).
//   Target term: the workgroup's GR_R + me - 1 target rows are read once into LDS and every epoch
//   of every window is accumulated from there (wider targets: streamed chunk by chunk instead).
//   Join term: 32-column chunks stream through LDS, GR_NSTG chunks in flight in registers.
//   Thread t accumulates its window's squared distance in the canonical column order, weights
//   and references coming from the step's table.  The workgroup's (min, argmin) goes to global
//   memory; the workgroup that arrives LAST (sc1 stores / arrival tree / sc1 loads) reduces all
//   partial results, appends the winner to the path and writes the next step's table (its
//   `current_join_rep` row is the next `prev`).
__global__ void __launch_bounds__(GR_R)
greedy_step_kernel(GreedyArgs a, int64_t step, int64_t nsteps, const double *__restrict__ tab,
                   double *__restrict__ tab_next, double *__restrict__ blk_min, int64_t *__restrict__ blk_arg,
                   unsigned int *__restrict__ arrive, int64_t *__restrict__ path, double *__restrict__ dist)
{
    __shared__ __align__(16) float buf[2][GR_R][GR_LP];
    __shared__ double red_v[GR_R];
    __shared__ int64_t red_i[GR_R];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * GR_R;

    // chunk schedule: part 0 = join columns, parts 1..nep = target columns of epoch ep[k]
    const int jch = greedy_join_chunks(a), tch = greedy_target_chunks(a);
    double acc_j = 0.0, acc_t = 0.0;

    // Target term from a resident block: when the GR_R + me - 1 target rows of this workgroup fit the
    // LDS of the chunk buffers (magphase-60: 261 rows x 68 floats), they are read from HBM ONCE and
    // every epoch of every window is accumulated from LDS; the chunked loop below then streams the
    // join columns only.  Otherwise the target columns go through the chunk loop too, epoch by epoch
    // (each target row is then re-read once per epoch, through L2).
    const int frows = GR_R + a.me - 1, fpitch = a.Fp + 4;
    // (whole 32-column chunks only: the accumulation reads 32 floats per chunk from a row)
    const bool resident = (a.Fp % GR_CC) == 0 && (size_t)frows * fpitch * sizeof(float) <= sizeof(buf);
    // Chunk loop: join columns (and the target columns when they are not resident).  GR_NSTG chunks
    // are in flight in registers.  The loads are inline asm: the compiler sinks ordinary loads of a
    // software pipeline to their use (measured: no load was in flight behind the arithmetic), and it
    // does not count asm loads in its own vmcnt waits, so every wait here is explicit.  Stage s is
    // consumed in chunk order, so before chunk c at most GR_NSTG-1 younger chunks (8 loads each) may
    // still be outstanding.
    const int n_chunks = resident ? jch : jch + a.nep * tch;
    const int lr = tid >> 3, lq = tid & 7;          // this lane's row (mod 32) and float4 of a chunk
    f32x4 stage[GR_NSTG][8];
    auto fetch = [&](int c, f32x4 (&st)[8]) {
        if (c >= n_chunks) c = n_chunks - 1;          // surplus request: re-read, never consumed
        const float *base; int pitch, col0; int64_t row0, nrows;
        if (c < jch) {
            base = a.JC_unw; pitch = a.Jp; row0 = a.prev_row0 + i0; nrows = a.n_jc_rows;
            col0 = a.prev_col0 + c * GR_CC;
        } else {
            const int k = (c - jch) / tch;
            base = a.F_unw; pitch = a.Fp; row0 = i0 + a.ep[k]; nrows = a.n_f_rows;
            col0 = ((c - jch) % tch) * GR_CC;
        }
        col0 += 4 * lq;
        if (col0 > pitch - 4) col0 = pitch - 4;       // float4s beyond the padded row: finite data, weight 0
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int64_t row = row0 + 32 * j + lr;         // 8 lanes cover 128 bytes of one row
            if (row >= nrows) row = nrows - 1;
            const float *src = base + row * pitch + col0;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st[j]) : "v"(src) : "memory");
        }
    };
    if (!resident) {
#pragma unroll
        for (int s = 0; s < GR_NSTG; ++s) fetch(s, stage[s]);
    }
    if (resident) {
        float *Fs = &buf[0][0][0];
        const int fq = a.Fp >> 2, nq = frows * fq;       // float4s of the block
        // all of a thread's loads are in flight together (GR_NFL x 16 bytes; the block holds at most
        // sizeof(buf) / 16 = 4608 float4 = 18 per thread)
        f32x4 v[GR_NFL];
#pragma unroll
        for (int u = 0; u < GR_NFL; ++u) {
            const int e = u * GR_R + tid;
            int64_t row = i0 + (e < nq ? e / fq : 0);
            if (row >= a.n_f_rows) row = a.n_f_rows - 1;
            v[u] = *reinterpret_cast<const f32x4 *>(a.F_unw + row * a.Fp + 4 * (e < nq ? e % fq : 0));
        }
#pragma unroll
        for (int u = 0; u < GR_NFL; ++u) {
            const int e = u * GR_R + tid;
            if (e < nq) *reinterpret_cast<f32x4 *>(Fs + (size_t)(e / fq) * fpitch + 4 * (e % fq)) = v[u];
        }
        __syncthreads();
        // the first join chunks are requested now and land behind the target arithmetic
#pragma unroll
        for (int s = 0; s < GR_NSTG; ++s) fetch(s, stage[s]);
        for (int k = 0; k < a.nep; ++k) {
            const float *frow = Fs + (size_t)(tid + a.ep[k]) * fpitch;
            acc_t = greedy_accumulate_chunk(frow, tab + 2 * (size_t)(jch + k * tch) * GR_CC, acc_t, tch * (GR_CC / 8));
        }
        __syncthreads();                                 // the chunk buffers alias the block
    }
    auto consume = [&](int c, f32x4 (&st)[8]) {
        float (*B)[GR_LP] = buf[c & 1];
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4 *>(&B[32 * j + lr][4 * lq]) = st[j];
        __syncthreads();
        fetch(c + GR_NSTG, st);                       // refill this stage
        const double acc = greedy_accumulate_chunk(&B[tid][0], tab + 2 * (size_t)c * GR_CC, (c < jch) ? acc_j : acc_t);
        if (c < jch) acc_j = acc; else acc_t = acc;
    };
    for (int c0 = 0; c0 < n_chunks; c0 += GR_NSTG) {
#pragma unroll
        for (int s = 0; s < GR_NSTG; ++s) {
            // chunk c0+s: everything but the (GR_NSTG-1) younger chunks has landed
            asm volatile("s_waitcnt vmcnt(%8)"
                         : "+v"(stage[s][0]), "+v"(stage[s][1]), "+v"(stage[s][2]), "+v"(stage[s][3]),
                           "+v"(stage[s][4]), "+v"(stage[s][5]), "+v"(stage[s][6]), "+v"(stage[s][7])
                         : "n"(8 * (GR_NSTG - 1)) : "memory");
            if (c0 + s < n_chunks) consume(c0 + s, stage[s]);      // uniform
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus requests of the last trip
    double best = DBL_MAX;
    int64_t arg = INT64_MAX;
    if (i0 + tid < a.Nwin) { best = __dadd_rn(acc_j, acc_t); arg = i0 + tid; }
    greedy_finish_step(a, step, nsteps, best, arg, tab_next, blk_min, blk_arg, arrive, path, dist, red_v, red_i, &is_last);
}

__global__ void greedy_init_kernel(GreedyArgs a, int64_t start_state, double *__restrict__ tab, unsigned int *arrive)
{
    for (int i = threadIdx.x; i < 32 * (GR_S1 + GR_S2 + 1); i += blockDim.x) arrive[i] = 0;
    greedy_write_table(a, 0, start_state, false, tab, threadIdx.x, blockDim.x);
}

static void fill_args(GreedyArgs &a, const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt,
                      const float *JC_unw, int Jp, int Dj, const double *wj, const double *Q, bool greedy_mode)
{
    a.JC_unw = JC_unw; a.Jp = Jp; a.Dj = Dj; a.wj = wj;
    a.F_unw = F_unw; a.Fp = Fp; a.Dt = Dt; a.wt = wt;
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
    a.Q = Q;
}

size_t greedy_counter_bytes() { return (size_t)32 * (GR_S1 + GR_S2 + 1) * sizeof(unsigned int); }

// doubles of one (weight, reference) table; the caller provides two (steps alternate)
size_t greedy_table_doubles(const GreedyLayout &g, int Dt)
{
    const int nep = (g.last_frame_as_target && g.me > 1) ? 2 : g.me;
    const int jch = (g.jdim + GR_CC - 1) / GR_CC;
    return (size_t)(jch + nep * ((Dt + GR_CC - 1) / GR_CC)) * GR_CC * 2;
}

void launch_greedy(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt,
                   const float *JC_unw, int Jp, int Dj, const double *wj, const double *Q,
                   int64_t nsteps, int64_t start_state, double *tables, double *blk_min,
                   int64_t *blk_arg, int nblk, unsigned int *arrive, int64_t *path, double *dist, hipStream_t s)
{
    GreedyArgs a{};
    fill_args(a, g, F_unw, Fp, Dt, wt, JC_unw, Jp, Dj, wj, Q, true);
    double *tab[2] = {tables, tables + greedy_table_doubles(g, Dt)};
    hipLaunchKernelGGL(greedy_init_kernel, dim3(1), dim3(256), 0, s, a, start_state, tab[0], arrive);
    for (int64_t st = 0; st < nsteps; ++st)
        hipLaunchKernelGGL(greedy_step_kernel, dim3(nblk), dim3(GR_R), 0, s, a, st, nsteps, tab[st & 1],
                           tab[(st + 1) & 1], blk_min, blk_arg, arrive, path, dist);
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
        const double f = __dmul_rn((double)a.F_unw[(p + a.ep[k]) * a.Fp + c], a.wt[c]);
        const double q = a.Q[(l * a.me + a.ep[k]) * a.Dt + c];
        const double d = __dsub_rn(f, q);
        tsq[l * tcols + e] = __dmul_rn(d, d);
    }
    if (l + 1 < L) {
        const int64_t pn = path[l + 1];
        for (int c = tid; c < jcols; c += blockDim.x) {
            double x, y;
            if (mode == 0) {   // (unit_end_data[p[:-1]] - unit_start_data[p[1:]])**2
                x = __dmul_rn((double)a.JC_unw[(p + 1) * a.Jp + c], a.wj[c]);
                y = __dmul_rn((double)a.JC_unw[pn * a.Jp + c], a.wj[c]);
                const double d = __dsub_rn(x, y);
                jsq[l * jcols + c] = __dmul_rn(d, d);
            } else {           // (prev_join_rep[p[1:]] - current_join_rep[p[:-1]])**2
                x = __dmul_rn((double)a.JC_unw[(a.prev_row0 + pn) * a.Jp + a.prev_col0 + c], a.wj[a.prev_col0 + c]);
                y = __dmul_rn((double)a.JC_unw[(a.cur_row0 + p) * a.Jp + a.cur_col0 + c], a.wj[a.cur_col0 + c]);
                const double d = __dsub_rn(x, y);
                jsq[l * jcols + c] = __dmul_rn(d, d);
            }
        }
    }
}

void launch_path_scores(const GreedyLayout &g, int mode, const float *F_unw, int Fp, int Dt, const double *wt,
                        const float *JC_unw, int Jp, int Dj, const double *wj, const double *Q,
                        const int64_t *path, int64_t L, double *tsq, double *jsq, int jcols,
                        hipStream_t s)
{
    GreedyArgs a{};
    fill_args(a, g, F_unw, Fp, Dt, wt, JC_unw, Jp, Dj, wj, Q, mode == 1);
    hipLaunchKernelGGL(path_scores_kernel, dim3((unsigned)L), dim3(256), 0, s, a, mode, path, L, jcols,
                       (int64_t)0, tsq, jsq);
}

}  // namespace snk
