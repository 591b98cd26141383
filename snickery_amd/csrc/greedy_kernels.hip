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
// matrix and row i of the unweighted join matrix -- from a lane-major copy of the two matrices
// that the device keeps for this scan (greedy_tile_kernel; built once per database and layout) --
// and applies the float64 stream weights on the fly (fl64(f32 * w) is bit-identical to the
// reference's speech_manip.weight()).  Every step is therefore a stream over (Dj + Dt) * 4
// bytes per unit (HBM-bound by design).  Squared distances are accumulated in the canonical
// oracle order (column by column, separately rounded sub/mul/add), so the argmin is bit-exact.
//
// Per step ONE launch of greedy_step_kernel (see the comment on the kernel).
#include "greedy_common.h"

namespace snk {

// The table of a step: per column of the scan, in chunk order, the weight and the reference of every
// utterance of the scan -- join columns against that utterance's `prev`, target columns against its
// query rows of this step -- padded with zeros to whole chunks.  UB = 1: (w, ref) pairs, 16 bytes per
// column; UB = 3: (w, ref0, ref1, ref2), 32 bytes.  Built by one workgroup for the NEXT step.
// prev_row < 0: prev = 0 (np.zeros, synth_simple.py:467-468).  An utterance that has no such step
// (shorter than the others) gets zero references; its result is ignored.
template <int UB>
__device__ void greedy_write_table(const GreedyArgs &a, int64_t step, const int64_t (&prev_row)[UB], bool prev_is_current,
                                   double *__restrict__ tab, int tid, int nthreads)
{
    constexpr int TS = (UB == 1) ? 2 : 4;
    const int jch = greedy_join_chunks(a), tch = greedy_target_chunks(a);
    const int nT = a.nep * tch, n = (jch + nT) * GR_CC;
    for (int e = tid; e < n; e += nthreads) {
        const int c = e / GR_CC, cc = e % GR_CC;
        double w = 0.0, ref[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) ref[u] = 0.0;
        int idx;
        if (greedy_chunk_slot(a, jch, nT, c, &idx)) {
            const int col = idx * GR_CC + cc;
            if (col < a.jdim) {
                w = a.wj[a.prev_col0 + col];
                // prev_join_vector = current_join_rep[ix] (synth_simple.py:501) or prev_join_rep[start]
                const int col0 = prev_is_current ? a.cur_col0 : a.prev_col0;
                const int64_t row0 = prev_is_current ? a.cur_row0 : a.prev_row0;
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (u < a.nu && step < a.nsteps_u[u] && prev_row[u] >= 0)
                        ref[u] = __dmul_rn((double)a.JC_unw[(row0 + prev_row[u]) * a.Jp + col0 + col], a.wj[col0 + col]);
            }
        } else {
            const int k = idx / tch, col = (idx % tch) * GR_CC + cc;
            if (col < a.Dt) {
                w = a.wt[col];
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (u < a.nu && step < a.nsteps_u[u])
                        ref[u] = a.Q[(a.q_off[u] + step * a.me + a.ep[k]) * a.Dt + col];
            }
        }
        tab[TS * e] = w;
#pragma unroll
        for (int u = 0; u < UB; ++u) tab[TS * e + 1 + u] = ref[u];
        if (UB == 2) tab[TS * e + 3] = 0.0;
    }
}

// Lane-major tiles of the scan columns.  The scan gives every window (= database row) to one lane
// and walks its columns in order, so the row-major matrices would need a transposition through LDS
// in front of the arithmetic (two phases per workgroup, a barrier per chunk, 78 KB of LDS and only
// two workgroups per compute unit: measured 4.5 TB/s with the arithmetic removed).  Instead the
// device keeps a second copy in the order the scan reads:
//     tile[row / 64][q][row % 64] = float4 of columns col0 + 4q .. col0 + 4q + 3 of `row`
// (Q float4 groups per row, zero-filled beyond `ncols`), so that ONE 16-byte load per lane fetches
// four columns of 64 consecutive windows as 1 KB of consecutive addresses, straight into the
// registers the arithmetic reads.  No LDS, no barrier in the scan.
__global__ void greedy_tile_kernel(const float *__restrict__ src, int pitch, int64_t nrows, int col0, int ncols,
                                   int Q, int64_t n_elems, f32x4 *__restrict__ dst)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_elems) return;
    const int lane = (int)(idx & 63);
    const int64_t g = idx >> 6;
    const int q = (int)(g % Q);
    const int64_t r = (g / Q) * 64 + lane;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < nrows) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = 4 * q + e;
            if (col < ncols) v[e] = src[r * pitch + col0 + col];
        }
    }
    dst[idx] = v;
}

// Tail of a step: per utterance, workgroup (min, argmin) -> global memory; the workgroup that arrives
// LAST (sc1 stores drained before an arrival counter, sc1 loads after) reduces all partial results,
// appends the winners to the paths and writes the next step's table (a winner's `current_join_rep` row
// is that utterance's next `prev`).
template <int UB>
__device__ void greedy_finish_step(const GreedyArgs &a, int64_t step, int64_t nsteps, const double (&best_in)[UB],
                                   const int64_t (&arg_in)[UB],
                                   double *__restrict__ tab_next, double *__restrict__ blk_min,
                                   int64_t *__restrict__ blk_arg, unsigned int *__restrict__ arrive,
                                   int64_t *__restrict__ path, double *__restrict__ dist,
                                   double *red_v, int64_t *red_i, int *is_last_p)
{
    const int tid = threadIdx.x;
    const unsigned int nb = gridDim.x;
    int top = 1;                                          // half of the next power of two >= blockDim.x
    while (2 * top < (int)blockDim.x) top <<= 1;
    auto block_reduce = [&](double best, int64_t arg) {
        red_v[tid] = best; red_i[tid] = arg;
        __syncthreads();
        for (int off = top; off > 0; off >>= 1) {
            if (tid < off && tid + off < (int)blockDim.x) {
                const double v2 = red_v[tid + off];
                const int64_t i2 = red_i[tid + off];
                if (v2 < red_v[tid] || (v2 == red_v[tid] && i2 < red_i[tid])) { red_v[tid] = v2; red_i[tid] = i2; }
            }
            __syncthreads();
        }
    };
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        block_reduce(best_in[u], arg_in[u]);
        // publish.  The handed-off bytes are written with sc1 (agent-scope) stores, drained before the
        // counter add, and read back with sc1 loads by the last workgroup: no release/acquire fence.
        // (A release fence per workgroup writes back the XCD's L2 and serialises at ~1.2 us per
        // workgroup and XCD: 1 ms per step at 5 860 workgroups, measured.)
        if (tid == 0) {
            __hip_atomic_store(&blk_min[u * nb + blockIdx.x], red_v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&blk_arg[u * nb + blockIdx.x], red_i[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Arrival tree: 256 -> 16 -> 1 monotonic counters, each on its own 128-byte line (one counter
        // for all workgroups serialises at ~0.1 us per add: 0.6 ms per step at 5 860 workgroups,
        // measured).  A workgroup climbs a level only when its add completes the counter's quota.
        const unsigned int b = blockIdx.x, round = (unsigned int)step + 1u;
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
    // ---- last workgroup: global argmin per utterance (lowest index on exact ties), paths, next table ----
    int64_t winner[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        double best = DBL_MAX;
        int64_t arg = INT64_MAX;
        for (int b = tid; b < (int)nb; b += (int)blockDim.x) {
            const double v = __hip_atomic_load(&blk_min[u * nb + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int64_t i = __hip_atomic_load(&blk_arg[u * nb + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v < best || (v == best && i < arg)) { best = v; arg = i; }
        }
        block_reduce(best, arg);
        winner[u] = red_i[0];
        if (a.shard_out) {
            // this rank's share of the windows only: the winner travels (greedy_shard_pick_kernel decides and writes the table)
            if (tid == 0 && u == 0) { a.shard_out[0] = red_v[0]; reinterpret_cast<int64_t *>(a.shard_out)[1] = red_i[0]; }
        } else if (tid == 0 && u < a.nu && step < a.nsteps_u[u]) {
            path[a.out_off[u] + step] = red_i[0];
            if (dist) dist[a.out_off[u] + step] = __dsqrt_rn(red_v[0]);
        }
        __syncthreads();
    }
    if (!a.shard_out && step + 1 < nsteps) greedy_write_table<UB>(a, step + 1, winner, true, tab_next, tid, (int)blockDim.x);
}

// One step of the greedy search = ONE launch of a persistent grid: one workgroup of up to 8 wavefronts
// per compute unit; every wavefront owns 64-window tiles (tile t -> wavefront t mod #wavefronts) and
// never waits for another.  Every lane owns one window of the tile and accumulates its squared
// distance in the canonical column order, weights and references coming from the step's table.
//
//   Join columns: 16-byte loads from the lane-major tiles (greedy_tile_kernel) straight into the
//   registers the arithmetic reads -- 1 KB of consecutive addresses per wavefront and request, no
//   transposition through LDS.
//   Target columns, lds_mode 1 (the default): the 64 + me - 1 target rows a tile's windows cover are
//   loaded ONCE (lane = row; the me - 1 extra rows by the first lanes only) and written to the
//   wavefront's private LDS block [row][column]; every epoch of every window reads its row from there
//   (16-byte LDS reads, conflict-free at a pitch of 4 mod 64 floats).
//   Target columns, lds_mode 0 (rows too wide for LDS): every epoch re-reads its rows from the
//   tiles; those re-reads hit L2, whose path into a compute unit carries ~70 GB/s -- measured
//   328 us per step at magphase-60 widths with the arithmetic removed, against 160 us of HBM time.
//   One request ring per wavefront carries the chunks (32 columns) of successive tiles, GR_NSTG - 1
//   chunks ahead of the arithmetic; it never drains inside a step.
//   The step's (weight, reference) table is copied to LDS once per workgroup (the only barrier of
//   the scan) and read from there with 16-byte broadcast reads, four columns ahead of the
//   arithmetic.  (Scalar loads of the table cost ~150 ns each here and can only be waited for with
//   lgkmcnt(0), so a request cannot stay in flight across a wait: the scan ran at the scalar-load
//   latency.  LDS reads return in order: counted waits.)
//
//   The workgroup's (min, argmin) goes to global memory; the workgroup that arrives LAST (sc1
//   stores / arrival tree / sc1 loads) reduces all partial results, appends the winner to the path
//   and writes the next step's table (its `current_join_rep` row is the next `prev`).
//
// Software pipelining: every load is an ordinary load, so the compiler places the counted waits
// itself and copies or spills a register only after its load has landed (loads written as inline
// asm look complete to the register allocator the moment they issue: a copy between such a load
// and its wait reads garbage -- seen here as results that changed from run to run).  What keeps a
// request ahead of its use is the empty `asm volatile("" ::: "memory")` behind it: a load cannot
// be sunk across a statement that may write memory.  The "+v" operands of the other empty
// statements tie the arithmetic to the program order (without them the scheduler moves the
// arithmetic of one batch in front of the requests for the next).
#define GR_W 64            // windows per wavefront tile
#define GR_MAXW 8          // wavefronts per workgroup
typedef double f64x2 __attribute__((ext_vector_type(2)));

// UB: utterances per scan (1, or up to GR_MAXU).  Every column's weighted database value fl64(x * w) is
// computed once and compared with the reference of each utterance: 2 + 3 UB float64 operations per
// column and window instead of 5 UB, and ONE pass over the database per step for all of them.
template <bool IN_LDS, int UB>
__global__ void __launch_bounds__(GR_W * GR_MAXW)
greedy_step_kernel(GreedyArgs a, int64_t step, int64_t nsteps, const double *__restrict__ tab,
                   double *__restrict__ tab_next, double *__restrict__ blk_min, int64_t *__restrict__ blk_arg,
                   unsigned int *__restrict__ arrive, int64_t *__restrict__ path, double *__restrict__ dist)
{
    extern __shared__ __align__(16) char lds[];          // table | one target block per wavefront (lds_mode 1)
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform: scalar address arithmetic
    constexpr bool in_lds = IN_LDS;                      // == (a.lds_mode != 0)
    const int jch = greedy_join_chunks(a), tch = greedy_target_chunks(a);
    const int nT = a.nep * tch, n_chunks = jch + nT, JQ = jch * 8, FQ = tch * 8;
    const int pitch = tch * GR_CC + 4;
    const int nB = a.me > 1 ? tch : 0;                   // chunks of the me - 1 extra rows
    const int ring_per_tile = in_lds ? tch + nB + jch : n_chunks;
    constexpr int TS = (UB == 1) ? 2 : 4;                // table doubles per column
    // ring stages: with several utterances a chunk carries 2-3 times the arithmetic, one chunk ahead is as
    // much time as two were, and the registers go to the accumulators instead
    constexpr int NSTG = (UB == 1) ? GR_NSTG : 2;
    const int table_bytes = n_chunks * GR_CC * TS * 8;
    float *const Fs = reinterpret_cast<float *>(lds + table_bytes) + (size_t)wave * (GR_W + a.me - 1) * pitch;
    const int ntiles = a.tile_n > 0 ? (int)a.tile_n : (int)((a.Nwin + GR_W - 1) / GR_W);       // (of this launch's range)
    const int tile0 = (int)a.tile_lo;
    const int wave_id = blockIdx.x * nwaves + wave, wave_stride = gridDim.x * nwaves;
    const int my_tiles = wave_id < ntiles ? (ntiles - 1 - wave_id) / wave_stride + 1 : 0;
    const int total = my_tiles * ring_per_tile;          // ring chunks this wavefront consumes

    // request side: (tile, position in the tile's ring sequence).  A wavefront's 64 rows of one float4
    // column group are 1 KB of consecutive addresses: scalar base + a per-lane byte offset (the tile
    // arrays are padded with zero-filled tiles, so no row index needs a clamp).
    //   lds_mode 1: own target rows, extra rows, join chunks;  lds_mode 0: the table's chunk order
    int f_tile = wave_id, f_pos = 0;
    const unsigned off_own = (unsigned)lane * 16u;
    const unsigned off_extra = (unsigned)(lane < a.me - 1 ? lane : (a.me > 1 ? a.me - 2 : 0)) * 16u;
    const unsigned jl = (unsigned)lane + (unsigned)a.prev_row0;
    const unsigned off_join = (jl >> 6) * ((unsigned)JQ << 10) + (jl & 63u) * 16u;
    const char *const FTb = reinterpret_cast<const char *>(a.FT), *const JTb = reinterpret_cast<const char *>(a.JT);
    f32x4 stage[NSTG][8];
    auto fetch = [&](f32x4 (&st)[8], int pin0) {
        const int t = tile0 + (f_tile < ntiles ? f_tile : ntiles - 1);      // surplus request: re-read, never consumed
        const char *base;
        unsigned voff = off_own;
        if (in_lds) {
            if (f_pos < tch) base = FTb + (((size_t)t * FQ + f_pos * 8) << 10);
            else if (f_pos < tch + nB) {
                // the me - 1 extra rows = first rows of the next tile; the other lanes repeat the last
                // of them (same cache line; every lane takes part, no exec mask around the loads)
                base = FTb + (((size_t)(t + 1) * FQ + (f_pos - tch) * 8) << 10);
                voff = off_extra;
            } else { base = JTb + (((size_t)t * JQ + (f_pos - tch - nB) * 8) << 10); voff = off_join; }
        } else if (f_pos < jch) {
            base = JTb + (((size_t)t * JQ + f_pos * 8) << 10); voff = off_join;
        } else {
            const int k = (f_pos - jch) / tch, cc = (f_pos - jch) - k * tch;
            const unsigned fl = (unsigned)lane + (unsigned)a.ep[k];       // row of epoch k: may reach into the next tile
            base = FTb + (((size_t)t * FQ + cc * 8) << 10);
            voff = (fl >> 6) * ((unsigned)FQ << 10) + (fl & 63u) * 16u;
        }
        voff += (unsigned)pin0;
        // nt: bytes this launch reads once (-5 % per step); the target re-reads of lds_mode 0 are not
        if (in_lds || f_pos < jch) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                st[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + voff + 1024 * j));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) st[j] = *reinterpret_cast<const f32x4 *>(base + voff + 1024 * j);
        }
        asm volatile("" ::: "memory");                  // the requests stay here (see above)
        if (++f_pos == ring_per_tile) { f_pos = 0; f_tile += wave_stride; }
    };
    // the first requests go out before the table copy, and land behind it
#pragma unroll
    for (int s = 0; s < NSTG - 1; ++s) fetch(stage[s], 0);

    // the step's table -> LDS
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(tab);
        f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
        for (int e = tid; e < table_bytes / 16; e += blockDim.x) dst[e] = src[e];
    }
    __syncthreads();

    // consume side: the table is read strictly in chunk order (greedy_chunk_slot), wrapping at the end
    // of a window.  tq[b & 1][i] = (w, ref) of column i of batch b (four columns = 64 bytes, the same
    // address in every lane: a broadcast read).
    double best[UB], acc_j[UB], acc_t[UB];
    int64_t arg[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) { best[u] = DBL_MAX; arg[u] = INT64_MAX; acc_j[u] = 0.0; acc_t[u] = 0.0; }
    int c_tile = wave_id, c_pos = 0, t_done = 0, slot = 0;
    // tq[c % RING]: the table entry of column c of the stream -- (w, ref) for one utterance, (w, ref0) and
    // (ref1, ref2) for up to three: a sliding window of RING columns, the read for column c + RING - 1
    // issued before column c is computed (its slot held column c - 1)
    constexpr int RING = (UB == 1) ? 8 : 4;
    constexpr int TQ = TS / 2;                            // 16-byte pieces per column
    f64x2 tq[RING][TQ];
    int pin = 0;                                          // always 0; orders the refill behind the arithmetic
    const f64x2 *const table = reinterpret_cast<const f64x2 *>(lds);
#pragma unroll
    for (int c = 0; c < RING - 1; ++c)
#pragma unroll
        for (int h = 0; h < TQ; ++h) tq[c][h] = table[c * TQ + h];
    asm volatile("" ::: "memory");
    // one chunk of arithmetic: x[g] = columns 4g .. 4g+3 of the chunk, table columns slot*32 .. slot*32+31
    auto chunk = [&](f32x4 (&x)[8], double (&acc)[UB]) {
        const f64x2 *const cur = table + slot * GR_CC * TQ;
        if (++slot == n_chunks) slot = 0;
        const f64x2 *const nxt = table + slot * GR_CC * TQ;
#pragma unroll
        for (int c = 0; c < GR_CC; ++c) {
#pragma unroll
            for (int h = 0; h < TQ; ++h)
                tq[(c + RING - 1) % RING][h] = c + RING - 1 < GR_CC ? cur[(c + RING - 1) * TQ + h]
                                                                   : nxt[(c + RING - 1 - GR_CC) * TQ + h];
            asm volatile("" ::: "memory");              // the request stays here
            float xv = x[c >> 2][c & 3];
            // column c: behind the request for column c + RING - 1; with two utterances also behind column
            // c - 1 (left alone the scheduler interleaves a whole chunk, runs out of registers and spills:
            // 2-3x slower; serialising only every second column was 6 % slower than every column)
            if (UB == 1) asm volatile("" : "+v"(xv));
            else asm volatile("" : "+v"(xv), "+v"(acc[0]), "+v"(acc[UB - 1]));
            const double xw = __dmul_rn((double)xv, tq[c % RING][0].x);
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const double ref = (u == 0) ? tq[c % RING][0].y : (u == 1 ? tq[c % RING][TQ - 1].x : tq[c % RING][TQ - 1].y);
                const double d = __dsub_rn(xw, ref);
                acc[u] = __dadd_rn(acc[u], __dmul_rn(d, d));      // padded columns: w = ref = 0 adds +0.0
            }
        }
    };
    auto end_of_window = [&]() {
        const int64_t i = (int64_t)(tile0 + c_tile) * GR_W + lane;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const double d = __dadd_rn(acc_j[u], acc_t[u]);
            if (i < a.Nwin && d < best[u]) { best[u] = d; arg[u] = i; }       // windows visited in increasing order
            acc_j[u] = 0.0; acc_t[u] = 0.0;
        }
        c_pos = 0; t_done = 0; c_tile += wave_stride;
    };
    auto handle = [&](f32x4 (&st)[8]) {
        if (!in_lds) {
            // chunk order of the table = ring order: join chunks, then the target chunks of each epoch
            if (c_pos < jch) chunk(st, acc_j); else chunk(st, acc_t);
            asm volatile("" : "+v"(acc_j[0]), "+v"(acc_t[0]), "+v"(pin));
        } else if (c_pos < tch + nB) {
            const bool extra = c_pos >= tch;
            const int col = (extra ? c_pos - tch : c_pos) * GR_CC;
            if (!extra || lane < a.me - 1) {
                float *dst = Fs + (size_t)(extra ? lane + GR_W : lane) * pitch + col;
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4 *>(dst + 4 * j) = st[j];
            }
        } else {
            // join chunk j, then its share of the target chunks (interleaved order, greedy_chunk_slot)
            const int j = c_pos - tch - nB;
            chunk(st, acc_j);
            asm volatile("" : "+v"(acc_j[0]), "+v"(pin));
            const int t_goal = ((j + 1) * nT) / jch;
            for (; t_done < t_goal; ++t_done) {
                const int k = t_done / tch, cc = t_done - k * tch;
                const float *row = Fs + (size_t)(lane + a.ep[k]) * pitch + cc * GR_CC;
                f32x4 x[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = *reinterpret_cast<const f32x4 *>(row + 4 * q);
                chunk(x, acc_t);
            }
        }
        if (++c_pos == ring_per_tile) end_of_window();
    };
    // Stage s holds ring chunk g with g % GR_NSTG == s.  Before chunk g is consumed, chunk g + GR_NSTG - 1
    // is requested into the stage that chunk g - 1 has just left.
    for (int g0 = 0; g0 < total; g0 += NSTG) {
#pragma unroll
        for (int s = 0; s < NSTG; ++s) {
            fetch(stage[(s + NSTG - 1) % NSTG], pin);
            asm volatile("" : "+v"(stage[s][0]), "+v"(pin));          // consume behind the request
            if (g0 + s < total) handle(stage[s]);       // uniform
        }
    }
    __syncthreads();                                      // the reduction arrays alias the table and the target blocks
    double *red_v = reinterpret_cast<double *>(lds);
    int64_t *red_i = reinterpret_cast<int64_t *>(lds + sizeof(double) * GR_W * GR_MAXW);
    greedy_finish_step<UB>(a, step, nsteps, best, arg, tab_next, blk_min, blk_arg, arrive, path, dist, red_v, red_i, &is_last);
}

template <int UB>
__global__ void greedy_init_kernel(GreedyArgs a, int64_t s0, int64_t s1, int64_t s2, double *__restrict__ tab, unsigned int *arrive)
{
    for (int i = threadIdx.x; i < 32 * (GR_S1 + GR_S2 + 1); i += blockDim.x) arrive[i] = 0;
    const int64_t all[3] = {s0, s1, s2};
    int64_t start[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) start[u] = all[u];
    greedy_write_table<UB>(a, 0, start, false, tab, threadIdx.x, blockDim.x);
}

static size_t greedy_join_tile_elems(const GreedyLayout &g);
static void fill_args(GreedyArgs &a, const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt,
                      const float *JC_unw, int Jp, int Dj, const double *wj, const double *Q, bool greedy_mode,
                      const float *tiles = nullptr)
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
    a.JT = reinterpret_cast<const f32x4 *>(tiles);
    a.FT = tiles ? a.JT + greedy_join_tile_elems(g) : nullptr;
}

void greedy_fill_args(GreedyArgs &a, const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt,
                      const float *JC_unw, int Jp, int Dj, const double *wj, const double *Q, bool greedy_mode,
                      const float *tiles)
{
    fill_args(a, g, F_unw, Fp, Dt, wt, JC_unw, Jp, Dj, wj, Q, greedy_mode, tiles);
}

size_t greedy_counter_bytes() { return (size_t)32 * (GR_S1 + GR_S2 + 1) * sizeof(unsigned int); }

// doubles of one (weight, reference) table; the caller provides two (steps alternate)
// ub: utterances per scan (1: two doubles per column; 2, 3: four)
size_t greedy_table_doubles(const GreedyLayout &g, int Dt, int ub)
{
    const int nep = (g.last_frame_as_target && g.me > 1) ? 2 : g.me;
    const int jch = (g.jdim + GR_CC - 1) / GR_CC;
    return (size_t)(jch + nep * ((Dt + GR_CC - 1) / GR_CC)) * GR_CC * (ub <= 1 ? 2 : 4);
}

// float4 elements of the two tile arrays (join columns first, target columns behind them)
static size_t greedy_join_tile_elems(const GreedyLayout &g)
{
    // join_contexts has Nwin + me rows; zero-filled tiles behind them so that every 64-row tile a
    // wavefront may ask for (window tiles rounded up, shifted by prev_row0) exists
    const int64_t tiles = (g.Nwin + 63) / 64 + (g.prev_row0 + 63) / 64 + 1;
    return (size_t)tiles * ((g.jdim + GR_CC - 1) / GR_CC * 8) * 64;
}
static size_t greedy_target_tile_elems(const GreedyLayout &g, int Dt)
{
    const int64_t tiles = (g.Nwin + 63) / 64 + 1;         // + the tile of the last windows' extra rows
    return (size_t)tiles * ((Dt + GR_CC - 1) / GR_CC * 8) * 64;
}
size_t greedy_tile_bytes(const GreedyLayout &g, int Dt)
{
    return (greedy_join_tile_elems(g) + greedy_target_tile_elems(g, Dt)) * sizeof(f32x4);
}

// Build the lane-major tiles from the row-major unweighted matrices (once per database + layout).
// float16 copy of the join tiles for the float32 scan with the hoisted target term (greedy32_kernels.hip, F16 instance):
// tile[row / 64][q][row % 64] = columns 8q .. 8q+7 as eight halves (round to nearest), 8 ceil(jdim / 64) q's per tile.
// *max_abs_bits: bit pattern of the largest |value| seen (the caller refuses the copy if it leaves the float16 range).
typedef _Float16 gk_h16x8 __attribute__((ext_vector_type(8)));
__global__ void greedy_tile16_kernel(const float *__restrict__ src, int pitch, int64_t nrows, int col0, int ncols,
                                     int Q, int64_t n_elems, gk_h16x8 *__restrict__ dst, unsigned int *__restrict__ max_abs_bits)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_elems) return;
    const int lane = (int)(idx & 63);
    const int64_t g = idx >> 6;
    const int q = (int)(g % Q);
    const int64_t r = (g / Q) * 64 + lane;
    gk_h16x8 v;
    float mx = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = 8 * q + e;
        const float x = (r < nrows && col < ncols) ? src[r * pitch + col0 + col] : 0.f;
        mx = fmaxf(mx, fabsf(x));
        v[e] = (_Float16)x;
    }
    dst[idx] = v;
    if (mx > 0.f) atomicMax(max_abs_bits, __float_as_uint(mx));
}
size_t greedy_tile16_bytes(const GreedyLayout &g)
{
    const int64_t tiles = (g.Nwin + 63) / 64 + (g.prev_row0 + 63) / 64 + 1;
    return (size_t)tiles * ((g.jdim + 63) / 64 * 8) * 64 * 16;
}
void launch_greedy_tiles16(const GreedyLayout &g, const float *JC_unw, int Jp, void *tiles16, unsigned int *max_abs_bits, hipStream_t s)
{
    const int Q = (g.jdim + 63) / 64 * 8;
    const int64_t n = (int64_t)(greedy_tile16_bytes(g) / 16);
    (void)hipMemsetAsync(max_abs_bits, 0, sizeof(unsigned int), s);
    hipLaunchKernelGGL(greedy_tile16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, JC_unw, Jp, g.Nwin + g.me, g.prev_col0, g.jdim,
                       Q, n, reinterpret_cast<gk_h16x8 *>(tiles16), max_abs_bits);
}

// out[0] = max over the windows' join rows of ||w o S'[i]||^2 (bit pattern of the float64), out[1] = ||w||^2 over the scan
// columns: what the float16 scan's bound needs, once per set of weights.  One thread per row of the lane-major float32 tiles.
__global__ void greedy_join_norms_kernel(const f32x4 *__restrict__ JT, int JQ, int64_t row0, int64_t nrows, const double *__restrict__ wj,
                                         int col0, int jdim, unsigned long long *__restrict__ out)
{
    __shared__ double red[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (i < nrows) {
        const int64_t r = row0 + i;
        const f32x4 *t = JT + ((r >> 6) * JQ << 6) + (r & 63);
        for (int q = 0; q * 4 < jdim; ++q) {
            const f32x4 x = t[(int64_t)q << 6];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < jdim) { const double y = (double)x[e] * wj[col0 + 4 * q + e]; v += y * y; }
        }
    }
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(out, (unsigned long long)__double_as_longlong(red[0]));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double w2 = 0.0;
        for (int c = 0; c < jdim; ++c) w2 += wj[col0 + c] * wj[col0 + c];
        out[1] = (unsigned long long)__double_as_longlong(w2);
    }
}
void launch_greedy_join_norms(const GreedyLayout &g, const float *tiles, const double *wj, unsigned long long *out, hipStream_t s)
{
    (void)hipMemsetAsync(out, 0, 2 * sizeof(unsigned long long), s);
    const int JQ = (g.jdim + GR_CC - 1) / GR_CC * 8;
    hipLaunchKernelGGL(greedy_join_norms_kernel, dim3((unsigned)((g.Nwin + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const f32x4 *>(tiles), JQ,
                       g.prev_row0, g.Nwin, wj, g.prev_col0, g.jdim, out);
}

void launch_greedy_tiles(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const float *JC_unw, int Jp,
                         float *tiles, hipStream_t s)
{
    f32x4 *JT = reinterpret_cast<f32x4 *>(tiles);
    f32x4 *FT = JT + greedy_join_tile_elems(g);
    const int64_t nj = (int64_t)greedy_join_tile_elems(g), nf = (int64_t)greedy_target_tile_elems(g, Dt);
    hipLaunchKernelGGL(greedy_tile_kernel, dim3((unsigned)((nj + 255) / 256)), dim3(256), 0, s, JC_unw, Jp,
                       g.Nwin + g.me, g.prev_col0, g.jdim, (g.jdim + GR_CC - 1) / GR_CC * 8, nj, JT);
    hipLaunchKernelGGL(greedy_tile_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, F_unw, Fp,
                       g.Nwin + g.me - 1, 0, Dt, (Dt + GR_CC - 1) / GR_CC * 8, nf, FT);
}

// Wavefronts per workgroup and LDS bytes of greedy_step_kernel.  One workgroup per compute unit.
// lds_mode 1 while at least 4 wavefronts' target blocks fit the 160 KB beside the table (8 at
// magphase-60 widths); never more wavefronts than it takes to give every compute unit work.
#define GR_LDS_BUDGET (160 * 1024)
static size_t greedy_lds_table_bytes(const GreedyLayout &g, int Dt, int ub)
{
    return greedy_table_doubles(g, Dt, ub) * sizeof(double);
}
static size_t greedy_lds_wave_bytes(const GreedyLayout &g, int Dt)
{
    return (size_t)(GR_W + g.me - 1) * ((Dt + GR_CC - 1) / GR_CC * GR_CC + 4) * sizeof(float);
}
static int greedy_lds_max_waves(const GreedyLayout &g, int Dt, int ub)
{
    const size_t fixed = greedy_lds_table_bytes(g, Dt, ub) + 64;
    if (fixed >= GR_LDS_BUDGET) return 0;
    const size_t w = (GR_LDS_BUDGET - fixed) / greedy_lds_wave_bytes(g, Dt);
    return (int)(w > GR_MAXW ? GR_MAXW : w);
}
// the layout decision does not depend on the number of utterances per scan (judged for one)
bool greedy_lds_mode(const GreedyLayout &g, int Dt) { return greedy_lds_max_waves(g, Dt, 1) >= 4; }
// false: the table alone does not fit the LDS (tens of thousands of scan columns)
bool greedy_supported(const GreedyLayout &g, int Dt)
{
    return greedy_lds_table_bytes(g, Dt, 1) + (size_t)GR_W * GR_MAXW * 16 + 64 <= GR_LDS_BUDGET;
}
// utterances per scan the LDS allows: three when the wider table still leaves four wavefronts their blocks
int greedy_max_utts(const GreedyLayout &g, int Dt)
{
    if (greedy_lds_table_bytes(g, Dt, GR_MAXU) + (size_t)GR_W * GR_MAXW * 16 + 64 > GR_LDS_BUDGET) return 1;
    if (greedy_lds_mode(g, Dt) && greedy_lds_max_waves(g, Dt, GR_MAXU) < 4) return 1;
    return GR_MAXU;
}
static int greedy_waves(const GreedyLayout &g, int Dt, int n_cus, int ub)
{
    const int64_t ntiles = (g.Nwin + GR_W - 1) / GR_W;
    int64_t w = (ntiles + n_cus - 1) / n_cus;
    const int wmax = greedy_lds_mode(g, Dt) ? greedy_lds_max_waves(g, Dt, ub) : GR_MAXW;
    if (w > wmax) w = wmax;
    return (int)(w < 1 ? 1 : w);
}
static size_t greedy_lds_bytes(const GreedyLayout &g, int Dt, int waves, int ub)
{
    size_t b = greedy_lds_table_bytes(g, Dt, ub) + (greedy_lds_mode(g, Dt) ? (size_t)waves * greedy_lds_wave_bytes(g, Dt) : 0);
    const size_t red = (size_t)GR_W * GR_MAXW * 16;      // reduction arrays of the step's tail (aliased)
    return b < red ? red : b;
}

// nu utterances (1 .. GR_MAXU) share every scan: q_off[u] = first row of utterance u in Q, nsteps_u[u] its
// steps, out_off[u] its first slot in path / dist, start[u] its start state.  blk_min / blk_arg hold
// max(nu, 1) x nblk entries; tables holds two tables of greedy_table_doubles(g, Dt, nu) doubles.
void launch_greedy_batch(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt,
                         const float *JC_unw, int Jp, int Dj, const double *wj, const float *tiles, const double *Q,
                         int nu, const int64_t *q_off, const int64_t *nsteps_u, const int64_t *out_off,
                         const int64_t *start, double *tables, double *blk_min,
                         int64_t *blk_arg, int nblk, int n_cus, unsigned int *arrive, int64_t *path, double *dist, hipStream_t s)
{
    GreedyArgs a{};
    fill_args(a, g, F_unw, Fp, Dt, wt, JC_unw, Jp, Dj, wj, Q, true, tiles);
    a.lds_mode = greedy_lds_mode(g, Dt) ? 1 : 0;
    a.nu = nu;
    int64_t nsteps = 0, st3[3] = {-1, -1, -1};
    for (int u = 0; u < 3; ++u) {
        a.q_off[u] = u < nu ? q_off[u] : 0;
        a.nsteps_u[u] = u < nu ? nsteps_u[u] : 0;
        a.out_off[u] = u < nu ? out_off[u] : 0;
        if (u < nu) { st3[u] = start[u]; if (nsteps_u[u] > nsteps) nsteps = nsteps_u[u]; }
    }
    const int ub = nu <= 1 ? 1 : GR_MAXU;
    double *tab[2] = {tables, tables + greedy_table_doubles(g, Dt, ub)};
    const int waves = greedy_waves(g, Dt, n_cus, ub);
    const size_t lds = greedy_lds_bytes(g, Dt, waves, ub);
    void (*kernel)(GreedyArgs, int64_t, int64_t, const double *, double *, double *, int64_t *, unsigned int *, int64_t *, double *);
    if (ub == 1) {
        hipLaunchKernelGGL(greedy_init_kernel<1>, dim3(1), dim3(256), 0, s, a, st3[0], st3[1], st3[2], tab[0], arrive);
        kernel = a.lds_mode ? greedy_step_kernel<true, 1> : greedy_step_kernel<false, 1>;
    } else {
        hipLaunchKernelGGL(greedy_init_kernel<GR_MAXU>, dim3(1), dim3(256), 0, s, a, st3[0], st3[1], st3[2], tab[0], arrive);
        kernel = a.lds_mode ? greedy_step_kernel<true, GR_MAXU> : greedy_step_kernel<false, GR_MAXU>;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int64_t st = 0; st < nsteps; ++st)
        hipLaunchKernelGGL(kernel, dim3(nblk), dim3(GR_W * waves), lds, s, a, st, nsteps,
                           tab[st & 1], tab[(st + 1) & 1], blk_min, blk_arg, arrive, path, dist);
}

void launch_greedy(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt,
                   const float *JC_unw, int Jp, int Dj, const double *wj, const float *tiles, const double *Q,
                   int64_t nsteps, int64_t start_state, double *tables, double *blk_min,
                   int64_t *blk_arg, int nblk, int n_cus, unsigned int *arrive, int64_t *path, double *dist, hipStream_t s)
{
    const int64_t zero = 0;
    launch_greedy_batch(g, F_unw, Fp, Dt, wt, JC_unw, Jp, Dj, wj, tiles, Q, 1, &zero, &nsteps, &zero, &start_state,
                        tables, blk_min, blk_arg, nblk, n_cus, arrive, path, dist, s);
}

// ---------------------------------------------------------------------------
// snk_sharded_greedy: the scan of a step split over the ranks (every rank holds the whole database; a rank scans the windows
// of its tiles [tile_lo, tile_lo + tile_n)), one all-gather of 16 bytes per rank and step, then every rank picks the same
// winner -- smallest distance, lowest window on exact ties: what one scan over all windows returns -- and writes the next table.
// ---------------------------------------------------------------------------
__global__ void greedy_shard_pick_kernel(GreedyArgs a, int64_t step, int64_t nsteps, const double *__restrict__ gathered, int G,
                                         double *__restrict__ tab_next, int64_t *__restrict__ path, double *__restrict__ dist)
{
    __shared__ int64_t win_s;
    if (threadIdx.x == 0) {
        double best = DBL_MAX;
        int64_t arg = INT64_MAX;
        for (int r = 0; r < G; ++r) {
            const double v = gathered[2 * r];
            const int64_t i = reinterpret_cast<const int64_t *>(gathered)[2 * r + 1];
            if (v < best || (v == best && i < arg)) { best = v; arg = i; }
        }
        win_s = arg;
        path[step] = arg;
        if (dist) dist[step] = __dsqrt_rn(best);
    }
    __syncthreads();
    const int64_t winner[1] = {win_s};
    if (step + 1 < nsteps) greedy_write_table<1>(a, step + 1, winner, true, tab_next, threadIdx.x, blockDim.x);
}

int greedy_shard_blocks(const GreedyLayout &g, int Dt, int n_cus, int64_t tile_n)
{
    const int waves = greedy_waves(g, Dt, n_cus, 1);
    const int64_t need = (tile_n + waves - 1) / waves;
    return (int)(need < 1 ? 1 : need < n_cus ? need : n_cus);
}

// table of step 0 + counters (once per utterance)
void launch_greedy_shard_init(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt, const float *JC_unw, int Jp,
                              int Dj, const double *wj, const float *tiles, const double *Q, int64_t nsteps, int64_t start_state,
                              double *tables, unsigned int *arrive, hipStream_t s)
{
    GreedyArgs a{};
    fill_args(a, g, F_unw, Fp, Dt, wt, JC_unw, Jp, Dj, wj, Q, true, tiles);
    a.lds_mode = greedy_lds_mode(g, Dt) ? 1 : 0;
    a.nu = 1; a.nsteps_u[0] = nsteps;
    hipLaunchKernelGGL(greedy_init_kernel<1>, dim3(1), dim3(256), 0, s, a, start_state, (int64_t)-1, (int64_t)-1, tables, arrive);
}

// one step of this rank's share: the local winner -> shard_out (16 bytes)
void launch_greedy_shard_step(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt, const float *JC_unw, int Jp,
                              int Dj, const double *wj, const float *tiles, const double *Q, int64_t step, int64_t nsteps,
                              int64_t tile_lo, int64_t tile_n, double *tables, double *blk_min, int64_t *blk_arg, int nblk, int n_cus,
                              unsigned int *arrive, double *shard_out, hipStream_t s)
{
    GreedyArgs a{};
    fill_args(a, g, F_unw, Fp, Dt, wt, JC_unw, Jp, Dj, wj, Q, true, tiles);
    a.lds_mode = greedy_lds_mode(g, Dt) ? 1 : 0;
    a.nu = 1; a.nsteps_u[0] = nsteps;
    a.tile_lo = tile_lo; a.tile_n = tile_n; a.shard_out = shard_out;
    double *tab[2] = {tables, tables + greedy_table_doubles(g, Dt, 1)};
    const int waves = greedy_waves(g, Dt, n_cus, 1);
    const size_t lds = greedy_lds_bytes(g, Dt, waves, 1);
    void (*kernel)(GreedyArgs, int64_t, int64_t, const double *, double *, double *, int64_t *, unsigned int *, int64_t *, double *) =
        a.lds_mode ? greedy_step_kernel<true, 1> : greedy_step_kernel<false, 1>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3(nblk), dim3(GR_W * waves), lds, s, a, step, nsteps, tab[step & 1], tab[(step + 1) & 1], blk_min,
                       blk_arg, arrive, (int64_t *)nullptr, (double *)nullptr);
}

// every rank's winner is there: path, distance, the next step's table
void launch_greedy_shard_pick(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt, const float *JC_unw, int Jp,
                              int Dj, const double *wj, const double *Q, int64_t step, int64_t nsteps, const double *gathered, int G,
                              double *tables, int64_t *path, double *dist, hipStream_t s)
{
    GreedyArgs a{};
    fill_args(a, g, F_unw, Fp, Dt, wt, JC_unw, Jp, Dj, wj, Q, true, nullptr);
    a.lds_mode = greedy_lds_mode(g, Dt) ? 1 : 0;
    a.nu = 1; a.nsteps_u[0] = nsteps;
    double *tab[2] = {tables, tables + greedy_table_doubles(g, Dt, 1)};
    hipLaunchKernelGGL(greedy_shard_pick_kernel, dim3(1), dim3(256), 0, s, a, step, nsteps, gathered, G, tab[(step + 1) & 1], path, dist);
}

// Persistent grid: one workgroup per compute unit (fewer when there are fewer tiles).
int greedy_blocks(const GreedyLayout &g, int Dt, int n_cus, int ub)
{
    const int64_t ntiles = (g.Nwin + GR_W - 1) / GR_W;
    const int waves = greedy_waves(g, Dt, n_cus, ub <= 1 ? 1 : GR_MAXU);
    const int64_t need = (ntiles + waves - 1) / waves;
    return (int)(need < n_cus ? need : n_cus);
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
