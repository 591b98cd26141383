// Greedy joint search, second form: float32 prefilter scan + exact float64 decision, ONE launch per
// utterance group.
//
// Replaces the same reference code as greedy_kernels.hip (greedy_joint_search, script/synth_simple.py:458-503;
// get_tree_for_greedy_search :190-229).  What changed against greedy_step_kernel:
//
//  * The scan accumulates every window's squared distance in FLOAT32 (x*w - ref and the square as two FMAs
//    per column; the float64 form needs convert + multiply + subtract + multiply + add at a quarter of the
//    rate and co-limited the HBM stream).  A float32 total d~ lies within a PROVEN distance of the canonical
//    float64 total d:
//        |d~ - d| <= E(d) = 6 u V sqrt(d) + (n + 8) u d,   u = 2^-24, n = scan columns, V = ||reference vector||
//    (operand roundings of w, ref and x*w - ref: |diff~ - diff| <= u (2 |ref| + 2 |diff|); squares and the
//    n-term FMA chain; Cauchy-Schwarz over the columns).  Only windows with d~ <= tau = min d~ + E(min) + E(tau)
//    can be the exact nearest neighbour or tie with it.  Usually that is ONE window: it wins without any
//    float64 arithmetic.  Two or more (duplicated speech, near ties) are decided by their canonical float64
//    distances, lowest index on exact ties -- the oracle's rule.  Every lane keeps its two best windows and
//    the value of its third, every workgroup publishes its two best and its third value; if a third value
//    reaches tau the step cannot be decided from what was kept: the launch stops and reports the step, and
//    the caller finishes the utterance with the exact scan (mass duplicates: digital silence).
//  * search_epsilon > 0 (synth_simple.py:488-490: `joint_tree.query(..., eps=...)`, shipped as 10.0 in
//    config/slt_simplified_mini.cfg:92): any window within (1 + eps) of the nearest distance is a valid
//    answer, and the float32 minimum is within 2 E of it -- below 1e-3 relative -- so for eps >= 1e-3 the
//    float32 minimum (lowest index among equal float32 totals) is returned and nothing is re-evaluated.
//  * ONE persistent launch walks all steps of up to three utterances (six with the hoisted target term,
//    greedy_hoist_kernels.hip: the scan then reads the join columns and one float32 per window and utterance).  Between two
//    steps every workgroup publishes its record and gathers everybody's (two trips through the fabric; tagged 8-byte granules,
//    see the step's tail): all of them derive the same minimum, bound and holders, so the usual step is settled everywhere at
//    once and the next (weight, reference) table is built by every workgroup for itself, in LDS, from the winners' join rows;
//    otherwise the workgroup that published the minimum decides -- in float16 scans usually BEFORE the gather, on its own windows,
//    while the slower workgroups still scan (see the step's tail) -- and its path entry (-1 before the launch) is the release.  A
//    generation word carries the end of the launch at an undecidable step.  A launch per step cost more than the scan itself
//    at 65 536 units.
//  * Instances: <target rows in LDS, hoisted target term, utterances per scan, float16 join tiles>.  With the hoisted term
//    the scan is the join stream alone; databases that do not fit the caches are then read from a float16 copy of the
//    join tiles (half the bytes) with the bound widened by the rounding of the tiles (g32_err16) -- about fifteen windows
//    per step inside it at 1.5 M units, the best window's neighbours, decided exactly: a wavefront stores the terms of up
//    to four candidates and their canonical chains of additions run side by side in its lanes.
//  * The wait inside the kernel has a watchdog (3 s without news: the launch ends, the caller finishes on the exact scan).
//
// Data layout, request ring, LDS target blocks and chunk order are those of greedy_kernels.hip.
#include "greedy_common.h"
#include "greedy32_device.h"
#include <stdlib.h>
#include <stdio.h>
#include <vector>

namespace snk {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define G32_W 64
#define G32_MAXW 8
#define G32_HOIST_NSTG 3      // ring stages of the hoisted scan (little arithmetic to hide requests behind: three chunks ahead)
#define G32_TRACE_STEPS 64
#define G32_STALL_TICKS 300000000ull      // 3 s of the 100 MHz clock
#define G32_UB 3              // utterances per scan: (w, ref0, ref1, ref2) = 16 table bytes per column
#define G32_UBX 6             // ... of the hoisted scan's wide instance: (w, ref0 .. ref5, -) = 32 table bytes per column

#define G32_SEND 63            // windows a holder's slot takes (+ the count granule)
#define G32_CAND 768           // windows the deciding workgroup weighs (LDS list)
struct G32Info {              // what the gather found for an utterance (LDS; the same in every workgroup)
    int state;                // 0 not active | 1 settled: winner | 2 nothing finite | 3 the workgroup D decides | 4 the launch was ended
    int D, nH, sender;        // who decides | holders of windows inside the bound | this workgroup sends its lanes' windows
    int64_t winner;
    double tau;
    float mv, pad_;           // the float32 minimum the bound was built from (tripwire of the bound: g32_trip)
};

struct Top3 {                 // two best windows (value, index) and the third value; indices fit 31 bits (snk_upload_db)
    float v1, v2, v3;
    int a1, a2;
};
__device__ __forceinline__ void top3_init(Top3 &t) { t.v1 = t.v2 = t.v3 = __builtin_inff(); t.a1 = t.a2 = INT32_MAX; }
__device__ __forceinline__ void top3_push(Top3 &t, float v, int i)
{
    if (lt_vi(v, i, t.v1, t.a1)) { t.v3 = t.v2; t.v2 = t.v1; t.a2 = t.a1; t.v1 = v; t.a1 = i; }
    else if (lt_vi(v, i, t.v2, t.a2)) { t.v3 = t.v2; t.v2 = v; t.a2 = i; }
    else if (v < t.v3) t.v3 = v;
}
__device__ __forceinline__ void top3_merge(Top3 &t, const Top3 &o)
{
    top3_push(t, o.v1, o.a1);
    top3_push(t, o.v2, o.a2);
    if (o.v3 < t.v3) t.v3 = o.v3;
}
__device__ __forceinline__ Top3 top3_shfl_xor(const Top3 &t, int m)
{
    Top3 o;
    o.v1 = __shfl_xor(t.v1, m, 64); o.v2 = __shfl_xor(t.v2, m, 64); o.v3 = __shfl_xor(t.v3, m, 64);
    o.a1 = __shfl_xor(t.a1, m, 64); o.a2 = __shfl_xor(t.a2, m, 64);
    return o;
}

// The float32 table of a step, built by EVERY workgroup for itself in its LDS: per scan column, in chunk order,
// (w, ref0, ref1, ref2) [| ref3, ref4, ref5, -] -- the references are the previous winners' join rows (plain cached loads:
// read-only data) and the step's target rows -- and, per utterance, the squared norm of the reference vector (float64, for
// the error bound) in V2.  (Until the hand-off carried the winners themselves, the deciding workgroup wrote this table to
// global memory and everybody fetched it with sc1 loads: two more fabric round trips per step.)
template <int UB>
__device__ void g32_build_table(const GreedyArgs &a, int64_t step, const int64_t (&prev_row)[UB], bool prev_is_current,
                                u32x4 *dst, double *red, double (&V2)[UB], int tid, int nthreads, int n_join_only)
{
    // n_join_only: entries of a join-only table whose chunks are wider than GR_CC (the float16 scan); entry e = column e
    const int jch = n_join_only ? n_join_only / GR_CC : greedy_join_chunks(a), tch = greedy_target_chunks(a);
    const int nT = a.hoist ? 0 : a.nep * tch, n = (jch + nT) * GR_CC;          // hoisted target term: join columns only
    constexpr int TE = UB <= 3 ? 1 : 2;                          // 16-byte pieces per column
    double n2[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) n2[u] = 0.0;
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
        const u32x4 bits = {__builtin_bit_cast(unsigned int, (float)w), __builtin_bit_cast(unsigned int, (float)ref[0]),
                            __builtin_bit_cast(unsigned int, (float)ref[1 % UB]), __builtin_bit_cast(unsigned int, (float)ref[2 % UB])};
        dst[e * TE] = bits;
        if (UB > 3) {
            const u32x4 hi = {__builtin_bit_cast(unsigned int, (float)ref[3 % UB]), __builtin_bit_cast(unsigned int, (float)ref[4 % UB]),
                              __builtin_bit_cast(unsigned int, (float)ref[5 % UB]), 0u};
            dst[e * TE + 1] = hi;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) n2[u] += ref[u] * ref[u];
    }
    // squared norms of the references: block sums (any order: they only scale an error bound, +1 % is added there);
    // wavefront shuffles, then every thread adds the wavefronts' partial sums
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        double v = n2[u];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) red[wave * UB + u] = v;
    }
    __syncthreads();                                              // (the table is complete behind this barrier, too)
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        double v = 0.0;
        for (int w = 0; w < nwaves; ++w) v += red[w * UB + u];
        V2[u] = g32_uniform_d(v);
    }
}

template <bool IN_LDS, bool HOIST, int UB, bool F16>
__global__ void __launch_bounds__(G32_W * G32_MAXW)
greedy32_kernel(GreedyArgs a, int64_t nsteps, int flags, int use_nt, int lds_bytes,
                unsigned long long *pub, unsigned long long *send, unsigned int *gen,
                int64_t *path, int64_t *status,                                       // shared between workgroups: no restrict
                unsigned long long *trace)
{
    const bool approx = (flags & 1) != 0;                  // search_epsilon mode
    const bool test_stall = (flags & 256) != 0;            // test hook: workgroup 0 never arrives at step 1 (the watchdog's case)
    // cross-check mode (option greedy_fenced): every hand-off additionally bracketed by agent-scope release / acquire
    // fences (buffer_wbl2 sc1 / buffer_inv sc1).  The default hand-off is the form "8-byte agent-scope atomics on both
    // sides, storing wavefronts drained (s_waitcnt vmcnt(0)) before the arrival" -- measured valid on gfx950 for
    // hipMalloc memory (MI355X_MICROARCH.md, inter-workgroup visibility), not a guarantee of the memory model; a test
    // runs 1 000 steps in both modes and compares.
    const bool fenced = (flags & 512) != 0;
    const bool speculate = (flags & 2048) == 0;             // option greedy_speculate: the decision before the gather (float16 scans)
    // optional timeline (SNK_G32_TRACE=file): 8 stamps of the 100 MHz clock per step and workgroup, steps 0 .. G32_TRACE_STEPS - 1
    auto stamp = [&](int64_t st, int k) {
        if (trace && st < G32_TRACE_STEPS && threadIdx.x == 0)
            trace[((size_t)st * gridDim.x + blockIdx.x) * 16 + k] = __builtin_amdgcn_s_memrealtime();
    };
    extern __shared__ __align__(16) char lds[];          // table | one target block per wavefront (lds_mode 1)
    __shared__ int gen_seen, spec_go[G32_UBX], spec_a[G32_UBX];
    __shared__ float spec_v[G32_UBX][2];                   // the workgroup's two best values of the step, per utterance
    __shared__ int64_t next_rows[G32_UBX];                 // the step's winners, as the polling thread saw them
    const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool in_lds = IN_LDS;
    // F16: the join columns are streamed as float16 (8 columns per 16-byte request, 64 per chunk): half the bytes, a
    // wider bound (see g32_err16), the same exact decision
    constexpr int CC = F16 ? 64 : GR_CC;                    // scan columns per chunk of eight 16-byte requests
    const int jch = F16 ? (a.jdim + 63) / 64 : greedy_join_chunks(a), tch = greedy_target_chunks(a);
    const int nT = HOIST ? 0 : a.nep * tch, n_chunks = jch + nT, JQ = jch * 8, FQ = tch * 8;
    const int ncols = n_chunks * CC;
    const int jq_last = (a.jdim + (F16 ? 7 : 3)) / (F16 ? 8 : 4) - (jch - 1) * 8;   // 16-byte columns of the last join chunk that hold data
    const int ecols = ncols + (HOIST ? 2 : 0);            // the bound's chain length: + the addition of the hoisted value
    const int pitch = tch * GR_CC + 4;
    const int nB = a.me > 1 ? tch : 0;
    const int ring_per_tile = in_lds ? tch + nB + jch : n_chunks;
    auto errf = [&](double d, double V2) { return F16 ? g32_err16(d, V2, ecols, a.f16_delta) : g32_err(d, V2, ecols); };
    constexpr int NSTG = HOIST ? G32_HOIST_NSTG : GR_NSTG;                         // ring stages: two chunks (16 KB per wavefront) requested ahead of the arithmetic
    constexpr int TE = UB <= 3 ? 1 : 2;                     // 16-byte pieces of a table entry: (w, ref0, ref1, ref2 | ref3, ref4, ref5, -)
    const int table_bytes = ncols * 16 * TE;
    float *const Fs = reinterpret_cast<float *>(lds + table_bytes) + (size_t)wave * (G32_W + a.me - 1) * pitch;
    const int ntiles = (int)((a.Nwin + G32_W - 1) / G32_W);
    // Tiles are dealt out in rounds of wave_stride: in a full round workgroup b takes the nwaves ADJACENT tiles b nwaves .. b nwaves
    // + nwaves - 1 (the windows inside a step's bound are the best window's neighbours: adjacent tiles in one workgroup mean one
    // holder, and a single holder can decide before the gather, see the step's tail; dealt out tile by tile across the workgroups,
    // two steps of three had several holders), and the tiles of the last, partial round go to ALL workgroups in groups of
    // ceil(rest / workgroups) adjacent ones (given to the first workgroups only -- eight more tiles each on 114 of 256 compute units
    // at 1.5 M units -- the last workgroup finished 10 us behind the median one).
    const int wave_stride = gridDim.x * nwaves;
    const int full_rounds = ntiles / wave_stride, rest_base = full_rounds * wave_stride, rest = ntiles - rest_base;
    const int rest_group = rest > 0 ? (rest + (int)gridDim.x - 1) / (int)gridDim.x : 1;
    const int rest_tile = rest_base + (int)blockIdx.x * rest_group + wave;
    const bool has_rest = wave < rest_group && rest_tile < ntiles;
    auto tile_of = [&](int k) { return k < full_rounds ? k * wave_stride + (int)blockIdx.x * nwaves + wave : (has_rest ? rest_tile : ntiles); };
    const int my_tiles = full_rounds + (has_rest ? 1 : 0);
    const int total = my_tiles * ring_per_tile;
    const unsigned int nb = gridDim.x;
    const unsigned off_own = (unsigned)lane * 16u;
    const unsigned off_extra = (unsigned)(lane < a.me - 1 ? lane : (a.me > 1 ? a.me - 2 : 0)) * 16u;
    const unsigned jl = (unsigned)lane + (unsigned)a.prev_row0;
    const unsigned off_join = (jl >> 6) * ((unsigned)JQ << 10) + (jl & 63u) * 16u;
    const char *const FTb = reinterpret_cast<const char *>(a.FT), *const JTb = reinterpret_cast<const char *>(F16 ? a.JT16 : a.JT);

    // reduction scratch of the step's tail aliases the table and the target blocks
    Top3 *red3 = reinterpret_cast<Top3 *>(lds);
    double *redd = reinterpret_cast<double *>(lds);

    float trip_seen = 0.f;                                 // tripwire of the bound: the largest share of it this lane has reported
    int64_t prev_row[UB];                                  // winners of the previous step (start states first)
#pragma unroll
    for (int u = 0; u < UB; ++u) prev_row[u] = a.start[u];

    // request ring
    int f_k = 0, f_tile = 0, f_pos = 0;
    f32x4 stage[NSTG][8];
    // hoisted target values of the tile in work and of the next one (one float per window and utterance)
    float wcur[UB], wnext[UB];
    const float *wrow[UB];                                 // uniform: W row of the step; the lane adds its window
#pragma unroll
    for (int u = 0; u < UB; ++u) { wcur[u] = 0.f; wnext[u] = 0.f; wrow[u] = nullptr; }
    auto fetch = [&](f32x4 (&st)[8], int pin0) {
        const int t = f_tile < ntiles ? f_tile : ntiles - 1;
        if (HOIST && f_pos == 0) {
            // the first request of a tile is issued before the previous tile's last chunk is summed (jch >= 2)
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                wcur[u] = wnext[u];
                if (wrow[u]) wnext[u] = __builtin_nontemporal_load(wrow[u] + ((unsigned)t * G32_W + (unsigned)lane));
            }
        }
        const char *base;
        unsigned voff = off_own;
        if (in_lds) {
            if (f_pos < tch) base = FTb + (((size_t)t * FQ + f_pos * 8) << 10);
            else if (f_pos < tch + nB) { base = FTb + (((size_t)(t + 1) * FQ + (f_pos - tch) * 8) << 10); voff = off_extra; }
            else { base = JTb + (((size_t)t * JQ + (f_pos - tch - nB) * 8) << 10); voff = off_join; }
        } else if (f_pos < jch) {
            base = JTb + (((size_t)t * JQ + f_pos * 8) << 10); voff = off_join;
        } else {
            const int k = (f_pos - jch) / tch, cc = (f_pos - jch) - k * tch;
            const unsigned fl = (unsigned)lane + (unsigned)a.ep[k];
            base = FTb + (((size_t)t * FQ + cc * 8) << 10);
            voff = (fl >> 6) * ((unsigned)FQ << 10) + (fl & 63u) * 16u;
        }
        voff += (unsigned)pin0;
        // nt for databases that are streamed from HBM every step; small ones stay in L2 / Infinity Cache
        if (HOIST && use_nt == 2) {
            // the last join chunk of a streamed database: float4 columns that are padding altogether (151 columns:
            // 38 of 40) are not asked from HBM a second time -- their requests repeat the chunk's first column
            // (weight 0 turns any finite value into +0).  Eight requests either way: the counted waits of the
            // ring stay static (a conditional number of requests cost 3 us per step at 65 536 units).
            const int lim = f_pos == jch - 1 ? jq_last : 8;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                st[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + voff + (j < lim ? 1024 * j : 0)));
        } else if (use_nt && (in_lds || f_pos < jch)) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                st[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + voff + 1024 * j));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) st[j] = *reinterpret_cast<const f32x4 *>(base + voff + 1024 * j);
        }
        asm volatile("" ::: "memory");
        if (++f_pos == ring_per_tile) { f_pos = 0; f_tile = tile_of(++f_k); }
    };
    // the first NSTG - 1 requests of step st: tile data does not depend on the step's table, only the references do
    auto start_fetch = [&](int64_t st) {
        f_k = 0; f_tile = tile_of(0); f_pos = 0;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            wcur[u] = 0.f; wnext[u] = 0.f;
            wrow[u] = (HOIST && u < a.nu && st < a.nsteps_u[u]) ? a.W[u] + st * a.Wp : nullptr;
        }
#pragma unroll
        for (int s = 0; s < NSTG - 1; ++s) fetch(stage[s], 0);
    };
    for (int64_t step = 0; step < nsteps; ++step) {
        // ---- the step's table: every workgroup builds its own from the previous winners (prev_row) ----
        stamp(step, 0);
        start_fetch(step);
        double qn2s[UB];                                      // ||target vector||^2 of the step: requested beside the join rows
#pragma unroll
        for (int u = 0; u < UB; ++u) qn2s[u] = (HOIST && u < a.nu && step < a.nsteps_u[u]) ? a.qn2[u][step] : 0.0;
        double V2w[UB];                                       // squared norms of the step's reference vectors
        // (the builder's 384 bytes of reduction scratch sit right behind the table: the target blocks that share the
        // place in LDS mode are filled later, by the scan)
        g32_build_table<UB>(a, step, prev_row, step > 0, reinterpret_cast<u32x4 *>(lds), reinterpret_cast<double *>(lds + table_bytes),
                            V2w, tid, (int)blockDim.x, F16 ? ncols : 0);
        double EWw[UB];                                       // bound of the hoisted values of this step: hoist_c (||q|| + ||f||max)^2
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const double r = sqrt(qn2s[u]) + sqrt(a.fwmax2);
            EWw[u] = (HOIST && u < a.nu && step < a.nsteps_u[u]) ? g32_uniform_d(a.hoist_c * r * r) : 0.0;
        }
        stamp(step, 1);

        Top3 best[UB];
        float acc[UB];                                    // ONE float32 total per utterance: the bound holds for any order
#pragma unroll
        for (int u = 0; u < UB; ++u) { top3_init(best[u]); acc[u] = 0.f; }
        int c_k = 0, c_tile = tile_of(0), c_pos = 0, t_done = 0, slot = 0;
        int pin = 0;
        const f32x4 *const table = reinterpret_cast<const f32x4 *>(lds);
        // one chunk of arithmetic: x[g] = columns 4g .. 4g+3 of the chunk.  The table entries (w, ref0, ref1, ref2;
        // the same address in every lane: broadcast reads) come four columns at a time
        auto chunk = [&](f32x4 (&x)[8]) {
            const f32x4 *const cur = table + slot * CC * TE;
            if (++slot == n_chunks) slot = 0;
            f32x4 tc[4], td[4];
            if (F16) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const h16x8 hv = __builtin_bit_cast(h16x8, x[g]);
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            tc[i] = cur[(8 * g + 4 * hh + i) * TE];
                            if (TE == 2) td[i] = cur[(8 * g + 4 * hh + i) * TE + 1];
                        }
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float xv = (float)hv[4 * hh + i];
                            asm volatile("" : "+v"(xv), "+v"(acc[0]), "+v"(acc[UB - 1]));
#pragma unroll
                            for (int u = 0; u < UB; ++u) {
                                const float d = __builtin_fmaf(xv, tc[i][0], u < 3 ? -tc[i][1 + (u % 3)] : -td[i][u % 3]);
                                acc[u] = __builtin_fmaf(d, d, acc[u]);
                            }
                        }
                    }
                }
                return;
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    tc[i] = cur[(4 * g + i) * TE];
                    if (TE == 2) td[i] = cur[(4 * g + i) * TE + 1];
                }
                asm volatile("" ::: "memory");              // four columns at a time (the other wavefront of the SIMD covers the read)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float xv = x[g][i];
                    // column after column (left alone the scheduler interleaves a whole chunk and runs out of registers)
                    asm volatile("" : "+v"(xv), "+v"(acc[0]), "+v"(acc[UB - 1]));
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const float d = __builtin_fmaf(xv, tc[i][0], u < 3 ? -tc[i][1 + (u % 3)] : -td[i][u % 3]);
                        acc[u] = __builtin_fmaf(d, d, acc[u]);    // padded columns: w = ref = 0 adds +0
                    }
                }
            }
        };
        auto end_of_window = [&]() {
            const int i = c_tile * G32_W + lane;
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (i < (int)a.Nwin) top3_push(best[u], HOIST ? acc[u] + wcur[u] : acc[u], i);
                acc[u] = 0.f;
            }
            c_pos = 0; t_done = 0; c_tile = tile_of(++c_k);
        };
        auto handle = [&](f32x4 (&st)[8]) {
            if (!in_lds) {
                chunk(st);
                asm volatile("" : "+v"(acc[0]), "+v"(pin));
            } else if (c_pos < tch + nB) {
                const bool extra = c_pos >= tch;
                const int col = (extra ? c_pos - tch : c_pos) * GR_CC;
                if (!extra || lane < a.me - 1) {
                    float *dst = Fs + (size_t)(extra ? lane + G32_W : lane) * pitch + col;
#pragma unroll
                    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4 *>(dst + 4 * j) = st[j];
                }
            } else {
                // join chunk j (from the ring), then its share of the target chunks (from the wavefront's LDS block):
                // ONE instance of the arithmetic, fed from either source
                const int j = c_pos - tch - nB;
                const int t_goal = ((j + 1) * nT) / jch;
                const int items = 1 + (t_goal - t_done);
#pragma nounroll
                for (int it = 0; it < items; ++it) {
                    f32x4 x[8];
                    if (it == 0) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) x[q] = st[q];
                    } else {
                        const int k = t_done / tch, cc = t_done - k * tch;
                        const float *row = Fs + (size_t)(lane + a.ep[k]) * pitch + cc * GR_CC;
#pragma unroll
                        for (int q = 0; q < 8; ++q) x[q] = *reinterpret_cast<const f32x4 *>(row + 4 * q);
                        ++t_done;
                    }
                    chunk(x);
                    asm volatile("" : "+v"(acc[0]), "+v"(pin));
                }
            }
            if (++c_pos == ring_per_tile) end_of_window();
        };
#pragma nounroll
        for (int g0 = 0; g0 < total; g0 += NSTG) {
#pragma unroll
            for (int s = 0; s < NSTG; ++s) {
                fetch(stage[(s + NSTG - 1) % NSTG], pin);
                asm volatile("" : "+v"(stage[s][0]), "+v"(pin));
                if (g0 + s < total) handle(stage[s]);
            }
        }
        __syncthreads();                                      // the reduction arrays alias the table and the target blocks
        stamp(step, 2);

        // ---- the step's tail: two trips through the fabric.  Every workgroup PUBLISHES its record (three 8-byte granules
        //      per utterance: value | window | tag bit; double buffered by the step's parity, the tag bit flips every second
        //      step: a granule proves its own step), every workgroup GATHERS all records (a wavefront per utterance) and
        //      derives the same facts from them: the float32 minimum, the bound tau, who holds windows inside it.  Usually
        //      that settles the step everywhere at once (one window inside the bound; search_epsilon mode).  Otherwise the
        //      workgroup that published the minimum decides -- from its own lanes' candidates (the float16 scan's usual case:
        //      the best window's neighbours sit in one tile), from the other holders' published windows, and from the lists
        //      of those that hold more than they published (tagged granules in a slot of their own: no ordering needed) -- and
        //      its path entry is the release the others poll.  (Until round 4 the step went through two counter trees, a
        //      deciding workgroup that read the records, a generation word, every lane's offers through global atomics and a
        //      second tree: seven dependent trips and more.)
        //      LDS: [0, 1024) wavefront records | [1024, 1536) what the gather found | [2048, 8192) the decider's candidates |
        //      [8192, 16384) exact decision | term arrays
        const unsigned int tagbit = (unsigned int)((step >> 1) & 1);
        const unsigned int stag = (unsigned int)step + 1u;             // tag of the senders' lists
        unsigned long long *const pub_s = pub + (size_t)(step & 1) * G32_UBX * nb * 4;
        G32Info *const info = reinterpret_cast<G32Info *>(lds + 1024);
        int *const ccount = reinterpret_cast<int *>(lds + 1024 + 384);   // [0] candidates, [1] more than anybody kept
        unsigned long long *const clist = reinterpret_cast<unsigned long long *>(lds + 2048);
        auto pack = [&](float v, int i) {
            return (unsigned long long)__builtin_bit_cast(unsigned int, v) | ((unsigned long long)(((unsigned int)i << 1) | tagbit) << 32);
        };
        auto ld64 = [](unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        auto st64 = [](unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        auto give_up = [&]() {                                        // watchdog: the launch ends, the caller falls back to the exact scan
            __hip_atomic_store(&status[3], (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(status, (int64_t)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(gen, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // tau = the largest solution of tau = M + 2 E(tau), approached from above (E(0) = 0: from below the
        // iteration would stall at M = 0, where the natural path lives).  E is increasing and concave: every
        // iterate stays above the solution, so any number of rounds gives a valid bound; the map contracts by
        // ~1e-5 per round and three rounds leave nothing to gain (eight cost 1 us of float64 square roots)
        auto tau_of = [&](float mv, double V2, double EW) {
            const double M = (double)mv + 2.0 * EW;
            double tau = 2.0 * M + 64.0 * 36.0 * 3.5527136788005009e-15 * V2 + 1e-300 + (F16 ? 64.0 * a.f16_delta * a.f16_delta : 0.0);
            for (int it = 0; it < 3; ++it) tau = M + 2.0 * errf(tau, V2);
            return tau * (1.0 + 1e-6) + 1e-300;
        };
        // the lanes' windows inside the bound -> the candidate list in LDS (cap entries); a third one that reaches tau is
        // beyond what was kept (mass ties): the step is then undecidable here
        auto offer = [&](int u, double tau, int cap) {
            const Top3 &t = best[u];
            if ((double)t.v1 <= tau) { const int p = atomicAdd(ccount, 1); if (p < cap) clist[p] = (unsigned long long)(unsigned int)t.a1; }
            if ((double)t.v2 <= tau) { const int p = atomicAdd(ccount, 1); if (p < cap) clist[p] = (unsigned long long)(unsigned int)t.a2; }
            if ((double)t.v3 <= tau) ccount[1] = 1;
        };
        // exact decision among n candidates (ids through `get`): a wavefront per candidate, canonical float64 totals,
        // lowest index on exact ties
        const int ex_cols = a.jdim + a.nep * a.Dt;
        // term arrays in LDS behind the 16 KB of scratch: kslots per wavefront (up to four candidates whose chains run side
        // by side in the lanes) if every wavefront gets one, else one array for as many wavefronts as fit
        int kslots = (lds_bytes - 16384) / (ex_cols * 8 * nwaves);
        kslots = kslots > 4 ? 4 : kslots;
        int nw_exact = kslots >= 1 ? nwaves : (lds_bytes - 16384) / (ex_cols * 8);
        if (kslots < 1) kslots = 1;
        double *const terms = reinterpret_cast<double *>(lds + 16384) + (size_t)wave * kslots * ex_cols;
        // M / V2 / EW: the float32 minimum, the reference norm and the hoisted values' bound the step's tau was built from -- every
        // total computed here is held against them (g32_trip: the tripwire of the scan's bound)
        auto exact_argmin = [&](int u, int n, auto get, float Mf, double V2t, double EWt) -> int64_t {
            double dbest = DBL_MAX;
            int64_t ibest = INT64_MAX;
            // (the six-utterance instance has no registers to spare: no tree sums, candidate after candidate)
            const bool single_round = UB <= 3 && nw_exact == nwaves && n <= nwaves * kslots;     // every wavefront in the loop exactly once
            // (one round: EVERY wavefront runs the body once, with or without candidates -- it holds workgroup barriers)
            for (int p0 = wave, it = 0; single_round ? it < 1 : (p0 < n && wave < nw_exact); p0 += nw_exact * kslots, ++it) {
                int64_t myid = INT64_MAX;                         // lane k < cnt: the k-th candidate of this round
                int cnt = 0;
                int64_t ids[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) ids[k] = (k < kslots && p0 + k * nw_exact < n) ? get(p0 + k * nw_exact) : -1;   // one round trip
                stamp(step, 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (ids[k] < 0) break;                         // uniform
                    if (lane == k) myid = ids[k];
                    ++cnt;
                }
                if constexpr (UB <= 3) {
                    g32_terms_multi(a, u, step, prev_row[u], step > 0, ids, cnt, terms, (size_t)ex_cols, lane);
                } else {
                    // (the six-utterance instance has no registers to spare: candidate after candidate)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < cnt) g32_exact_d2_wave(a, u, step, prev_row[u], step > 0, ids[k], terms + (size_t)k * ex_cols, lane, true);
                }
                stamp(step, 9);
                // Which candidates need their canonical total at all?  Any order of adding the same float64 terms lands
                // within n 2^-53 of the true sum, the canonical one too: a candidate whose tree sum is beyond
                // (1 + 4 n 2^-53) of the smallest tree sum cannot be the canonical minimum.  With all candidates in LDS
                // at once (one round) the block compares tree sums first; ONE survivor is the winner without any chain,
                // ties and near ties go through the chains below.
                unsigned int surv = 0xfu;                         // bit k: candidate k of this wavefront still matters
                if (UB <= 3 && single_round) {
                    double pmin = DBL_MAX, ps[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        ps[k] = DBL_MAX;
                        if (k < cnt) {                             // uniform
                            double v = 0.0;
                            for (int idx = lane; idx < ex_cols; idx += 64) v += terms[(size_t)k * ex_cols + idx];
#pragma unroll
                            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
                            ps[k] = v;
                            pmin = v < pmin ? v : pmin;
                            if (lane == 0) g32_trip(status, (double)Mf, v, errf(v, V2t) + EWt, trip_seen);
                        }
                    }
                    double *pm = reinterpret_cast<double *>(lds + 8192 + 128);
                    int *sc = reinterpret_cast<int *>(lds + 8192 + 192);
                    int64_t *sid = reinterpret_cast<int64_t *>(lds + 8192 + 224);
                    if (lane == 0) pm[wave] = pmin;
                    __syncthreads();
                    double gmin = pm[0];
                    for (int w = 1; w < nwaves; ++w) gmin = pm[w] < gmin ? pm[w] : gmin;
                    const double thr = gmin * (1.0 + 4.0 * (double)ex_cols * 1.1102230246251565e-16 * 1.01) + 1e-300;
                    surv = 0u;
                    int64_t first = -1;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < cnt && ps[k] <= thr) { surv |= 1u << k; if (first < 0) first = ids[k]; }
                    if (lane == 0) { sc[wave] = __popc(surv); sid[wave] = first; }
                    __syncthreads();
                    int total = 0;
                    int64_t only = -1;
                    for (int w = 0; w < nwaves; ++w) { total += sc[w]; if (sc[w] > 0) only = sid[w]; }
                    __syncthreads();
                    if (total == 1) return only;                    // uniform over the workgroup
                }
                // lane 2k sums the join columns of candidate k, lane 2k + 1 its target columns (two independent chains of
                // the canonical order), then one rounded addition joins them
                double d = DBL_MAX;
                {
                    const int k = lane >> 1;
                    double part = 0.0;
                    if (k < cnt && ((surv >> k) & 1u)) {
                        // (one loop for both kinds of lanes -- pointer and length differ: a branch would run the two chains
                        // one after the other)
                        const double *t = terms + (size_t)k * ex_cols + ((lane & 1) ? a.jdim : 0);
                        part = g32_chain_sum(t, (lane & 1) ? a.nep * a.Dt : a.jdim, 0);
                    }
                    const double other = __shfl_xor(part, 1, 64);
                    const double tot = (lane & 1) ? __dadd_rn(other, part) : __dadd_rn(part, other);      // acc_j + acc_t
                    const double dk = __shfl(tot, (lane & 3) << 1, 64);         // lane k < 4 fetches candidate k's total
                    if (lane < cnt && ((surv >> lane) & 1u)) {
                        d = dk;
                        if (!(UB <= 3 && single_round)) g32_trip(status, (double)Mf, dk, errf(dk, V2t) + EWt, trip_seen);      // (else: done on the tree sums)
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int m = 1; m <= 2; m <<= 1) {                 // kslots <= 4: lanes 0 .. 3
                    const double od = __shfl_xor(d, m, 64); const int64_t oi = __shfl_xor(myid, m, 64);
                    if (od < d || (od == d && oi < myid)) { d = od; myid = oi; }
                }
                d = __shfl(d, 0, 64); myid = __shfl(myid, 0, 64);
                stamp(step, 10);
                if (d < dbest || (d == dbest && myid < ibest)) { dbest = d; ibest = myid; }
            }
            __syncthreads();
            double *rd = reinterpret_cast<double *>(lds + 8192);
            int64_t *ri = reinterpret_cast<int64_t *>(lds + 8192 + 64);
            if (lane == 0) { rd[wave] = dbest; ri[wave] = ibest; }
            __syncthreads();
            double d0 = rd[0];
            int64_t i0 = ri[0];
            for (int w = 1; w < nwaves; ++w) if (rd[w] < d0 || (rd[w] == d0 && ri[w] < i0)) { d0 = rd[w]; i0 = ri[w]; }
            __syncthreads();
            return i0;
        };
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (u >= a.nu) break;                                 // uniform
            Top3 t = best[u];
#pragma unroll
            for (int m = 1; m <= 32; m <<= 1) { const Top3 o = top3_shfl_xor(t, m); top3_merge(t, o); }
            // (recursive doubling: the partner's set is disjoint from the lane's at every level)
            if (lane == 0) red3[u * G32_MAXW + wave] = t;
        }
        if (tid < G32_UBX) info[tid].state = 0;
        __syncthreads();
        // publish: wavefront (u mod nwaves) merges the wavefronts' sets of utterance u -- a lane per wavefront, three levels of
        // exchanges (one thread merging them one after the other was 1.3 us of dependent instructions) -- and its lane 0 stores
        // the record; a test hook keeps workgroup 0 silent at step 1
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (!(u < a.nu && step < a.nsteps_u[u]) || (test_stall && step == 1 && blockIdx.x == 0)) continue;      // uniform
            if (wave != u % nwaves) continue;
            Top3 r;
            top3_init(r);
            if (lane < nwaves) r = red3[u * G32_MAXW + lane];
#pragma unroll
            for (int m = 1; m < G32_MAXW; m <<= 1) { const Top3 o = top3_shfl_xor(r, m); top3_merge(r, o); }
            if (lane == 0) {
                if (fenced) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                unsigned long long *o = pub_s + ((size_t)u * nb + blockIdx.x) * 4;
                st64(o, pack(r.v1, r.a1));
                st64(o + 1, pack(r.v2, r.a2));
                st64(o + 2, pack(r.v3, 0));
                spec_v[u][0] = r.v1; spec_v[u][1] = r.v2; spec_a[u] = r.a1;
            }
        }
        stamp(step, 3);
        // ---- float16 scans decide exactly at every step (the best window's neighbours are inside the bound), and the workgroup that
        //      will decide has usually been waiting for the last one to finish its scan: 8 us for the median workgroup, + the trip
        //      of the gather.  So a workgroup that finds the records incomplete decides on ITS OWN windows meanwhile -- the bound
        //      from its own minimum, its lanes' windows inside it, the exact decision among them.  If the gather then names it the
        //      only holder, its minimum was the minimum, its bound the bound (the same arithmetic on the same numbers: compared
        //      bit for bit) and its decision is the step's: released at once.  Everybody else's speculation is thrown away (18 KB
        //      of cold rows per workgroup and step, 1 % of the scan's bytes).
        bool spec_valid[UB];
        double spec_tau[UB];
        int64_t spec_win[UB];
        int spec_n[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) { spec_valid[u] = false; spec_tau[u] = 0.0; spec_win[u] = 0; spec_n[u] = 0; }
        if constexpr (F16) {
            if (!approx && speculate) {
                // a first look at the records (a wavefront per utterance).  All there: nothing to hide behind.  Somebody has
                // published a smaller minimum: this workgroup will not decide.  Otherwise -- its minimum is the smallest so far --
                // it decides on its own windows now.  (Without the second test every early workgroup speculated: 255 useless
                // decisions per step disturbed the last scans by 8 us and made late speculators miss the release.  With it about
                // ln 256 = 6 workgroups per step do, and one that is overtaken later is released no earlier than it is done:
                // the one that overtakes it publishes later and needs as long.)
                __syncthreads();                                      // spec_v / spec_a of this step
                for (int u = wave; u < UB; u += nwaves) {             // uniform per wavefront
                    if (!(u < a.nu && step < a.nsteps_u[u])) continue;
                    const float myv = spec_v[u][0];
                    const int mya = spec_a[u];
                    bool in = true, smaller = false;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned int b = lane + 64 * q;
                        if (b < nb) {
                            const unsigned long long g0 = ld64(pub_s + ((size_t)u * nb + b) * 4);
                            const bool valid = ((unsigned int)(g0 >> 32) & 1u) == tagbit;
                            in = in && valid;
                            smaller = smaller || (valid && lt_vi(__builtin_bit_cast(float, (unsigned int)g0), (int)((unsigned int)(g0 >> 32) >> 1), myv, mya));
                        }
                    }
                    const bool all_in = __builtin_amdgcn_ballot_w64(!in) == 0ull, overtaken = __builtin_amdgcn_ballot_w64(smaller) != 0ull;
                    if (lane == 0) spec_go[u] = (!all_in && !overtaken) ? 1 : 0;
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    if (!(u < a.nu && step < a.nsteps_u[u]) || !spec_go[u]) continue;      // uniform
                    const float mvl = spec_v[u][0], v2l = spec_v[u][1];
                    if (!(mvl < __builtin_inff())) continue;
                    const double tau = tau_of(mvl, V2w[u], EWw[u]);
                    if (!((double)v2l <= tau)) continue;                    // one window of its own inside the bound: nothing to weigh
                    if (tid == 0) { ccount[0] = 0; ccount[1] = 0; }
                    __syncthreads();
                    offer(u, tau, G32_CAND);
                    __syncthreads();
                    const int n = ccount[0];
                    const bool over = ccount[1] != 0 || n > G32_CAND;
                    __syncthreads();
                    if (over || n < 2) continue;
                    spec_win[u] = exact_argmin(u, n, [&](int p) { return (int64_t)clist[p]; }, mvl, V2w[u], EWw[u]);
                    spec_tau[u] = tau; spec_n[u] = n; spec_valid[u] = true;
                }
                __syncthreads();
            }
        }
        // gather and classify: a wavefront per utterance, four records per lane
        for (int u = wave; u < UB; u += nwaves) {                     // uniform per wavefront
            if (!(u < a.nu && step < a.nsteps_u[u])) continue;
            float rv1[4], rv2[4], rv3[4];
            int ra1[4], ra2[4];
            bool got = false;
            const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned int b = lane + 64 * q;
                    rv1[q] = rv2[q] = rv3[q] = __builtin_inff(); ra1[q] = ra2[q] = INT32_MAX;
                    if (b < nb) {
                        unsigned long long *o = pub_s + ((size_t)u * nb + b) * 4;
                        const unsigned long long g0 = ld64(o), g1 = ld64(o + 1), g2 = ld64(o + 2);
                        ok = ok && ((unsigned int)(g0 >> 32) & 1u) == tagbit && ((unsigned int)(g1 >> 32) & 1u) == tagbit
                                && ((unsigned int)(g2 >> 32) & 1u) == tagbit;
                        rv1[q] = __builtin_bit_cast(float, (unsigned int)g0); ra1[q] = (int)((unsigned int)(g0 >> 32) >> 1);
                        rv2[q] = __builtin_bit_cast(float, (unsigned int)g1); ra2[q] = (int)((unsigned int)(g1 >> 32) >> 1);
                        rv3[q] = __builtin_bit_cast(float, (unsigned int)g2);
                    }
                }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) { got = true; break; }
                if (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0xffffffffu) break;
                // watchdog: a step takes microseconds.  Seconds without news mean that some workgroup of the launch is not
                // running (the device shared with another spinning launch)
                if (__builtin_amdgcn_s_memrealtime() - t_wait > G32_STALL_TICKS) { if (lane == 0) give_up(); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (fenced) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if (!got) { if (lane == 0) info[u].state = 4; continue; }
            float mv = __builtin_inff();
            int mi = INT32_MAX, mb = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (lt_vi(rv1[q], ra1[q], mv, mi)) { mv = rv1[q]; mi = ra1[q]; mb = lane + 64 * q; }
#pragma unroll
            for (int m = 1; m <= 32; m <<= 1) {
                const float ov = __shfl_xor(mv, m, 64); const int oi = __shfl_xor(mi, m, 64), ob = __shfl_xor(mb, m, 64);
                if (lt_vi(ov, oi, mv, mi)) { mv = ov; mi = oi; mb = ob; }
            }
            int state = 3, nH = 0, sender = 0;
            double tau = 0.0;
            if (!(mv < __builtin_inff())) state = 2;
            else {
                double V2 = V2w[0];
#pragma unroll
                for (int k = 1; k < UB; ++k) V2 = u == k ? V2w[k] : V2;
                double EW = EWw[0];
#pragma unroll
                for (int k = 1; k < UB; ++k) EW = u == k ? EWw[k] : EW;
                // search_epsilon mode: the float32 minimum is the answer -- with a hoisted target term only where its
                // ABSOLUTE bound is small against the minimum (it is not for near-exact matches: decided exactly then)
                if (approx && (!HOIST || 4.0 * (errf((double)mv, V2) + EW) <= 1e-3 * (double)mv)) state = 1;
                else {
                    tau = tau_of(mv, V2, EW);
                    int nc = 0;
                    bool cov = false;
                    float v3me = __builtin_inff();
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned long long h1 = __builtin_amdgcn_ballot_w64((double)rv1[q] <= tau);
                        nH += __popcll(h1);
                        nc += __popcll(h1) + __popcll(__builtin_amdgcn_ballot_w64((double)rv2[q] <= tau));
                        cov = cov || __builtin_amdgcn_ballot_w64((double)rv3[q] <= tau) != 0ull;
                        if ((int)(blockIdx.x >> 6) == q) v3me = __shfl(rv3[q], (int)(blockIdx.x & 63u), 64);
                    }
                    if (nc == 1 && !cov) state = 1;               // the only window inside the bound is the float32 minimum itself
                    else sender = (mb != (int)blockIdx.x && (double)v3me <= tau) ? 1 : 0;
                }
            }
            if (lane == 0) {
                G32Info x;
                x.state = state; x.D = mb; x.nH = nH; x.sender = sender; x.winner = (int64_t)mi; x.tau = tau; x.mv = mv; x.pad_ = 0.f;
                info[u] = x;
            }
        }
        __syncthreads();
        stamp(step, 4);
        int st[UB];
        {
            bool leave = false, none = false;
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                st[u] = (u < a.nu && step < a.nsteps_u[u]) ? info[u].state : 0;
                leave = leave || st[u] == 4;
                none = none || st[u] == 2;
            }
            if (leave) return;                                    // the launch was ended (watchdog, an undecidable step)
            if (none) {                                           // nothing finite: every workgroup sees it, one reports it
                if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(status, (int64_t)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        // a holder of more windows than it published, and not the one who decides: its lanes' windows go to its slot
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (st[u] != 3 || !info[u].sender) continue;            // uniform
            const double tau = info[u].tau;
            if (tid == 0) { ccount[0] = 0; ccount[1] = 0; }
            __syncthreads();
            offer(u, tau, G32_SEND);
            __syncthreads();
            const int n = ccount[0];
            const bool over = ccount[1] != 0 || n > G32_SEND;
            unsigned long long *slot = send + ((size_t)u * nb + blockIdx.x) * (G32_SEND + 1);
            if (fenced && tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if (tid < n && tid < G32_SEND) st64(slot + 1 + tid, clist[tid] | ((unsigned long long)stag << 32));
            if (tid == 0) st64(slot, (unsigned long long)(over ? 0xffffu : (unsigned int)n) | ((unsigned long long)stag << 32));
            __syncthreads();
            stamp(step, 7);
        }
        unsigned long long stat_rounds = 0, stat_windows = 0;
        int64_t winner[UB];
        bool pending[UB];
        bool any_pending = false;
#pragma unroll
        for (int u = 0; u < UB; ++u) { winner[u] = 0; pending[u] = false; }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (st[u] == 1) {                                     // settled everywhere; workgroup 0 writes the path
                winner[u] = info[u].winner;
                if (blockIdx.x == 0 && tid == 0)
                    __hip_atomic_store(&path[a.out_off[u] + step], winner[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
            if (st[u] != 3) continue;
            if (info[u].D != (int)blockIdx.x) { pending[u] = true; any_pending = true; continue; }
            // ---- this workgroup published the minimum: it decides ----
            const double tau = info[u].tau;
            const int nH = info[u].nH;
            if (F16 && nH == 1 && spec_valid[u] && spec_tau[u] == tau) {
                // decided while the others were still scanning
                winner[u] = spec_win[u];
                if (tid == 0) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&status[4]), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                stat_rounds += 1; stat_windows += (unsigned long long)spec_n[u];
                if (tid == 0) {
                    if (fenced) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                    __hip_atomic_store(&path[a.out_off[u] + step], winner[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                stamp(step, 6);
                continue;
            }
            if (tid == 0 && nH > 1) { __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&status[5]), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            // (several holders, but this workgroup has decided among its OWN windows already: only that winner goes on)
            const bool own_known = F16 && spec_valid[u] && spec_tau[u] == tau;
            if (tid == 0) { ccount[0] = own_known ? 1 : 0; ccount[1] = 0; if (own_known) clist[0] = (unsigned long long)spec_win[u]; }
            __syncthreads();
            if (!own_known) offer(u, tau, G32_CAND);
            if (nH > 1) {
                // other holders: the published windows of those that kept nothing more, the lists of the others
                for (unsigned int b = tid; b < nb; b += blockDim.x) {
                    if (b == blockIdx.x) continue;
                    unsigned long long *o = pub_s + ((size_t)u * nb + b) * 4;
                    const unsigned long long g0 = ld64(o), g1 = ld64(o + 1), g2 = ld64(o + 2);
                    const float v1 = __builtin_bit_cast(float, (unsigned int)g0), v2 = __builtin_bit_cast(float, (unsigned int)g1),
                                v3 = __builtin_bit_cast(float, (unsigned int)g2);
                    if (!((double)v1 <= tau)) continue;
                    if (!((double)v3 <= tau)) {
                        { const int p = atomicAdd(ccount, 1); if (p < G32_CAND) clist[p] = (unsigned long long)((unsigned int)(g0 >> 32) >> 1); }
                        if ((double)v2 <= tau) { const int p = atomicAdd(ccount, 1); if (p < G32_CAND) clist[p] = (unsigned long long)((unsigned int)(g1 >> 32) >> 1); }
                        continue;
                    }
                    unsigned long long *slot = send + ((size_t)u * nb + b) * (G32_SEND + 1);
                    const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
                    unsigned int cnt = 0xffffu;
                    for (int i = -1; i < (int)(cnt == 0xffffu ? 0 : cnt); ) {
                        const unsigned long long g = ld64(slot + 1 + i);
                        if ((unsigned int)(g >> 32) == stag) {
                            if (i < 0) { cnt = (unsigned int)g & 0xffffu; if (cnt == 0xffffu) break; }
                            else { const int p = atomicAdd(ccount, 1); if (p < G32_CAND) clist[p] = g & 0xffffffffull; }
                            ++i;
                            continue;
                        }
                        if (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0xffffffffu
                            || __builtin_amdgcn_s_memrealtime() - t_wait > G32_STALL_TICKS) { cnt = 0xffffu; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (cnt == 0xffffu) ccount[1] = 1;              // more than the slot takes, or no news
                }
                if (fenced) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            }
            __syncthreads();
            const int n = ccount[0];
            const bool undecided = ccount[1] != 0 || n > G32_CAND || n < 1;
            __syncthreads();
            stamp(step, 5);
            if (undecided) {
                if (tid == 0) {
                    __hip_atomic_store(status, (int64_t)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(gen, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;                                           // uniform over the workgroup; the others see the generation word
            }
            if (n == 1) winner[u] = (int64_t)clist[0];
            else {
                stat_rounds += 1; stat_windows += (unsigned long long)n;      // statistics: decisions by exact totals, their windows
                winner[u] = exact_argmin(u, n, [&](int p) { return (int64_t)clist[p]; }, info[u].mv, V2w[u], EWw[u]);
            }
            // the winner IS the release
            if (tid == 0) {
                if (fenced) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                __hip_atomic_store(&path[a.out_off[u] + step], winner[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            stamp(step, 6);
        }
        if (tid == 0 && (stat_rounds | stat_windows)) {           // statistics, behind the release
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&status[1]), stat_rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&status[2]), stat_windows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // winners decided elsewhere: the path entries (-1 until written), or the end of the launch (generation word)
        if (any_pending) {
            if (tid == 0) {
                int seen = 0;
                const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    const unsigned int g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool ready = true;
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        int64_t p = 0;
                        if (pending[u]) {
                            p = __hip_atomic_load(&path[a.out_off[u] + step], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (p < 0) ready = false;
                        }
                        next_rows[u] = p;
                    }
                    if (g == 0xffffffffu) { seen = -1; break; }
                    if (ready) break;
                    if (__builtin_amdgcn_s_memrealtime() - t_wait > G32_STALL_TICKS) { give_up(); seen = -1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (fenced) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                gen_seen = seen;
            }
            __syncthreads();
            if (gen_seen < 0) return;                             // an undecidable step: everybody leaves
#pragma unroll
            for (int u = 0; u < UB; ++u) if (pending[u]) winner[u] = next_rows[u];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
            if (u < a.nu && step < a.nsteps_u[u]) prev_row[u] = g32_uniform_i(winner[u]);
        __syncthreads();                                          // the next table is built over this step's scratch
    }
}

// prologue: generation word, status
__global__ void greedy32_init_kernel(unsigned int *gen, int64_t *status)
{
    if (threadIdx.x == 0) { *status = 0; *gen = 0u; for (int i = 1; i < 16; ++i) status[i] = 0; }
}

// exact Euclidean distance of every pick (what the tree query returns beside the index): a wavefront per step
__global__ void __launch_bounds__(64)
greedy32_dist_kernel(GreedyArgs a, int u, int64_t start, const int64_t *__restrict__ path, double *__restrict__ dist)
{
    extern __shared__ __align__(16) char lds[];
    const int64_t s = blockIdx.x;
    if (s >= a.nsteps_u[u]) return;
    const int64_t prev = s == 0 ? start : path[a.out_off[u] + s - 1];
    const double d2 = g32_exact_d2_wave(a, u, s, prev, s > 0, path[a.out_off[u] + s], reinterpret_cast<double *>(lds), threadIdx.x);
    if (threadIdx.x == 0) dist[a.out_off[u] + s] = __dsqrt_rn(d2);
}

size_t greedy32_table_floats(const GreedyLayout &g, int Dt, bool hoist)
{
    const int nep = (g.last_frame_as_target && g.me > 1) ? 2 : g.me;
    const int jch = (g.jdim + GR_CC - 1) / GR_CC;
    return (size_t)(jch + (hoist ? 0 : nep * ((Dt + GR_CC - 1) / GR_CC))) * GR_CC * (hoist ? 8 : 4);      // hoisted: room for the wide entries
}
// workspace of one launch: the published records (two step parities x utterances x workgroups x four granules; all bits set
// before the launch: the tag bit of steps 0 and 1 is 0) | the holders' slots (zero before the launch: no tag is 0)
static size_t g32_pub_bytes(int nblk) { return (size_t)2 * G32_UBX * nblk * 4 * sizeof(unsigned long long); }
static size_t g32_send_bytes(int nblk) { return (size_t)G32_UBX * nblk * (G32_SEND + 1) * sizeof(unsigned long long); }
size_t greedy32_block_bytes(int nblk) { return g32_pub_bytes(nblk) + g32_send_bytes(nblk); }
int greedy32_max_utts(bool hoist) { return hoist ? G32_UBX : G32_UB; }

static size_t g32_lds_wave_bytes(const GreedyLayout &g, int Dt)
{
    return (size_t)(G32_W + g.me - 1) * ((Dt + GR_CC - 1) / GR_CC * GR_CC + 4) * sizeof(float);
}
// wavefronts per workgroup / LDS bytes: the table (16 bytes per column) + a target block per wavefront
static int g32_waves(const GreedyLayout &g, int Dt, int n_cus, bool in_lds, bool hoist = false)
{
    if (hoist) in_lds = false;
    const int64_t ntiles = (g.Nwin + G32_W - 1) / G32_W;
    int64_t w = (ntiles + n_cus - 1) / n_cus;
    int wmax = G32_MAXW;
    if (in_lds) {
        const size_t fixed = greedy32_table_floats(g, Dt, false) * 4 + 512;
        const size_t per = g32_lds_wave_bytes(g, Dt);
        const size_t fit = fixed < (size_t)(160 * 1024) ? ((size_t)(160 * 1024) - fixed) / per : 0;
        if ((int64_t)fit < wmax) wmax = (int)fit;
    }
    if (w > wmax) w = wmax;
    return (int)(w < 1 ? 1 : w);
}
bool greedy32_supported(const GreedyLayout &g, int Dt)
{
    // the table must leave room for the reduction scratch and, in LDS mode, for at least four target blocks
    const size_t tb = greedy32_table_floats(g, Dt, false) * 4;
    if (tb + 16384 > (size_t)(160 * 1024)) return false;
    if (greedy_lds_mode(g, Dt) && tb + 512 + 4 * g32_lds_wave_bytes(g, Dt) > (size_t)(160 * 1024)) return false;
    return true;
}
int greedy32_blocks(const GreedyLayout &g, int Dt, int n_cus, bool hoist)
{
    const bool in_lds = greedy_lds_mode(g, Dt);
    const int64_t ntiles = (g.Nwin + G32_W - 1) / G32_W;
    const int waves = g32_waves(g, Dt, n_cus, in_lds, hoist);
    const int64_t need = (ntiles + waves - 1) / waves;
    const int cap = n_cus < 256 ? n_cus : 256;              // the deciding workgroup keeps four records per thread
    return (int)(need < cap ? need : cap);
}

// developer aid: SNK_G32_TRACE=<file> writes the last launch's timeline (see `stamp` in the kernel) after the caller's sync
static void *g32_trace_dev = nullptr;
static int g32_trace_blocks = 0;
void greedy32_trace_dump()
{
    const char *fn = getenv("SNK_G32_TRACE");
    if (!fn || !g32_trace_dev || !g32_trace_blocks) return;
    const size_t n = (size_t)G32_TRACE_STEPS * g32_trace_blocks * 16;
    std::vector<unsigned long long> host(n);
    if (hipMemcpy(host.data(), g32_trace_dev, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
    if (FILE *f = fopen(fn, "wb")) {
        const long long hdr[2] = {G32_TRACE_STEPS, g32_trace_blocks};
        fwrite(hdr, sizeof(hdr), 1, f);
        fwrite(host.data(), sizeof(unsigned long long), n, f);
        fclose(f);
    }
}

// One persistent launch for up to three utterances (q_off / nsteps_u / out_off / start per utterance).
// approx bit 0: search_epsilon mode (float32 minimum, nothing re-evaluated); bit 8: test hook (a workgroup that never arrives).  *status (device): 0, or 1 + the
// first step that could not be decided (mass ties); the caller then falls back to the exact scan.
void launch_greedy32(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt, const float *JC_unw, int Jp,
                     int Dj, const double *wj, const float *tiles, const double *Q, int nu, const int64_t *q_off,
                     const int64_t *nsteps_u, const int64_t *out_off, const int64_t *start, int approx,
                     void *blk, int n_cus, unsigned int *gen, int64_t *status,
                     int64_t *path, const G32Hoist *hst, hipStream_t s)
{
    GreedyArgs a{};
    greedy_fill_args(a, g, F_unw, Fp, Dt, wt, JC_unw, Jp, Dj, wj, Q, true, tiles);
    const bool hoist = hst != nullptr;
    const bool in_lds = !hoist && greedy_lds_mode(g, Dt);
    a.lds_mode = in_lds ? 1 : 0;
    a.nu = nu;
    if (hoist) {
        a.hoist = 1; a.Wp = hst->Wp; a.hoist_c = hst->c; a.fwmax2 = hst->fwmax2;
        for (int u = 0; u < G32_UBX; ++u) { a.W[u] = u < nu ? hst->W[u] : nullptr; a.qn2[u] = u < nu ? hst->qn2[u] : nullptr; }
    }
    int64_t nsteps = 0;
    for (int u = 0; u < G32_UBX; ++u) {
        a.q_off[u] = u < nu ? q_off[u] : 0;
        a.nsteps_u[u] = u < nu ? nsteps_u[u] : 0;
        a.out_off[u] = u < nu ? out_off[u] : 0;
        a.start[u] = u < nu ? start[u] : -1;
        if (u < nu && nsteps_u[u] > nsteps) nsteps = nsteps_u[u];
    }
    if (nsteps <= 0) return;
    const bool wide = hoist && nu > G32_UB;                        // four to six utterances: the wide table entries
    hipLaunchKernelGGL(greedy32_init_kernel, dim3(1), dim3(64), 0, s, gen, status);
    // the winners are the hand-off between the steps: a path entry is -1 until its step is decided
    for (int u = 0; u < nu; ++u)
        if (nsteps_u[u] > 0) (void)hipMemsetAsync(path + out_off[u], 0xff, (size_t)nsteps_u[u] * sizeof(int64_t), s);
    const int waves = g32_waves(g, Dt, n_cus, in_lds, hoist);
    const int nblk = greedy32_blocks(g, Dt, n_cus, hoist);
    size_t lds = greedy32_table_floats(g, Dt, hoist) * 4 + 512 + (in_lds ? (size_t)waves * g32_lds_wave_bytes(g, Dt) : 0);
    // the step's tail: 16 KB of reduction scratch + a term array per wavefront for exact decisions
    const int nep = (g.last_frame_as_target && g.me > 1) ? 2 : g.me;
    const size_t per_wave = (size_t)(g.jdim + nep * Dt) * 8;
    size_t tail = 16384 + (size_t)waves * per_wave;
    for (int k = 4; k >= 2; --k)
        if (16384 + (size_t)waves * per_wave * k <= (size_t)(144 * 1024)) { tail = 16384 + (size_t)waves * per_wave * k; break; }
    if (lds < tail) lds = tail < (size_t)(160 * 1024) ? tail : (size_t)(160 * 1024);
    // small scans live in L2 / Infinity Cache across the steps of the launch; big ones are streamed
    const size_t scan_bytes = (size_t)g.Nwin * (size_t)(g.jdim + (hoist ? 1 : Dt)) * 4;
    // (2: the hoisted scan sends the requests of all-padding float4 columns to the chunk's first column)
    const int use_nt = scan_bytes > ((size_t)192 << 20) ? (hoist ? 2 : 1) : 0;
    // float16 join tiles: databases that are streamed from HBM, up to three utterances per scan
    const bool f16 = hoist && hst->JT16 != nullptr && !wide && (use_nt != 0 || hst->f16_force);
    if (f16) { a.JT16 = reinterpret_cast<const f32x4 *>(hst->JT16); a.f16 = 1; a.f16_delta = hst->f16_delta; }
    auto kernel = f16 ? (nu == 1 ? greedy32_kernel<false, true, 1, true> : greedy32_kernel<false, true, G32_UB, true>)
                  : wide ? greedy32_kernel<false, true, G32_UBX, false> : hoist ? greedy32_kernel<false, true, G32_UB, false>
                  : in_lds ? greedy32_kernel<true, false, G32_UB, false> : greedy32_kernel<false, false, G32_UB, false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    char *wb = reinterpret_cast<char *>(blk);
    unsigned long long *pub = reinterpret_cast<unsigned long long *>(wb);
    unsigned long long *send = reinterpret_cast<unsigned long long *>(wb + g32_pub_bytes(nblk));
    (void)hipMemsetAsync(pub, 0xff, g32_pub_bytes(nblk), s);
    (void)hipMemsetAsync(send, 0, g32_send_bytes(nblk), s);
    unsigned long long *trace = nullptr;
    if (getenv("SNK_G32_TRACE")) {
        const size_t tb = (size_t)G32_TRACE_STEPS * nblk * 16 * sizeof(unsigned long long);
        if (!g32_trace_dev) (void)hipMalloc(&g32_trace_dev, (size_t)G32_TRACE_STEPS * 1024 * 16 * sizeof(unsigned long long));
        (void)hipMemsetAsync(g32_trace_dev, 0, tb, s);
        trace = reinterpret_cast<unsigned long long *>(g32_trace_dev);
        g32_trace_blocks = nblk;
    }
    hipLaunchKernelGGL(kernel, dim3(nblk), dim3(G32_W * waves), lds, s, a, nsteps, approx, use_nt, (int)lds,
                       pub, send, gen, path, status, trace);
}

void launch_greedy32_dist(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt, const float *JC_unw, int Jp,
                          int Dj, const double *wj, const double *Q, int u_slot, int64_t q_off, int64_t nsteps, int64_t out_off,
                          int64_t start, const int64_t *path, double *dist, hipStream_t s)
{
    if (nsteps <= 0) return;
    GreedyArgs a{};
    greedy_fill_args(a, g, F_unw, Fp, Dt, wt, JC_unw, Jp, Dj, wj, Q, true, nullptr);
    a.nu = 1;
    (void)u_slot;
    a.q_off[0] = q_off; a.nsteps_u[0] = nsteps; a.out_off[0] = out_off;
    const int nep = (g.last_frame_as_target && g.me > 1) ? 2 : g.me;
    hipLaunchKernelGGL(greedy32_dist_kernel, dim3((unsigned)nsteps), dim3(64), (size_t)(g.jdim + nep * Dt) * 8, s, a, 0, start, path, dist);
}

}  // namespace snk
