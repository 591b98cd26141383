// Join costs off the float64 vector pipe: lower bounds on the matrix pipe, exact values only where the
// recursion can see them.
//
// Replaces, together with viterbi_kernels.hip (which stays as the dense exact path and the A/B partner):
//   make_on_the_fly_join_lattice_BLOCK_DIRECT / get_natural_distance_vectorised
//                                               (script/synth_halfphone.py:3206-3322, :2942-2951)
//   make_target_sausage_lattice / cost_cache_to_compiled_fst / openfst.compose / openfst.shortestpath
//                                               (script/fst_functions_wrapped.py:28-58,172-217,368,389)
//
// The recursion  delta_t[k] = tdist[t,k] + min_k' ( delta_{t-1}[k'] + c(k',k) )  needs the EXACT canonical
// float64 join cost c only for the predecessors that can win.  Four passes:
//
//   1  join_lb_kernel        (all rows in parallel, f32 matrix pipe)  clo(k',k) <= c(k',k), PROVEN:
//        rows are centred on a per-step reference row m (float64 subtraction, then rounded to f32:
//        candidates of one step are close to each other, so a plain f32 ||e||^2+||s||^2-2e.s would lose
//        every digit to cancellation), G = ye~ . ys~ on v_mfma_f32_16x16x4_f32,
//        c2~ = ne + ns - 2G,  |c2~ - c^2| <= e2 := 2.2 (D+6) 2^-24 (ne + ns),  clo = sqrt(max(c2~ - e2, 0)) (1 - 2^-21).
//   2  viterbi_lb_kernel     (one workgroup per utterance)  the recursion on clo: dlb_t[k] <= delta_t[k];
//        per (t,k) the predecessors within theta of the minimum (at most JF_CAP of them; their slots) and
//        X(t,k) = the smallest lower-bound total among all the OTHER predecessors.
//   3  join_exact_sparse_kernel (parallel)  canonical float64 c for the recorded predecessors only
//        (about 1 % of the K x K pairs).
//   4  viterbi_sparse_kernel (one workgroup per utterance)  the exact recursion over the recorded sets,
//        with a proof per (t,k) that no other predecessor can win or tie:
//             X(t,k) + min_k'( delta_{t-1}[k'] - dlb_{t-1}[k'] )  >  best exact total
//        (every excluded k' has delta[k'] + c >= dlb[k'] + off + clo >= X + off).  Where the proof fails,
//        or a set overflowed, that (t,k) is recomputed densely on the spot: all K exact join costs of its
//        column.  Path and cost are therefore bit-identical to the dense exact recursion -- the margins
//        only decide how often the slow branch runs, never the result.
#include "snk_internal.h"
#include <float.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

namespace snk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define JF_CAP 4              // recorded predecessors per (t,k); more: that column is recomputed densely

__device__ __forceinline__ bool jf_usable(int64_t id, int64_t n_units)
{
    return id >= 1 && id < n_units - 1;        // synth_halfphone.py:3238-3268
}

// ---------------------------------------------------------------------------------------------
// pass 1
// One workgroup per pair of consecutive candidate rows (r, r+1); KT wavefronts; wavefront w owns the
// 16 E rows (unit_end_data of cand[r, 16w .. 16w+15]) and stages the 16 S rows (unit_start_data of
// cand[r+1, 16w ..]) of its number.  Columns go in chunks of 64 = 4 MFMA blocks of 16:
//   lane l <-> row l & 15, columns 16 b + 4 (l >> 4) + i, i = 0..3, of block b
// which is at once the A / B operand map of v_mfma_f32_16x16x4_f32 for the k-steps i = 0..3 (k = l >> 4
// is then column 4 k + i of the block: a permutation of the sum, the same on both operands).
// The E fragments stay in registers; the S fragments of all KT tiles go through LDS in fragment order
// (a wavefront writes / reads 1 KB of consecutive addresses per instruction: conflict-free), double
// buffered, one barrier per chunk.  The next chunk's rows are in flight during the MFMAs.
// ---------------------------------------------------------------------------------------------
#define JF_MAXD 1024           // join columns (padded to 16) the w / m tables in LDS hold

template <int KT>
__global__ void __launch_bounds__(64 * KT, (KT >= 5 && KT <= 8) ? 4 : 0)      // two workgroups per compute unit: the other one's MFMAs cover a chunk's staging
join_lb_kernel(const float *__restrict__ JC_unw, int Jp, int Dj, const double *__restrict__ wj, int64_t n_units,
               const int64_t *__restrict__ cand, int K, float *__restrict__ Jlo, float *__restrict__ scale_out)
{
    __shared__ f32x4 Bs[2][KT][(KT >= 5) ? 2 : 4][64];
    __shared__ double w_s[JF_MAXD], m_s[JF_MAXD];
    __shared__ float ne_s[16 * KT], ns_s[16 * KT];
    __shared__ int first_ok, smax_bits;
    const int64_t r = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row16 = lane & 15, q = lane >> 4;
    const int DC = (Dj + 15) & ~15;
    const int n_blocks = DC / 16;

    // this lane's E row and S row
    const int kk = wave * 16 + row16;
    const int64_t idE = kk < K ? cand[r * K + kk] : -1;
    const int64_t idS = kk < K ? cand[(r + 1) * K + kk] : -1;
    const bool okE = jf_usable(idE, n_units), okS = jf_usable(idS, n_units);
    const float *const rowE = JC_unw + (okE ? idE + 1 : 0) * (int64_t)Jp;     // unit_end_data[a]   = JC[a+1]
    const float *const rowS = JC_unw + (okS ? idS : 0) * (int64_t)Jp;         // unit_start_data[b] = JC[b]
    if (tid == 0) { first_ok = 1 << 30; smax_bits = 0; }
    __syncthreads();
    {
        const unsigned long long bal = __ballot(okS && q == 0);
        if (lane == 0 && bal) atomicMin(&first_ok, wave * 16 + __builtin_ctzll(bal));
    }
    __syncthreads();
    // reference row m: the start vector of the first usable candidate of row r+1 (any row would do:
    // the bound below is in terms of the centred norms, whatever m is)
    {
        const int k0 = first_ok;
        const int64_t id0 = (k0 < K) ? cand[(r + 1) * K + k0] : 0;
        const float *row0 = JC_unw + (k0 < K ? id0 : 0) * (int64_t)Jp;
        for (int c = tid; c < DC; c += (int)blockDim.x) {
            const double w = c < Dj ? wj[c] : 0.0;
            w_s[c] = w;
            m_s[c] = c < Dj ? __dmul_rn((double)row0[c], w) : 0.0;
        }
    }
    __syncthreads();

    // blocks per chunk: 4 (64 columns); 2 for the widest tiles, whose 4 KT accumulators leave fewer registers
    constexpr int CB = (KT >= 5) ? 2 : 4;
    f32x4 acc[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ne = 0.f, ns = 0.f;

    // raw rows of one chunk: CB blocks x 16 bytes per matrix and lane
    f32x4 rawE[CB], rawS[CB];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int b = 0; b < CB; ++b) {
            int c0 = (ch * CB + b) * 16 + 4 * q;
            if (c0 > Jp - 4) c0 = Jp - 4;                  // padded blocks: any readable address (weights are 0 there)
            rawE[b] = *reinterpret_cast<const f32x4 *>(rowE + c0);
            rawS[b] = *reinterpret_cast<const f32x4 *>(rowS + c0);
        }
    };
    const int n_chunks = (n_blocks + CB - 1) / CB;
    fetch(0);
    for (int ch = 0; ch < n_chunks; ++ch) {
        // centre in float64, round to float32, norms in float32 -- block by block (the fence keeps the
        // compiler from hoisting every block's table reads to the top: 64 registers)
        f32x4 a[CB];
#pragma unroll
        for (int b = 0; b < CB; ++b) {
            const int c0 = (ch * CB + b) * 16 + 4 * q;
            double w[4], m[4];
            if (c0 < DC) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { w[i] = w_s[c0 + i]; m[i] = m_s[c0 + i]; }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { w[i] = 0.0; m[i] = 0.0; }
            }
            f32x4 ys;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e = (float)__dsub_rn(__dmul_rn((double)rawE[b][i], w[i]), m[i]);
                const float s = (float)__dsub_rn(__dmul_rn((double)rawS[b][i], w[i]), m[i]);
                const bool live = c0 + i < Dj;             // columns beyond Dj read a neighbour's bytes: force 0
                a[b][i] = live ? e : 0.f;
                ys[i] = live ? s : 0.f;
                ne = __builtin_fmaf(a[b][i], a[b][i], ne);
                ns = __builtin_fmaf(ys[i], ys[i], ns);
            }
            Bs[ch & 1][wave][b][lane] = ys;
            asm volatile("" ::: "memory");
        }
        if (ch + 1 < n_chunks) fetch(ch + 1);
        __syncthreads();
        const int nb = (n_blocks - ch * CB) < CB ? (n_blocks - ch * CB) : CB;
#pragma unroll
        for (int b = 0; b < CB; ++b) {
            if (b >= nb) break;                              // uniform
            // tiles in groups of JG: consecutive MFMAs go to different accumulators (40-cycle dependent
            // latency against a 32-cycle issue interval), the group's S fragments are all the registers needed
            constexpr int JG = (KT <= 4) ? KT : 4;
#pragma unroll
            for (int j0 = 0; j0 < KT; j0 += JG) {
                f32x4 bf[JG];
#pragma unroll
                for (int j = 0; j < JG; ++j)
                    if (j0 + j < KT) bf[j] = Bs[ch & 1][j0 + j][b][lane];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < JG; ++j)
                        if (j0 + j < KT)
                            acc[j0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b][i], bf[j][i], acc[j0 + j], 0, 0, 0);
            }
        }
    }
    // row norms: the four lanes of a row hold the partial sums of its four column quarters
    ne += __shfl_xor(ne, 16, 64); ne += __shfl_xor(ne, 32, 64);
    ns += __shfl_xor(ns, 16, 64); ns += __shfl_xor(ns, 32, 64);
    if (q == 0) {
        ne_s[kk] = okE ? ne : __builtin_inff();
        ns_s[kk] = okS ? ns : __builtin_inff();
        // scale of the step (margin of pass 2): the largest centred norm among the usable rows
        const float big = fmaxf(okE ? ne : 0.f, okS ? ns : 0.f);
        atomicMax(&smax_bits, __float_as_int(big));
    }
    __syncthreads();
    // |c2~ - c^2| <= e2: operand rounding 2^-24 each, two f32 FMA chains of DC terms for the norms, one for
    // G, two f32 operations in the epilogue, (||ye|| + ||ys||)^2 <= 2 (ne + ns)
    const float ceps = 2.2f * (float)(DC + 6) * 5.9604644775390625e-08f;
    float *const out = Jlo + r * (int64_t)K * K;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const int k = j * 16 + row16;                       // column of the result = S row
        const float nsv = ns_s[k];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kp = wave * 16 + 4 * q + i;           // row of the result = E row
            const float nev = ne_s[kp];
            const float sum = nev + nsv;
            const float c2 = sum - 2.f * acc[j][i];
            const float lo2 = c2 - (ceps * sum + 1e-30f);
            float clo = 0.f;                                 // NaN / overflow: 0 is a valid lower bound
            if (lo2 > 0.f && lo2 < __builtin_inff()) clo = __builtin_sqrtf(lo2) * (1.f - 4.76837158203125e-07f);
            if (!(sum < __builtin_inff())) clo = (sum == __builtin_inff()) ? __builtin_inff() : 0.f;   // unusable unit: +inf
            if (kp < K && k < K) out[(int64_t)kp * K + k] = clo;
        }
    }
    if (tid == 0) scale_out[r] = __builtin_sqrtf(__int_as_float(smax_bits));
}

template <int KT>
static void launch_join_lb_t(const float *JC_unw, int Jp, int Dj, const double *wj, int64_t n_units, const int64_t *cand,
                             int64_t R, int K, float *Jlo, float *scale, hipStream_t s)
{
    hipLaunchKernelGGL((join_lb_kernel<KT>), dim3((unsigned)(R - 1)), dim3(64 * KT), 0, s, JC_unw, Jp, Dj, wj, n_units,
                       cand, K, Jlo, scale);
}

bool join_lb_supported(int Dj, int K) { return ((Dj + 15) & ~15) <= JF_MAXD && K >= 1 && K <= 208; }

// the error budget of the squared cost, as a fraction of the two centred norms' sum, of the form of pass 1 in use (the unit of the
// tripwire's margin): the float32 form's 2.2 (DC + 6) 2^-24 (join_lb_kernel), the bf16 form's join_lb2_ceps
double join_lb_ceps(int variant, int Dj, int K)
{
    if (variant == 1) return join_lb2_ceps(Dj, K);
    return 2.2 * (double)(((Dj + 15) & ~15) + 6) * 5.9604644775390625e-08;
}

// (pass 1, second form -- bf16 matrix pipe over a weighted float32 copy of the join rows: joinlb2_kernels.hip)

void launch_join_lb(const float *JC_unw, int Jp, int Dj, const double *wj, int64_t n_units, const int64_t *cand,
                    int64_t R, int K, float *Jlo, float *scale, hipStream_t s)
{
    if (R < 2) return;
    const int kt = (K + 15) / 16;
#define SNK_JLB(KT_) launch_join_lb_t<KT_>(JC_unw, Jp, Dj, wj, n_units, cand, R, K, Jlo, scale, s)
    if (kt <= 1) SNK_JLB(1);
    else if (kt <= 2) SNK_JLB(2);
    else if (kt <= 4) SNK_JLB(4);
    else if (kt <= 5) SNK_JLB(5);
    else if (kt <= 7) SNK_JLB(7);
    else if (kt <= 8) SNK_JLB(8);
    else if (kt <= 10) SNK_JLB(10);
    else SNK_JLB(13);
#undef SNK_JLB
}


// ---------------------------------------------------------------------------------------------
// pass 2: an APPROXIMATE recursion on the lower bounds, in float32, + the predecessor sets.
// Pass 4's proof needs no property of d~ itself, only that X is consistent with it:
//     X(t,k) <= d~_{t-1}[k'] + clo(k',k)   for every k' outside the set, in exact arithmetic on the stored values
// (then delta[k'] + c >= d~[k'] + off + clo >= X + off with off = min_k'(delta[k'] - d~[k'])).  So d~ runs in
// float32 (X is lowered by 2 ulp for the rounding of the sum) and is shifted every 64 steps to keep its
// magnitude near the costs of a step (the shift is common to all columns of a step: off absorbs it).
// Layout of viterbi_dp_kernel (viterbi_kernels.hip): a wavefront owns 16 columns k, its four 16-lane
// rows split the predecessors; slabs of the next NB steps in flight in registers (float32: half the
// stream).  After the minimum a second sweep over the slab registers appends every predecessor within
// theta of it to the column's list in LDS (the four lanes of a column sit in one wavefront: LDS
// operations of a wavefront execute in order, no barrier) and reduces the rest to X.
// theta = beta * scale[t-1] + 4e-7 |min|: scale is the step's largest centred norm (pass 1).
// Output per cell, 16 bytes: { d~ (f32), X (f32), slots (4 x u8), n }.
// ---------------------------------------------------------------------------------------------
template <int KPM, int NB, int NTH>
__global__ void __launch_bounds__(NTH)
viterbi_lb_kernel(const int64_t *__restrict__ cand_all, const double *__restrict__ tdist_all,
                  const float *__restrict__ J_all, const float *__restrict__ scale_all, const DpBatch batch, int K,
                  int64_t n_units, int KP, float beta, u32x4 *__restrict__ sets_all, int64_t chunk_len, int warm, float slack32)
{
    // a T-step chain on one compute unit beside the K-NN sweep's MFMA wavefronts: its few instructions go first
    __builtin_amdgcn_s_setprio(3);
    const int64_t r0 = batch.off[blockIdx.x];
    const int64_t T = batch.off[blockIdx.x + 1] - r0;
    const int64_t *__restrict__ cand = cand_all + r0 * K;
    const double *__restrict__ tdist = tdist_all + r0 * K;
    const float *__restrict__ J = J_all + r0 * K * K;
    const float *__restrict__ scale = scale_all + r0;
    u32x4 *__restrict__ sets = sets_all + r0 * K;

    extern __shared__ __align__(16) unsigned char smem[];
    float *delta = reinterpret_cast<float *>(smem);                // [2][KP]
    int *lcnt = reinterpret_cast<int *>(delta + 2 * KP);           // [KP]
    unsigned int *lidx = reinterpret_cast<unsigned int *>(lcnt + KP);     // [KP] four slots each
    float *wmin = reinterpret_cast<float *>(lidx + KP);            // [16] per-wavefront minima (shift steps)

    const int tid = threadIdx.x, lane = tid & 63;
    const int k = (tid >> 6) * 16 + (lane & 15);
    const int part = lane >> 4, kp0 = part * KPM;
    const bool col = k < K;
    const bool lead = col && part == 0;
    const float inf = __builtin_inff();
    const int nwaves = (int)blockDim.x >> 6;

    if (T < 1) return;
    // Chunks in time (blockIdx.y; chunk_len >= T: one chunk, the whole utterance).  Chunk c owns the records of the steps
    // t0 .. t1 - 1 and starts `warm` steps earlier from d~ = the target costs of row ts - 1, as if the utterance began
    // there: d~ needs no property of its own (above), only X must be consistent with the STORED d~ of the step before.
    // So the record of step t0 - 1 is shared: its d~ is written by chunk c (the value its X of step t0 was computed
    // against), its X / set by chunk c - 1 (computed against that chunk's own d~ of step t0 - 2, stored by it).
    // After a few dozen steps the two d~ differ by nearly a constant (the paths into step t0 - 1 have merged), which
    // pass 4's `off` absorbs; what remains only decides how often pass 4 refines.
    const int chunk = (int)blockIdx.y;
    const int64_t t0 = 1 + chunk * chunk_len;
    if (chunk > 0 && t0 >= T) return;
    const int64_t t1 = t0 + chunk_len < T ? t0 + chunk_len : T;
    const bool last_chunk = t1 >= T;
    const int64_t ts = (chunk == 0 || t0 - warm < 1) ? 1 : t0 - warm;
    for (int i = tid; i < 2 * KP; i += (int)blockDim.x) delta[i] = inf;
    for (int i = tid; i < KP; i += (int)blockDim.x) { lcnt[i] = 0; lidx[i] = 0; }
    if (tid < 16) wmin[tid] = inf;
    __syncthreads();
    if (lead) {
        const float d0 = jf_usable(cand[(ts - 1) * K + k], n_units) ? (float)tdist[(ts - 1) * K + k] : inf;
        delta[((ts - 1) & 1) * KP + k] = d0;
        if (chunk == 0) sets[k] = (u32x4){__builtin_bit_cast(unsigned int, d0), __builtin_bit_cast(unsigned int, inf), 0u, 0u};
    }
    if (T < 2) return;

    const int voff = col ? (kp0 * K + k) * 4 : 0x7ffffffc;
    float jb[NB][KPM];
    auto load_slab = [&](int64_t slab, float (&dst)[KPM]) {
        const __amdgpu_buffer_rsrc_t jres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(J + slab * K * K), 0, K * K * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < KPM; ++i)
            dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(jres, voff, i * K * 4, 0));
    };
    double td_raw[NB];
    int64_t id_raw[NB];
    float sc_raw[NB];
    auto load_target = [&](int64_t t, double &td, int64_t &id, float &sc) {
        const int64_t tc = t < T ? t : T - 1;
        const int64_t tn = tc * K + (col ? k : 0);
        td = tdist[tn];
        id = cand[tn];
        sc = scale[tc - 1];
    };
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        load_slab(ts - 1 + s < T - 2 ? ts - 1 + s : T - 2, jb[s]);
        load_target(ts + s, td_raw[s], id_raw[s], sc_raw[s]);
    }
    __syncthreads();

    // Every total is a non-negative float (or +inf): ordered like its bit pattern, so minima are integer
    // minima (fminf costs a canonicalising v_max per operand) and the sweeps have no branch per element.
    const unsigned int INFU = 0x7f800000u;
    // viterbi_weights 1 (slack32 > 0): pass 4 then runs OpenFST's float32 chain, where an excluded predecessor must exceed the
    // best total by two float32 roundings of the ABSOLUTE total (2.4e-7 of it: 7e-4 at the end of a B* utterance, several times
    // theta) for the proof to hold -- the sets are widened by that much.  A chunk knows its totals only since its own start
    // (`cum`: the shifts it has applied): the absolute total is estimated from their growth per step.  Heuristic on purpose:
    // the sets decide how often pass 4 refines, never what it returns.
    float cum = 0.f;
    auto step = [&](int64_t t, float (&jr)[KPM], double &tdr, int64_t &idr, float &scr) {
        const bool valid = t < t1;                      // uniform
        const float *dprev = delta + ((t - 1) & 1) * KP;
        float *dcur = delta + (t & 1) * KP;
        const float td = jf_usable(idr, n_units) ? (float)tdr : inf;
        const float sc = scr;
        load_target(t + NB, tdr, idr, scr);
        // the step after a shift step: every d~ of the previous step is read lowered by the block's minimum
        float shift = 0.f;
        unsigned int vu[KPM];
        unsigned int bestu = INFU;
        if (((t - 1) & 63) == 0 && t > 1) {             // uniform, one step in 64
            shift = inf;
            for (int w = 0; w < nwaves; ++w) { const float o = wmin[w]; shift = o < shift ? o : shift; }
            if (!(shift < inf)) shift = 0.f;
            cum += shift;
#pragma unroll
            for (int i = 0; i < KPM; ++i) {
                vu[i] = __builtin_bit_cast(unsigned int, (dprev[kp0 + i] - shift) + jr[i]);
                bestu = vu[i] < bestu ? vu[i] : bestu;
            }
        } else {
#pragma unroll
            for (int i = 0; i < KPM; ++i) {
                // predecessors beyond K: d~ stays +inf (and an out-of-range buffer load returns 0)
                vu[i] = __builtin_bit_cast(unsigned int, dprev[kp0 + i] + jr[i]);
                bestu = vu[i] < bestu ? vu[i] : bestu;
            }
        }
        load_slab(t - 1 + NB < T - 2 ? t - 1 + NB : T - 2, jr);
        {
            unsigned int o = (unsigned int)__shfl_xor((int)bestu, 16, 64); bestu = o < bestu ? o : bestu;
            o = (unsigned int)__shfl_xor((int)bestu, 32, 64); bestu = o < bestu ? o : bestu;
        }
        const float best = __builtin_bit_cast(float, bestu);
        // second sweep: the set within theta of the minimum (a bit per element), the minimum of the rest
        float theta = beta * sc + 4e-7f * best;
        if (slack32 > 0.f) theta += slack32 * ((best + cum) * ((float)t / (float)(t - ts + 1)));
        const float thr = best + theta;
        const unsigned int thru = bestu < INFU ? __builtin_bit_cast(unsigned int, thr) : 0u;
        unsigned int xminu = INFU, mem_lo = 0u, mem_hi = 0u;
#pragma unroll
        for (int i = 0; i < KPM; ++i) {
            const bool in = vu[i] <= thru;
            if (i < 32) mem_lo |= in ? (1u << i) : 0u;
            else mem_hi |= in ? (1u << (i - 32)) : 0u;
            const unsigned int out = in ? INFU : vu[i];
            xminu = out < xminu ? out : xminu;
        }
        if (col && valid && (mem_lo | mem_hi)) {         // a handful of lanes, one or two members each
            for (int word = 0; word < 2; ++word)
                for (unsigned int m = word ? mem_hi : mem_lo; m; m &= m - 1) {
                    const int i = 32 * word + __builtin_ctz(m);
                    const int slot = atomicAdd(&lcnt[k], 1);
                    if (slot < JF_CAP) reinterpret_cast<unsigned char *>(lidx)[k * JF_CAP + slot] = (unsigned char)(kp0 + i);
                }
        }
        {
            unsigned int o = (unsigned int)__shfl_xor((int)xminu, 16, 64); xminu = o < xminu ? o : xminu;
            o = (unsigned int)__shfl_xor((int)xminu, 32, 64); xminu = o < xminu ? o : xminu;
        }
        const float xmin = __builtin_bit_cast(float, xminu);
        float d = inf;
        if (lead && valid) {
            d = td + best;
            dcur[k] = d;
            // X two ulp down: fl32(a + b) may lie above a + b; the shift is applied to the STORED d~ of the
            // previous step as well (pass 4 sees d~_{t-1} as stored), so it goes back in here
            float x = xmin < inf ? (xmin + shift) : inf;
            if (x < inf) x -= 3.6e-7f * fabsf(x) + 3.6e-7f * fabsf(shift) + 1e-37f;
            const int n = lcnt[k];
            lcnt[k] = 0;
            unsigned int *cell = reinterpret_cast<unsigned int *>(sets + t * K + k);
            if (t < t0) {
                if (t == t0 - 1) cell[0] = __builtin_bit_cast(unsigned int, d);          // the shared record: this chunk's d~
            } else if (t == t1 - 1 && !last_chunk) {                                    // ... and the chunk before's X and set
                cell[1] = __builtin_bit_cast(unsigned int, x); cell[2] = lidx[k]; cell[3] = (unsigned int)(n > JF_CAP ? JF_CAP + 1 : n);
            } else
                sets[t * K + k] = (u32x4){__builtin_bit_cast(unsigned int, d), __builtin_bit_cast(unsigned int, x), lidx[k],
                                          (unsigned int)(n > JF_CAP ? JF_CAP + 1 : n)};
        }
        if ((t & 63) == 0) {                            // uniform: a shift step publishes the wavefront's minimum
            float m = lead ? d : inf;
#pragma unroll
            for (int s = 1; s <= 8; s <<= 1) { const float o = __shfl_xor(m, s, 64); m = o < m ? o : m; }
            if (lane == 0) wmin[tid >> 6] = m;
        }
        __syncthreads();
    };

    for (int64_t t = ts; t < t1; t += NB) {
#pragma unroll
        for (int s = 0; s < NB; ++s) step(t + s, jb[s], td_raw[s], id_raw[s], sc_raw[s]);
    }
}

void launch_viterbi_lb(const int64_t *cand, const double *tdist, const float *Jlo, const float *scale, const int64_t *off,
                       int n_utts, int K, int64_t n_units, float beta, void *sets, hipStream_t s, int chunk_len, int warm, float slack32)
{
    static_assert(JF_CAP == 4, "the sets travel as one 32-bit word");
    for (int u0 = 0; u0 < n_utts; u0 += DpBatch::MAX) {
        const int n = (n_utts - u0 < DpBatch::MAX) ? n_utts - u0 : DpBatch::MAX;
        DpBatch batch;
        int64_t Tmax = 1;
        for (int i = 0; i <= n; ++i) batch.off[i] = off[u0 + i];
        for (int i = 0; i < n; ++i) Tmax = off[u0 + i + 1] - off[u0 + i] > Tmax ? off[u0 + i + 1] - off[u0 + i] : Tmax;
        const int64_t clen = chunk_len > 0 ? chunk_len : (int64_t)1 << 40;
        const unsigned n_chunks = chunk_len > 0 && Tmax > 1 ? (unsigned)((Tmax - 1 + clen - 1) / clen) : 1u;
        batch.first = u0;
        int variant, kpm;
        if (K <= 64) { variant = 0; kpm = 16; }
        else if (K <= 100) { variant = 1; kpm = 25; }
        else if (K <= 128) { variant = 2; kpm = 32; }
        else { variant = 3; kpm = 52; }
        const int KP = 4 * kpm;
        const int nth = 64 * ((K + 15) / 16);
        const size_t shmem = (size_t)2 * KP * 4 + (size_t)KP * 4 + (size_t)KP * 4 + 64;
#define SNK_LB(KPM_, NB_, NTH_)                                                                               \
    hipLaunchKernelGGL((viterbi_lb_kernel<KPM_, NB_, NTH_>), dim3(n, n_chunks), dim3(nth), shmem, s, cand, tdist, Jlo, scale, \
                       batch, K, n_units, KP, beta, reinterpret_cast<u32x4 *>(sets), clen, warm, slack32)
        if (variant == 0) SNK_LB(16, 4, 256);
        else if (variant == 1) SNK_LB(25, 6, 448);
        else if (variant == 2) SNK_LB(32, 3, 512);
        else SNK_LB(52, 1, 832);
#undef SNK_LB
    }
}

// ---------------------------------------------------------------------------------------------
// canonical float64 join cost of one pair (the oracle's order: column by column, separately rounded
// sub / mul / add on fl64(f32 * w), the values speech_manip.weight() produces)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double jf_exact_cost(const float *__restrict__ JC_unw, int Jp, int Dj,
                                                const double *__restrict__ wj, int64_t a, int64_t b)
{
    // the natural successor: unit_end_data[a] and unit_start_data[a + 1] are the SAME row of join_contexts, every
    // difference is exactly 0.0 and so is the canonical sum -- the common winner of a column needs no gather at all
    if (a + 1 == b) return 0.0;
    const f32x4 *__restrict__ re = reinterpret_cast<const f32x4 *>(JC_unw + (a + 1) * (int64_t)Jp);   // unit_end_data[a]
    const f32x4 *__restrict__ rs = reinterpret_cast<const f32x4 *>(JC_unw + b * (int64_t)Jp);         // unit_start_data[b]
    double acc = 0.0;
    const int n4 = Dj >> 2;
    int c4 = 0;
    for (; c4 + 4 <= n4; c4 += 4) {
        f32x4 e[4], s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { e[u] = re[c4 + u]; s[u] = rs[c4 + u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double w = wj[4 * (c4 + u) + i];
                const double d = __dsub_rn(__dmul_rn((double)e[u][i], w), __dmul_rn((double)s[u][i], w));
                acc = __dadd_rn(acc, __dmul_rn(d, d));
            }
    }
    for (int c = 4 * c4; c < Dj; ++c) {
        const double w = wj[c];
        const double d = __dsub_rn(__dmul_rn((double)JC_unw[(a + 1) * (int64_t)Jp + c], w),
                                   __dmul_rn((double)JC_unw[b * (int64_t)Jp + c], w));
        acc = __dadd_rn(acc, __dmul_rn(d, d));
    }
    return __dsqrt_rn(acc);
}

// ---------------------------------------------------------------------------------------------
// Tripwire of pass 1's bounds.  The bf16 form's proof (joinlb2_kernels.hip) rests on a PROBED property of
// v_mfma_f32_32x32x16_bf16 (off by at most 2^-20 of |products| + |C| per instruction); a bound that is not a bound does not
// crash -- it silently drops the optimal predecessor.  Every exact join cost this path computes anyway (pass 3: the members
// of the predecessor sets; pass 4: the refinements) is therefore held against the float32 bound pass 1 gave that cell:
//   stats[4] += cells with lo > exact  (must stay 0),
//   stats[5] (low word: order-preserving image of a float) = the smallest (exact^2 - lo^2) / (2 ceps scale^2) over the cells
//   with lo > 0 -- ceps (ne + ns) <= 2 ceps scale^2 is the error budget of the squared cost, scale the step's largest centred
//   norm: what is left of the budget, in units of (an upper bound of) the budget; negative = a violation.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int jf_trip_image(float f)
{
    const unsigned int u = __builtin_bit_cast(unsigned int, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ void jf_trip_check(double c, float lo, double unit, int &viol, float &mmin)
{
    if ((double)lo > c) ++viol;
    if (lo > 0.f && lo < __builtin_inff() && unit > 0.0) {
        const float m = (float)((c * c - (double)lo * (double)lo) / unit);
        mmin = m < mmin ? m : mmin;
    }
}
// whole wavefront (every lane active): one atomic per wavefront at most, none in the steady state
__device__ __forceinline__ void jf_trip_commit(unsigned long long *stats, int viol, float mmin)
{
#pragma unroll
    for (int m = 1; m <= 32; m <<= 1) {
        viol += __shfl_xor(viol, m, 64);
        const float o = __shfl_xor(mmin, m, 64);
        mmin = o < mmin ? o : mmin;
    }
    if ((threadIdx.x & 63) == 0) {
        if (viol) atomicAdd(&stats[4], (unsigned long long)viol);
        unsigned int *slot = reinterpret_cast<unsigned int *>(&stats[5]);
        const unsigned int im = jf_trip_image(mmin);
        if (mmin < __builtin_inff() && im < __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slot, im);
    }
}

// pass 3: one lane per (row, column): the exact costs of its recorded predecessors, and the cell's 64-byte
// record for pass 4:  { td (target cost, +inf for an unusable unit), X, dlb, c[0..3], slots | n << 32 }
struct __attribute__((aligned(16))) JfRecord { double td, x, lb, c[JF_CAP]; unsigned long long meta; };

__global__ void __launch_bounds__(256)
join_exact_sparse_kernel(const float *__restrict__ JC_unw, int Jp, int Dj, const double *__restrict__ wj,
                         int64_t n_units, const int64_t *__restrict__ cand, const double *__restrict__ tdist,
                         int64_t R, int K, const u32x4 *__restrict__ sets, JfRecord *__restrict__ rec,
                         const float *__restrict__ Jlo, const float *__restrict__ scale, float ceps,
                         unsigned long long *__restrict__ stats)
{
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= R * K) return;
    const u32x4 st = sets[cell];
    const int n_raw = (int)st[3];
    const int n = n_raw > JF_CAP ? JF_CAP : n_raw;       // an overflowed set: its first members still start pass 4's refinement
    const int64_t t = cell / K;                            // n > 0 implies a predecessor row inside the utterance
    const int64_t b = cand[cell];
    const unsigned int ix = st[2];
    // (scalars first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this compiler)
    const unsigned int w_d = st[0], w_x = st[1];
    const double lb64 = (double)__builtin_bit_cast(float, w_d), x64 = (double)__builtin_bit_cast(float, w_x);
    double c[JF_CAP];
#pragma unroll
    for (int j = 0; j < JF_CAP; ++j) {
        c[j] = __builtin_inf();
        if (j < n) {
            const int64_t a = cand[(t - 1) * K + ((ix >> (8 * j)) & 0xffu)];
            if (jf_usable(a, n_units) && jf_usable(b, n_units)) c[j] = jf_exact_cost(JC_unw, Jp, Dj, wj, a, b);
        }
    }
    // four 16-byte pieces, as pass 4's loader streams them
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2 *dst = reinterpret_cast<f64x2 *>(rec + cell);
    dst[0] = (f64x2){jf_usable(b, n_units) ? tdist[cell] : __builtin_inf(), x64};
    dst[1] = (f64x2){lb64, c[0]};
    dst[2] = (f64x2){c[1], c[2]};
    dst[3] = (f64x2){c[3], __builtin_bit_cast(double, (unsigned long long)ix | ((unsigned long long)n_raw << 32))};
    if (stats && n > 0) {                                     // tripwire (above); this form is the A/B partner: plain atomics
        const float *slab = Jlo + (t - 1) * (int64_t)K * K + (cell - t * K);
        const double sc = (double)scale[t - 1], unit = 2.0 * (double)ceps * sc * sc;
        int viol = 0;
        float mmin = __builtin_inff();
        for (int j = 0; j < n; ++j)
            if (c[j] < __builtin_inf()) jf_trip_check(c[j], slab[(int64_t)((ix >> (8 * j)) & 0xffu) * K], unit, viol, mmin);
        if (viol) atomicAdd(&stats[4], (unsigned long long)viol);
        if (mmin < __builtin_inff()) atomicMin(reinterpret_cast<unsigned int *>(&stats[5]), jf_trip_image(mmin));
    }
}

// pass 3, cooperative form (the default; g_exact_form 0 selects the kernel above): the same records, the same canonical
// costs.  The kernel above gives a lane a cell and walks its rows with 16-byte loads: every load instruction of a wavefront
// touches 64 different rows, and the address unit serves one line per cycle -- 0.40 ms per 9 600 rows at B*, address-bound
// with half the lanes idle (a cell has one to four predecessors).  Here a workgroup (two wavefronts, 128 cells) first lists
// its REAL costs (predecessor usable, not the natural successor -- whose cost is exactly 0 without a look at the rows), then
// takes them 128 at a time, one cost per lane: the rows of a wavefront's 64 costs come in 16-column chunks by loads in which
// four neighbouring lanes read 64 consecutive bytes of ONE row (16 rows per instruction instead of 64), go through a
// wavefront-private LDS tile (pitch 20 floats: conflict-free both ways) and come back to the cost's own lane for the
// canonical column-by-column sum; the next chunk's loads are in flight meanwhile.
#define JX_T 128               // threads = cells per workgroup
#define JX_PITCH 20            // floats per row of the transposition tile (16 + 4: bank-conflict-free 16-byte accesses)
__global__ void __launch_bounds__(JX_T)
join_exact_sparse2_kernel(const float *__restrict__ JC_unw, int Jp, int Dj, const double *__restrict__ wj,
                          int64_t n_units, const int64_t *__restrict__ cand, const double *__restrict__ tdist,
                          int64_t R, int K, const u32x4 *__restrict__ sets, JfRecord *__restrict__ rec,
                          const float *__restrict__ Jlo, const float *__restrict__ scale, float ceps,
                          unsigned long long *__restrict__ stats)
{
    __shared__ __align__(16) float tile[JX_T / 64][2][64 * JX_PITCH];        // [wavefront][E, S][cost][column of the chunk]
    __shared__ double c_s[JX_T][JF_CAP];
    __shared__ int a_s[JX_T][JF_CAP], b_s[JX_T];
    __shared__ unsigned short item_s[JX_T * JF_CAP];
    __shared__ int wtot[JX_T / 64], total_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t cell = (int64_t)blockIdx.x * JX_T + tid;
    const bool live = cell < R * K;
    // ---- this cell: set, predecessors, which of its costs need the rows ----
    u32x4 st = {0u, 0u, 0u, 0u};
    int64_t b = -1;
    if (live) { st = sets[cell]; b = cand[cell]; }
    const int n_raw = (int)st[3];
    const int n = n_raw > JF_CAP ? JF_CAP : n_raw;
    const int64_t t = live ? cell / K : 0;
    const unsigned int ix = st[2];
    const bool okb = jf_usable(b, n_units);
    int mine = 0;
    unsigned int need = 0u;
#pragma unroll
    for (int j = 0; j < JF_CAP; ++j) {
        double c = __builtin_inf();
        if (j < n) {
            const int64_t a = cand[(t - 1) * K + ((ix >> (8 * j)) & 0xffu)];
            if (jf_usable(a, n_units) && okb) {
                if (a + 1 == b) c = 0.0;             // the natural successor: the same row of join_contexts on both sides
                else { need |= 1u << j; ++mine; a_s[tid][j] = (int)a; }
            }
        }
        c_s[tid][j] = c;
    }
    b_s[tid] = (int)b;
    // ---- the workgroup's list of costs (prefix sum over the threads) ----
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < wv; ++w) base += wtot[w];
    if (tid == JX_T - 1) total_s = base + mine;
#pragma unroll
    for (int j = 0; j < JF_CAP; ++j)
        if (need & (1u << j)) item_s[base++] = (unsigned short)((tid << 2) | j);
    __syncthreads();
    const int total = total_s;
    if (stats && stats[8]) {                                 // what the kernel's roofline is priced on (snk_get_info sparse_exact_costs / sparse_set_members): option roofline_counters only -- same-address atomics of 7 500 workgroups cost 0.4 ms per launch
        if (tid == 0 && total) atomicAdd(&stats[6], (unsigned long long)total);
        int mem = n;
#pragma unroll
        for (int m = 1; m <= 32; m <<= 1) mem += __shfl_xor(mem, m, 64);
        if (lane == 0 && mem) atomicAdd(&stats[7], (unsigned long long)mem);
    }
    const int n_chunks = (Dj + 15) / 16;
    const int g4 = lane >> 2, q = lane & 3;                 // loader role: rows 4 g4 + i, columns 4 q .. 4 q + 3 of the chunk
    float *const tE = tile[wv][0], *const tS = tile[wv][1];
    for (int r0 = 0; r0 < total; r0 += JX_T) {
        // (a wavefront loads and sums for its own 64 items: one without items has nothing to do)
        if (r0 + 64 * wv >= total) continue;
        // this lane's cost
        const int it = r0 + tid;
        const bool have = it < total;
        const unsigned int code = have ? item_s[it] : 0u;
        // rows of the four costs this lane loads for: items r0 + 64 wv + 4 g4 + i
        const float *pe[4], *ps[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int it2 = r0 + 64 * wv + 4 * g4 + i;
            const unsigned int cd = it2 < total ? item_s[it2] : 0u;
            const int64_t a2 = it2 < total ? a_s[cd >> 2][cd & 3u] : 0, b2 = it2 < total ? b_s[cd >> 2] : 0;
            pe[i] = JC_unw + (a2 + 1) * (int64_t)Jp;        // unit_end_data[a]   = JC[a+1]
            ps[i] = JC_unw + b2 * (int64_t)Jp;              // unit_start_data[b] = JC[b]
        }
        f32x4 ge[4], gs[4];
        auto fetch = [&](int ch) {
            int c0 = ch * 16 + 4 * q;
            if (c0 > Jp - 4) c0 = Jp - 4;                    // a quad beyond the row: any readable address (its columns are not used)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ge[i] = *reinterpret_cast<const f32x4 *>(pe[i] + c0);
                gs[i] = *reinterpret_cast<const f32x4 *>(ps[i] + c0);
            }
        };
        fetch(0);
        double acc = 0.0;
        for (int ch = 0; ch < n_chunks; ++ch) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4 *>(tE + (4 * g4 + i) * JX_PITCH + 4 * q) = ge[i];
                *reinterpret_cast<f32x4 *>(tS + (4 * g4 + i) * JX_PITCH + 4 * q) = gs[i];
            }
            if (ch + 1 < n_chunks) fetch(ch + 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                 // the tile is this wavefront's own: LDS operations of a wavefront execute in order
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            f32x4 e[4], sv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                e[u] = *reinterpret_cast<const f32x4 *>(tE + lane * JX_PITCH + 4 * u);
                sv[u] = *reinterpret_cast<const f32x4 *>(tS + lane * JX_PITCH + 4 * u);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                 // ... read before the next chunk is written
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int c0 = ch * 16;
            if (c0 + 16 <= Dj) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double w = wj[c0 + 4 * u + i];
                        const double d = __dsub_rn(__dmul_rn((double)e[u][i], w), __dmul_rn((double)sv[u][i], w));
                        acc = __dadd_rn(acc, __dmul_rn(d, d));
                    }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (c0 + 4 * u + i < Dj) {           // uniform
                            const double w = wj[c0 + 4 * u + i];
                            const double d = __dsub_rn(__dmul_rn((double)e[u][i], w), __dmul_rn((double)sv[u][i], w));
                            acc = __dadd_rn(acc, __dmul_rn(d, d));
                        }
            }
        }
        if (have) c_s[code >> 2][code & 3u] = __dsqrt_rn(acc);
    }
    __syncthreads();
    if (live) {
        const unsigned int w_d = st[0], w_x = st[1];
        const double lb64 = (double)__builtin_bit_cast(float, w_d), x64 = (double)__builtin_bit_cast(float, w_x);
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        f64x2 *dst = reinterpret_cast<f64x2 *>(rec + cell);
        dst[0] = (f64x2){okb ? tdist[cell] : __builtin_inf(), x64};
        dst[1] = (f64x2){lb64, c_s[tid][0]};
        dst[2] = (f64x2){c_s[tid][1], c_s[tid][2]};
        dst[3] = (f64x2){c_s[tid][3], __builtin_bit_cast(double, (unsigned long long)ix | ((unsigned long long)n_raw << 32))};
    }
    if (stats) {                                              // tripwire of pass 1's bounds (above): the costs of this cell against them
        int viol = 0;
        float mmin = __builtin_inff();
        if (live && n > 0) {
            const float *slab = Jlo + (t - 1) * (int64_t)K * K + (cell - t * K);
            const double sc = (double)scale[t - 1], unit = 2.0 * (double)ceps * sc * sc;
            float lo[JF_CAP];
#pragma unroll
            for (int j = 0; j < JF_CAP; ++j) lo[j] = j < n ? slab[(int64_t)((ix >> (8 * j)) & 0xffu) * K] : 0.f;
#pragma unroll
            for (int j = 0; j < JF_CAP; ++j)
                if (j < n && c_s[tid][j] < __builtin_inf()) jf_trip_check(c_s[tid][j], lo[j], unit, viol, mmin);
        }
        jf_trip_commit(stats, viol, mmin);
    }
}

// test hook of the bounds' tripwire (option join_lb_test_scale): pass 1's bounds multiplied by a factor > 1 are no bounds any more
__global__ void jf_scale_kernel(float *__restrict__ x, int64_t n, float f)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= f;
}
void launch_scale_f32(float *x, int64_t n, float f, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(jf_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, f);
}

static int g_exact_form = 1;
void set_join_exact_form(int f) { g_exact_form = f ? 1 : 0; }

size_t join_record_bytes() { return sizeof(JfRecord); }

void launch_join_exact_sparse(const float *JC_unw, int Jp, int Dj, const double *wj, int64_t n_units, const int64_t *cand,
                              const double *tdist, int64_t R, int K, const void *sets, void *rec, hipStream_t s,
                              const float *Jlo, const float *scale, float ceps, unsigned long long *stats)
{
    if (!Jlo || !scale) stats = nullptr;
    static_assert(sizeof(JfRecord) == 64, "pass 4 streams 64-byte records");
    const int64_t cells = R * K;
    if (g_exact_form == 1 && n_units < ((int64_t)1 << 31) && K <= 256) {
        hipLaunchKernelGGL(join_exact_sparse2_kernel, dim3((unsigned)((cells + JX_T - 1) / JX_T)), dim3(JX_T), 0, s, JC_unw, Jp, Dj,
                           wj, n_units, cand, tdist, R, K, reinterpret_cast<const u32x4 *>(sets), reinterpret_cast<JfRecord *>(rec),
                           Jlo, scale, ceps, stats);
        return;
    }
    hipLaunchKernelGGL(join_exact_sparse_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, JC_unw, Jp, Dj,
                       wj, n_units, cand, tdist, R, K, reinterpret_cast<const u32x4 *>(sets), reinterpret_cast<JfRecord *>(rec),
                       Jlo, scale, ceps, stats);
}

// ---------------------------------------------------------------------------------------------
// pass 4: the exact recursion over the recorded sets.  One thread per column, ONE barrier per step, no
// global load in the step: the records of the next steps come through an LDS ring filled by a LOADER
// wavefront (the last one of the workgroup; batches of BS steps, issued a batch ahead, written to the
// ring at the end of the batch before).  Every (t,k) is verified (file header); one that cannot be is
// REFINED: with the exact delta_{t-1} at hand only predecessors with delta[k'] + clo(k',k) <= best so
// far can win or tie -- those (a handful) get their exact cost, one lane each.
// stats[0] += cells refined, [1] += steps with a refinement, [2] += exact costs computed there,
// [3] += cells whose set had overflowed.
// ---------------------------------------------------------------------------------------------
// wavefront minimum of a double: rotations inside the 16-lane rows on the DPP path, then the four row
// results through scalar registers (six ds_bpermute round trips otherwise: the longest piece of a step)
template <int CTRL>
__device__ __forceinline__ double jf_dpp_f64(double x)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    lo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
    hi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
__device__ __forceinline__ double jf_readlane_f64(double x, int l)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
__device__ __forceinline__ double jf_wave_min(double x)
{
    double o;
    o = jf_dpp_f64<0x121>(x); x = o < x ? o : x;       // row_ror:1
    o = jf_dpp_f64<0x122>(x); x = o < x ? o : x;       // row_ror:2
    o = jf_dpp_f64<0x124>(x); x = o < x ? o : x;       // row_ror:4
    o = jf_dpp_f64<0x128>(x); x = o < x ? o : x;       // row_ror:8
    const double a = jf_readlane_f64(x, 0), b = jf_readlane_f64(x, 16), c = jf_readlane_f64(x, 32), d = jf_readlane_f64(x, 48);
    const double ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// -DSNK_JF_TRACE: a debug build whose pass 4 stamps the shader clock at six points of each of the first 1024 steps
// (wavefront 0 and the loader) behind the statistics words; launch_viterbi_sparse prints the averages.
#ifdef SNK_JF_TRACE
#define JF_STAMP(i) do { if (stats && lane == 0 && t < 1024 && (wave == 0 || loader)) stats[128 + (size_t)t * 16 + (loader ? 8 : 0) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define JF_STAMP(i) do { } while (0)
#endif

template <int BS, bool BPL>
__global__ void __launch_bounds__(320)
viterbi_sparse_kernel(const int64_t *__restrict__ cand_all, const JfRecord *__restrict__ rec_all,
                      const float *__restrict__ Jlo_all, const float *__restrict__ JC_unw, int Jp, int Dj,
                      const double *__restrict__ wj, const DpBatch batch, int K, int64_t n_units,
                      unsigned char *__restrict__ bp_all, int64_t *__restrict__ path_all,
                      int64_t *__restrict__ path_len_all, double *__restrict__ cost_all,
                      unsigned long long *__restrict__ stats, const float *__restrict__ scale_all, float ceps)
{
    __builtin_amdgcn_s_setprio(3);           // as viterbi_lb_kernel
    const int64_t r0 = batch.off[blockIdx.x];
    const int64_t T = batch.off[blockIdx.x + 1] - r0;
    const int64_t *__restrict__ cand = cand_all + r0 * K;
    const JfRecord *__restrict__ rec = rec_all + r0 * K;
    const float *__restrict__ Jlo = Jlo_all + r0 * K * K;
    const float *__restrict__ scale = scale_all + r0;
    unsigned char *__restrict__ bp_global = bp_all + r0 * K;
    int64_t *__restrict__ path = path_all + r0;
    int64_t *__restrict__ path_len = path_len_all + batch.first + blockIdx.x;
    double *__restrict__ cost = cost_all + batch.first + blockIdx.x;

    extern __shared__ __align__(16) unsigned char smem[];
    double *delta = reinterpret_cast<double *>(smem);              // [2][256]
    double *rbest = delta + 2 * 256;                               // [256] refinement: a failing column's best total
    double *wrec = rbest + 256;                                    // [2][4] x {minimum of delta - dlb, any column failed} per wavefront
    int *failw = reinterpret_cast<int *>(wrec + 16);               // [4] scratch; [3] the final slot
    int *rarg = failw + 4;                                         // [256] ... and its predecessor slot
    unsigned char *fail_s = reinterpret_cast<unsigned char *>(rarg + 256);   // [256]
    int *final_slot_p = failw + 3;                                 // (all LDS in one array: a second object can de-pipeline the loader)
    u32x4 *ring = reinterpret_cast<u32x4 *>(fail_s + 256);         // [3][slot_pieces] records of three batches (4 x 16 bytes per cell)
    const int pieces = BS * K * 4;                                 // 16-byte pieces of one batch
    const int slot_pieces = (pieces + 63) & ~63;                   // ring slots are whole 1-KB LDS-DMA writes
    unsigned char *bp_lds = reinterpret_cast<unsigned char *>(ring + (size_t)3 * slot_pieces);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = (int)blockDim.x >> 6, ncw = nwaves - 1;     // compute wavefronts; the last one loads
    const bool loader = wave == ncw;
    const int k = tid;
    const bool col = !loader && k < K;
    const double inf = __builtin_inf();

    if (T < 2) {                      // the reference's J has no states for T < 2
        if (tid == 0) { *path_len = 0; *cost = inf; }
        return;
    }
    for (int i = tid; i < 2 * 256; i += (int)blockDim.x) delta[i] = inf;
    if (tid < 4) failw[tid] = 0;
    if (tid < 16) wrec[tid] = 0.0;                                 // off_0 = 0: delta_0 == dlb_0
    __syncthreads();
    if (col) delta[k] = rec[k].td;                                 // td of row 0 (+inf for an unusable unit)
    // Loader: the records of batch j (steps j BS + 1 .. j BS + BS) are consecutive in global memory and go to
    // ring slot j % 3 by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes to M0 + 16 lane, no registers).
    // Inline asm: the compiler then neither counts these loads nor drains them at barriers and LDS reads;
    // the loader waits for them itself (vmcnt(0) at the last step of a batch, with ONE batch outstanding).
    const u32x4 *const rsrc = reinterpret_cast<const u32x4 *>(rec);
    const int64_t total_pieces = T * (int64_t)K * 4;
    const unsigned ring_lds = (unsigned)(size_t)ring;              // LDS byte offset of the ring
    auto dma_batch = [&](int64_t j) {
        const int64_t base = (j * BS + 1) * (int64_t)K * 4;
        const unsigned dst0 = ring_lds + (unsigned)(j % 3) * (unsigned)slot_pieces * 16u;
        for (int i = 0; i < slot_pieces; i += 64) {
            int64_t pc = base + i + lane;
            if (i + lane >= pieces || pc >= total_pieces) pc = 0;  // padding of the slot / beyond the utterance: never read
            const u32x4 *src = rsrc + pc;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)i * 16u);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
        }
    };
    if (loader) {
        dma_batch(0); dma_batch(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    double off = 0.0;                                              // min over the columns of delta_{t-1} - d~_{t-1}
    // step counters kept incrementally in 32 bits (a wavefront issues about one instruction per four cycles
    // here: the step is as long as its instruction count)
    const int Ti = (int)T, K4 = K * 4;
    int bidx = 0, sin = 0, ring_slot = 0;                          // batch of the step, step inside it, ring slot of the batch
    int rp_off = (col ? k : 0) * 4;                                // this column's record of the step, in 16-byte pieces
    int bp_off = K + k;
    for (int t = 1; t < Ti; ++t) {
        double best = inf, d = inf, diff = inf, td = inf, lbv = inf;
        int arg = 0;
        bool fail = false;
        const double *dprev = delta + ((t - 1) & 1) * 256;
        double *dcur = delta + (t & 1) * 256;
        double *wr = wrec + (t & 1) * 8;
        JF_STAMP(0);
        if (loader) {
            if (sin == BS - 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // batch bidx + 1 has landed (issued BS steps ago)
                dma_batch((int64_t)bidx + 2);                      // into the slot batch bidx - 1 has left
            }
            JF_STAMP(1);
        } else {
            // this step's record
            const u32x4 *rp = ring + rp_off;
            const u32x4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
            auto f64 = [](unsigned int lo, unsigned int hi) { return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32)); };
            td = f64(q0[0], q0[1]);
            const double xv = f64(q0[2], q0[3]);
            lbv = f64(q1[0], q1[1]);
            double c[JF_CAP];
            c[0] = f64(q1[2], q1[3]);
            c[1] = f64(q2[0], q2[1]);
            c[2] = f64(q2[2], q2[3]);
            c[3] = f64(q3[0], q3[1]);
            const unsigned int ix = q3[2];
#ifdef SNK_JF_TRACE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            JF_STAMP(1);
#endif
            const int n_raw = (int)(q3[3] & 0xffu);
            const int n = n_raw > JF_CAP ? JF_CAP : n_raw;
#pragma unroll
            for (int j = 0; j < JF_CAP; ++j) {                     // predicated, no branches
                const int p = (int)((ix >> (8 * j)) & 0xffu);
                double tot = __dadd_rn(dprev[p], c[j]);
                tot = (j < n) ? tot : inf;
                const bool take = tot < best || (tot == best && p < arg && j < n);
                best = take ? tot : best;
                arg = take ? p : arg;
            }
#ifdef SNK_JF_TRACE
            asm volatile("" : "+v"(best));
            JF_STAMP(2);
#endif
            // proof that no predecessor outside the set wins or ties (strict, with room for the roundings)
            const bool usable = td < inf;
            bool ok = true;
            if (n_raw > JF_CAP) ok = false;
            else if (xv < inf) {
                const double bound = xv + off - 1e-12 * (fabs(xv) + fabs(off));
                ok = (off < inf) && (bound > best);
            }   // xv == inf: every predecessor with a finite lower-bound total is in the set
            fail = col && usable && !ok;
            if (col) fail_s[k] = fail ? 1 : 0;
            d = (col && usable) ? __dadd_rn(td, best) : inf;
            if (col) dcur[k] = d;
            diff = (col && d < inf) ? __dsub_rn(d, lbv) : inf;
            const double wm = jf_wave_min(diff);
            const bool wfail = __ballot(fail) != 0ull;
            if (lane == 0) { wr[2 * wave] = wm; wr[2 * wave + 1] = wfail ? 1.0 : 0.0; }
            JF_STAMP(3);
        }
        __syncthreads();
        JF_STAMP(4);
        // four compute wavefronts, always (launch_viterbi_sparse): { minimum, failed } x 4
        const double o0 = wr[0], f0 = wr[1], o1 = wr[2], f1 = wr[3], o2 = wr[4], f2 = wr[5], o3 = wr[6], f3 = wr[7];
        const double o01 = o0 < o1 ? o0 : o1, o23 = o2 < o3 ? o2 : o3;
        off = o01 < o23 ? o01 : o23;
        const bool anyfail = (f0 + f1 + f2 + f3) != 0.0;
        if (anyfail) {                                             // rare; every wavefront takes the same barriers
            if (fail) { rbest[k] = best; rarg[k] = arg; }
            __syncthreads();
            int n_failed = 0, n_exact = 0, tviol = 0;
            float tmin = __builtin_inff();
            const double tsc = (double)scale[t - 1], tunit = 2.0 * (double)ceps * tsc * tsc;
            if (!loader) {
                for (int kf = 0; kf < K; ++kf) {
                    if (!fail_s[kf]) continue;                         // uniform (LDS)
                    if ((n_failed++ % ncw) != wave) continue;          // wave-uniform
                    const int64_t b = cand[t * K + kf];
                    const double bS = rbest[kf];
                    double lb = bS;
                    int la = rarg[kf];
                    const float *jcol = Jlo + (t - 1) * (int64_t)K * K + kf;
                    for (int p = lane; p < K; p += 64) {
                        const double dp = dprev[p];
                        const float lo = jcol[(int64_t)p * K];
                        if (dp < inf && lo < __builtin_inff() && __dadd_rn(dp, (double)lo) <= bS) {
                            const int64_t a = cand[(t - 1) * K + p];
                            const double cex = jf_exact_cost(JC_unw, Jp, Dj, wj, a, b);
                            const double tot = __dadd_rn(dp, cex);
                            if (stats) jf_trip_check(cex, lo, tunit, tviol, tmin);        // tripwire of pass 1's bounds
                            ++n_exact;
                            if (tot < lb || (tot == lb && p < la)) { lb = tot; la = p; }
                        }
                    }
#pragma unroll
                    for (int m = 1; m <= 32; m <<= 1) {
                        const double ob = __shfl_xor(lb, m, 64);
                        const int oa = __shfl_xor(la, m, 64);
                        if (ob < lb || (ob == lb && oa < la)) { lb = ob; la = oa; }
                    }
                    if (lane == 0) { rbest[kf] = lb; rarg[kf] = la; }
                }
            }
            __syncthreads();
            if (!loader) {
                if (fail) {
                    best = rbest[k]; arg = rarg[k];
                    d = __dadd_rn(td, best);
                    dcur[k] = d;
                    diff = d < inf ? __dsub_rn(d, lbv) : inf;
                }
                const double wm = jf_wave_min(diff);
                if (lane == 0) wr[2 * wave] = wm;
                if (stats) {
                    n_exact += __shfl_xor(n_exact, 1, 64); n_exact += __shfl_xor(n_exact, 2, 64); n_exact += __shfl_xor(n_exact, 4, 64);
                    n_exact += __shfl_xor(n_exact, 8, 64); n_exact += __shfl_xor(n_exact, 16, 64); n_exact += __shfl_xor(n_exact, 32, 64);
                    if (lane == 0 && n_exact) atomicAdd(&stats[2], (unsigned long long)n_exact);
                    if (tid == 0) { atomicAdd(&stats[0], (unsigned long long)n_failed); atomicAdd(&stats[1], 1ull); }
                    jf_trip_commit(stats, tviol, tmin);
                }
            }
            __syncthreads();
            off = inf;
            for (int w = 0; w < ncw; ++w) { const double o = wr[2 * w]; off = o < off ? o : off; }
        }
        if (col) {
            if constexpr (BPL) bp_lds[bp_off] = (unsigned char)arg;
            else bp_global[bp_off] = (unsigned char)arg;
        }
        JF_STAMP(5);
        bp_off += K;
        rp_off += K4;
        if (++sin == BS) {
            sin = 0; ++bidx;
            ring_slot = ring_slot == 2 ? 0 : ring_slot + 1;
            rp_off = ring_slot * slot_pieces + (col ? k : 0) * 4;
        }
    }
    __syncthreads();

    const double *dlast = delta + ((T - 1) & 1) * 256;
    if (tid == 0) {
        double best = inf;
        int slot = 0;
        for (int kk = 0; kk < K; ++kk)
            if (dlast[kk] < best) { best = dlast[kk]; slot = kk; }
        if (best == inf) { *path_len = 0; *cost = inf; *final_slot_p = -1; }
        else { *path_len = T; *cost = best; *final_slot_p = slot; }
    }
    if (!BPL) __threadfence();
    __syncthreads();
    // back-trace: one thread walks the back-pointers (LDS when they fit), then ALL threads fetch the unit ids
    // in parallel (one dependent global load per step would cost a memory round trip each)
    if (*final_slot_p >= 0) {
        if constexpr (BPL) {
            unsigned char *slots = reinterpret_cast<unsigned char *>(ring);       // the ring is free now (T <= its size is checked by the launcher)
            if (tid == 0) {
                int slot = *final_slot_p;
                for (int64_t t = T - 1; t >= 0; --t) {
                    slots[t] = (unsigned char)slot;
                    if (t > 0) slot = bp_lds[t * K + slot];
                }
            }
            __syncthreads();
            for (int64_t t = tid; t < T; t += (int)blockDim.x) path[t] = cand[t * K + slots[t]];
        } else if (tid == 0) {
            int slot = *final_slot_p;
            for (int64_t t = T - 1; t >= 0; --t) {
                path[t] = cand[t * K + slot];
                if (t > 0) slot = bp_global[t * K + slot];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pass 4, one compute wavefront per utterance (the form launch_viterbi_sparse runs; the four-wavefront kernel above is
// its A/B partner, option viterbi_sparse_waves = 4).  A step of the recursion is a chain of dependent instructions: what
// it costs is the length of that chain, and with four wavefronts a third of it was the exchange between them (a barrier
// and an LDS round trip for `off` and the failure flag; the loader's 25 LDS-DMA instructions, issued before the barrier
// of every fourth step, on top).  Here lane l owns the columns l, l + 64, ... (NC of them):
//   * delta_{t-1} lives in LDS, written and read by the same wavefront (LDS operations of a wavefront execute in order:
//     no barrier inside a batch),
//   * `off` is a DPP minimum over float32 images rounded to below (a LOWER bound of the exact minimum: the proof only
//     gets stricter, by 1.2e-7 of |off|),
//   * the loader wavefront meets the compute wavefront at ONE barrier per batch of steps and issues its LDS-DMA loads
//     right after it, beside the batch's steps; batches are NL KB each, two of them in flight (s_waitcnt vmcnt(NL):
//     loads return in order),
//   * a failing cell's candidates get their exact cost from the whole wavefront: squares in parallel (one column of
//     the join vectors per lane), the canonical ordered sum read back from LDS by every lane.
// Same arithmetic and the same decisions as the kernel above (path, cost and back-pointers bit for bit); the counters may
// differ by a few exact costs (a column's later candidates are tested against its best total SO FAR, and the bound
// `off` is a hair lower).
// ---------------------------------------------------------------------------------------------
#define JF1_NL 16              // 1-KB LDS-DMA loads per batch (a ring slot is JF1_NL KB)
#define JF1_SQ 4096            // doubles: the squares of JF1_SQ / (Dj + 2) exact costs at a time (>= JF_MAXD + 3: one at least)
#define JF1_PAIRS 512          // pairs listed before their exact costs are taken

__device__ __forceinline__ unsigned int jf_image(float f)       // order-preserving image of a float (no NaN)
{
    const unsigned int u = __builtin_bit_cast(unsigned int, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float jf_unimage(unsigned int u)
{
    return __builtin_bit_cast(float, (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ unsigned int jf_wave_min_u32(unsigned int v)
{
#define JF_DPP(ctrl_, rmask_)                                                                         \
    {                                                                                                 \
        const unsigned int o = (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl_, rmask_, 0xf, false); \
        v = o < v ? o : v;                                                                            \
    }
    JF_DPP(0xB1, 0xf)        // quad_perm [1,0,3,2]
    JF_DPP(0x4E, 0xf)        // quad_perm [2,3,0,1]
    JF_DPP(0x141, 0xf)       // row_half_mirror
    JF_DPP(0x140, 0xf)       // row_mirror
    JF_DPP(0x142, 0xa)       // row_bcast:15 into rows 1 and 3
    JF_DPP(0x143, 0xc)       // row_bcast:31 into rows 2 and 3
#undef JF_DPP
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}

// FST32 (option viterbi_weights 1): the reference's own arithmetic -- OpenFST's float32 tropical weights (viterbi_kernels.hip,
// fst_functions_wrapped.py:47,201,368,389): acc_t[k] = min_k' fl32(acc_{t-1}[k'] + fl32(fl32(td[t-1,k']) + fl32(c(k',k)))), the last row's
// target cost on the exit arc.  In exact arithmetic acc + td IS the float64 recursion's delta, so passes 1-3 serve unchanged (pass 2's
// d~ needs no property of its own) and only the proof changes: float32 rounding is monotone and clo <= c is a float32 value, so an
// excluded predecessor's total is at least fl32(acc + fl32(td + clo)) >= (acc + td + clo)(1 - 2^-24)^2 >= (off' + X)(1 - 2^-23) with
// off' = min (acc + td - d~); the refinement tests fl32(acc + fl32(td + clo)) <= best, exact by the same monotonicity.  delta holds the
// float32 totals (widened), tdp the float32 target costs of the previous row; same tie rule.
template <int NC, bool BPL, bool FST32>
__global__ void __launch_bounds__(128)
viterbi_sparse1_kernel(const int64_t *__restrict__ cand_all, const JfRecord *__restrict__ rec_all,
                       const float *__restrict__ Jlo_all, const float *__restrict__ JC_unw, int Jp, int Dj,
                       const double *__restrict__ wj, const DpBatch batch, int K, int BS, int64_t n_units,
                       unsigned char *__restrict__ bp_all, int64_t *__restrict__ path_all,
                       int64_t *__restrict__ path_len_all, double *__restrict__ cost_all,
                       unsigned long long *__restrict__ stats, const float *__restrict__ scale_all, float ceps)
{
    __builtin_amdgcn_s_setprio(3);           // as viterbi_lb_kernel
    const int64_t r0 = batch.off[blockIdx.x];
    const int64_t T = batch.off[blockIdx.x + 1] - r0;
    const int64_t *__restrict__ cand = cand_all + r0 * K;
    const JfRecord *__restrict__ rec = rec_all + r0 * K;
    const float *__restrict__ Jlo = Jlo_all + r0 * K * K;
    const float *__restrict__ scale = scale_all + r0;
    unsigned char *__restrict__ bp_global = bp_all + r0 * K;
    int64_t *__restrict__ path = path_all + r0;
    int64_t *__restrict__ path_len = path_len_all + batch.first + blockIdx.x;
    double *__restrict__ cost = cost_all + batch.first + blockIdx.x;

    extern __shared__ __align__(16) unsigned char smem[];
    double *delta = reinterpret_cast<double *>(smem);              // [2][256]
    double *rbest = delta + 2 * 256;                               // [256] refinement: a failing column's best total
    double *sq = rbest + 256;                                      // [JF1_SQ] squares of a round of exact costs
    int *rarg = reinterpret_cast<int *>(sq + JF1_SQ);              // [256] ... and its predecessor slot
    int *final_slot_p = rarg + 256;                                // [4]
    unsigned short *plist = reinterpret_cast<unsigned short *>(final_slot_p + 4);   // [JF1_PAIRS] (failing column << 8) | predecessor
    unsigned char *flist = reinterpret_cast<unsigned char *>(plist + JF1_PAIRS);    // [256] the failing columns of a step
    float *tdp = reinterpret_cast<float *>(flist + 256);           // [2][256] FST32: float32 target costs of a row
    u32x4 *ring = reinterpret_cast<u32x4 *>(tdp + 2 * 256);        // [3][JF1_NL * 64] records of three batches (4 x 16 bytes per cell)
    const int slot_pieces = JF1_NL * 64;
    unsigned char *bp_lds = reinterpret_cast<unsigned char *>(ring + (size_t)3 * slot_pieces);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave == 1;
    const double inf = __builtin_inf();
    int kc[NC];
    bool col[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { kc[c] = lane + 64 * c; col[c] = !loader && kc[c] < K; }

    if (T < 2) {                      // the reference's J has no states for T < 2
        if (tid == 0) { *path_len = 0; *cost = inf; }
        return;
    }
    for (int i = tid; i < 2 * 256; i += (int)blockDim.x) { delta[i] = inf; tdp[i] = 0.f; }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (col[c]) {
            const double td0 = rec[kc[c]].td;                                   // td of row 0 (+inf for an unusable unit)
            if constexpr (FST32) { delta[kc[c]] = td0 < inf ? 0.0 : inf; tdp[kc[c]] = (float)td0; }
            else delta[kc[c]] = td0;
        }
    // Loader: the records of batch j (steps j BS + 1 .. j BS + BS) are consecutive in global memory and go to ring slot
    // j % 3 by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes to M0 + 16 lane, no registers), always JF1_NL
    // loads (the tail of a slot beyond BS K cells is never read).  Inline asm: the compiler neither counts these loads
    // nor drains them; the loader waits for them itself.
    const u32x4 *const rsrc = reinterpret_cast<const u32x4 *>(rec);
    const int64_t total_pieces = T * (int64_t)K * 4;
    const unsigned ring_lds = (unsigned)(size_t)ring;              // LDS byte offset of the ring
    const int pieces = BS * K * 4;                                 // 16-byte pieces of one batch (<= slot_pieces: the launcher)
    auto dma_batch = [&](int j) {
        const int64_t base = ((int64_t)j * BS + 1) * (int64_t)K * 4;
        const unsigned dst0 = ring_lds + (unsigned)(j % 3) * (unsigned)slot_pieces * 16u;
#pragma unroll 4
        for (int i = 0; i < slot_pieces; i += 64) {
            int64_t pc = base + i + lane;
            if (i + lane >= pieces || pc >= total_pieces) pc = 0;  // padding of the slot / beyond the utterance: never read
            const u32x4 *src = rsrc + pc;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)i * 16u);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
        }
    };
    if (loader) {
        dma_batch(0); dma_batch(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (loader) dma_batch(2);                                      // lands beside batch 0 (slot 2 is free)

    double off = 0.0;                                              // a lower bound of min over the columns of delta_{t-1} - d~_{t-1}; off_0 = 0
    double off_base = 0.0;                                         // the last finite one
    const int Ti = (int)T, K4 = K * 4;
    int bidx = 0, sin = 0, ring_slot = 0;
    int rp_base = 0;                                               // first piece of this step's records in the ring
    int bp_off = K;
    for (int t = 1; t < Ti; ++t) {
        const bool last_of_batch = sin == BS - 1;
        if (!loader) {
            JF_STAMP(0);
            const double *dprev = delta + ((t - 1) & 1) * 256;
            double *dcur = delta + (t & 1) * 256;
            const float *tprev = tdp + ((t - 1) & 1) * 256;
            float *tcur = tdp + (t & 1) * 256;
            double best[NC], d[NC], td[NC], lbv[NC];
            int arg[NC];
            bool fail[NC];
            float dmin = __builtin_inff();
            bool anyfail_lane = false;
            const double offm = off < inf ? off - 1e-12 * fabs(off) : 0.0;
            u32x4 q[NC][4];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const u32x4 *rp = ring + rp_base + (col[c] ? kc[c] : 0) * 4;
                q[c][0] = rp[0]; q[c][1] = rp[1]; q[c][2] = rp[2]; q[c][3] = rp[3];
            }
#ifdef SNK_JF_TRACE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            JF_STAMP(1);
#endif
            auto f64 = [](unsigned int lo, unsigned int hi) { return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32)); };
            double dp[NC][JF_CAP];
            float tp[FST32 ? NC : 1][JF_CAP];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const unsigned int ix = q[c][3][2];
#pragma unroll
                for (int j = 0; j < JF_CAP; ++j) {
                    dp[c][j] = dprev[(ix >> (8 * j)) & 0xffu];
                    if constexpr (FST32) tp[c][j] = tprev[(ix >> (8 * j)) & 0xffu];
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const u32x4 q0 = q[c][0], q1 = q[c][1], q2 = q[c][2], q3 = q[c][3];
                td[c] = f64(q0[0], q0[1]);
                const double xv = f64(q0[2], q0[3]);
                lbv[c] = f64(q1[0], q1[1]);
                double cj[JF_CAP];
                cj[0] = f64(q1[2], q1[3]);
                cj[1] = f64(q2[0], q2[1]);
                cj[2] = f64(q2[2], q2[3]);
                cj[3] = f64(q3[0], q3[1]);
                const unsigned int ix = q3[2];
                const int n_raw = (int)(q3[3] & 0xffu);
                const int n = n_raw > JF_CAP ? JF_CAP : n_raw;
                // the record holds +inf costs beyond the set's n members (join_exact_sparse_kernel): their totals are +inf
                double tot[JF_CAP];
#pragma unroll
                for (int j = 0; j < JF_CAP; ++j) {
                    if constexpr (FST32) tot[j] = (double)((float)dp[c][j] + (tp[c][j] + (float)cj[j]));       // the composed arc weight, then the path weight
                    else tot[j] = __dadd_rn(dp[c][j], cj[j]);
                }
                const double bst = __builtin_fmin(__builtin_fmin(tot[0], tot[1]), __builtin_fmin(tot[2], tot[3]));
                unsigned int ag = 0x7fffffffu;                         // among equal totals the lowest slot
#pragma unroll
                for (int j = 0; j < JF_CAP; ++j) {
                    const unsigned int pj = (tot[j] == bst) ? ((ix >> (8 * j)) & 0xffu) : 0x7fffffffu;
                    ag = pj < ag ? pj : ag;
                }
                ag = bst < inf ? ag : 0u;
                // proof that no predecessor outside the set wins or ties (strict, with room for the roundings)
                const bool usable = td[c] < inf;
                // (FST32: two float32 roundings of positive sums sit between an excluded predecessor's total and off' + X)
                const double bound = FST32 ? __builtin_fma(-2.4e-7, fabs(xv + offm), xv + offm) - 1e-30
                                           : __dadd_rn(__builtin_fma(-1e-12, fabs(xv), xv), offm);
                // xv == inf: every predecessor with a finite lower-bound total is in the set
                const bool ok = (n_raw <= JF_CAP) & (!(xv < inf) | ((off < inf) & (bound > bst)));
                fail[c] = col[c] & usable & !ok;
                anyfail_lane |= fail[c];
                best[c] = bst; arg[c] = (int)ag;
                // (FST32: d is acc + td in exact arithmetic -- what pass 2's d~ follows; the stored total is best itself)
                d[c] = (col[c] & usable) ? (FST32 ? bst + (double)(float)td[c] : __dadd_rn(td[c], bst)) : inf;
            }
#ifdef SNK_JF_TRACE
            asm volatile("" : "+v"(d[0]));
            JF_STAMP(2);
#endif
            const bool anyfail = __ballot(anyfail_lane) != 0ull;
            if (__builtin_expect(anyfail, 0)) {
                // Refinement, the whole step at once: (1) the failing columns into a list; (2) every (failing column,
                // predecessor) pair tested against the column's best total so far (delta[k'] + clo(k',k) <= best: a
                // handful pass), the loads of all of them in flight together; (3) the exact costs of the pairs that
                // passed, a few per round: squares by the whole wavefront, the canonical ordered sums one lane each.
                const unsigned long long lt_mask = (1ull << lane) - 1ull;
                int nf = 0, n_exact = 0, np = 0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const unsigned long long fm = __ballot(fail[c]);
                    if (fail[c]) {
                        flist[nf + __builtin_popcountll(fm & lt_mask)] = (unsigned char)kc[c];
                        rbest[kc[c]] = best[c]; rarg[kc[c]] = arg[c];
                    }
                    nf += __builtin_popcountll(fm);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int stride = (Dj + 2) | 1;                        // odd: the summing lanes read different banks
                const int npr = JF1_SQ / stride < 64 ? JF1_SQ / stride : 64;
                const float *__restrict__ jslab = Jlo + (t - 1) * (int64_t)K * K;
                int tviol = 0;                                          // tripwire of pass 1's bounds: every exact cost against its bound
                float tmin = __builtin_inff();
                const double tsc = (double)scale[t - 1], tunit = 2.0 * (double)ceps * tsc * tsc;
                auto flush = [&]() {
                    for (int r0 = 0; r0 < np; r0 += npr) {
                        const int nr = np - r0 < npr ? np - r0 : npr;
                        for (int r = 0; r < nr; ++r) {
                            const unsigned int e = (unsigned int)__builtin_amdgcn_readfirstlane((int)plist[r0 + r]);
                            const int64_t a = cand[(int64_t)(t - 1) * K + (e & 0xffu)], b = cand[(int64_t)t * K + (e >> 8)];
                            const float *__restrict__ re = JC_unw + (a + 1) * (int64_t)Jp;
                            const float *__restrict__ rs = JC_unw + b * (int64_t)Jp;
                            double *row = sq + r * stride;
#pragma unroll 4
                            for (int cc = lane; cc < Dj; cc += 64) {
                                const double w = wj[cc];
                                const double dd = __dsub_rn(__dmul_rn((double)re[cc], w), __dmul_rn((double)rs[cc], w));
                                row[cc] = __dmul_rn(dd, dd);
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        double totr = inf;
                        if (lane < nr) {                            // jf_exact_cost's sum, in its order
                            double acc = 0.0;
                            const double *row = sq + lane * stride;
                            int cc = 0;
                            for (; cc + 8 <= Dj; cc += 8) {
                                double v[8];
#pragma unroll
                                for (int i = 0; i < 8; ++i) v[i] = row[cc + i];
#pragma unroll
                                for (int i = 0; i < 8; ++i) acc = __dadd_rn(acc, v[i]);
                            }
                            for (; cc < Dj; ++cc) acc = __dadd_rn(acc, row[cc]);
                            const double cex = __dsqrt_rn(acc);
                            const unsigned int pe = plist[r0 + lane];
                            if constexpr (FST32) totr = (double)((float)dprev[pe & 0xffu] + (tprev[pe & 0xffu] + (float)cex));
                            else totr = __dadd_rn(dprev[pe & 0xffu], cex);
                            if (stats) jf_trip_check(cex, jslab[(int64_t)(pe & 0xffu) * K + (pe >> 8)], tunit, tviol, tmin);
                        }
                        for (int r = 0; r < nr; ++r) {              // in list order: a column's candidates one after the other
                            const double tr = jf_readlane_f64(totr, r);
                            const unsigned int e = plist[r0 + r];
                            const int kf = (int)(e >> 8), pp = (int)(e & 0xffu);
                            const double cur = rbest[kf];
                            const int ca = rarg[kf];
                            __builtin_amdgcn_wave_barrier();
                            if ((tr < cur || (tr == cur && pp < ca)) && lane == 0) { rbest[kf] = tr; rarg[kf] = pp; }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        }
                        n_exact += nr;
                    }
                    np = 0;
                };
                const int total = nf * K;
                for (int base = 0; base < total; base += 256) {     // four rounds of 64 pairs: their loads in flight together
                    bool want[4];
                    unsigned int e[4];
                    float lo[4], tpp[4];
                    double dpp[4], rb[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int idx = base + 64 * i + lane;
                        const int ix2 = idx < total ? idx : 0;
                        const int fi = ix2 / K, pp = ix2 - fi * K, kf = flist[fi];
                        e[i] = ((unsigned int)kf << 8) | (unsigned int)pp;
                        lo[i] = jslab[(int64_t)pp * K + kf];
                        dpp[i] = dprev[pp];
                        tpp[i] = FST32 ? tprev[pp] : 0.f;
                        rb[i] = rbest[kf];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // (FST32: the float32 total with the bound in place of the cost -- rounding is monotone, so it bounds the real one)
                        const double tlo = FST32 ? (double)((float)dpp[i] + (tpp[i] + lo[i])) : __dadd_rn(dpp[i], (double)lo[i]);
                        want[i] = base + 64 * i + lane < total && dpp[i] < inf && lo[i] < __builtin_inff() && tlo <= rb[i];
                        const unsigned long long wm = __ballot(want[i]);
                        if (want[i]) plist[np + __builtin_popcountll(wm & lt_mask)] = (unsigned short)e[i];
                        np += __builtin_popcountll(wm);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (np > JF1_PAIRS - 256) flush();
                }
                flush();
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (fail[c]) { best[c] = rbest[kc[c]]; arg[c] = rarg[kc[c]]; d[c] = FST32 ? best[c] + (double)(float)td[c] : __dadd_rn(td[c], best[c]); }
                if (stats) {
                    if (lane == 0) {
                        atomicAdd(&stats[0], (unsigned long long)nf); atomicAdd(&stats[1], 1ull);
                        if (n_exact) atomicAdd(&stats[2], (unsigned long long)n_exact);
                    }
                    jf_trip_commit(stats, tviol, tmin);
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (col[c]) {
                    if constexpr (FST32) { dcur[kc[c]] = td[c] < inf ? best[c] : inf; tcur[kc[c]] = td[c] < inf ? (float)td[c] : 0.f; }
                    else dcur[kc[c]] = d[c];
                    if constexpr (BPL) bp_lds[bp_off + kc[c]] = (unsigned char)arg[c];
                    else bp_global[bp_off + kc[c]] = (unsigned char)arg[c];
                }
                // relative to the previous step's bound: d~ is shifted towards 0 every 64 steps while delta grows, so the
                // differences are large numbers that move by about a step's cost -- THAT is what float32 may round
                const double diff = (col[c] && d[c] < inf) ? __dsub_rn(__dsub_rn(d[c], lbv[c]), off_base) : inf;
                const float df = (float)diff;
                dmin = df < dmin ? df : dmin;
            }
            const float m32 = jf_unimage(jf_wave_min_u32(jf_image(dmin)));
            // float32 rounding is monotone: the minimum of the rounded values is the rounded minimum, within 2^-24 of it;
            // the subtraction and the addition of off_base round once each (1e-15 of the magnitudes)
            if (m32 < __builtin_inff()) {
                const double o = off_base + ((double)m32 - 1.2e-7 * fabs((double)m32));
                off = o - 4e-16 * (fabs(o) + fabs(off_base));
                off_base = off;
            } else off = inf;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                       // delta_t is in LDS before the next step reads it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            JF_STAMP(3);
        } else if (last_of_batch) {
            // batch bidx + 1 has landed when at most the JF1_NL loads of batch bidx + 2 are still in flight
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(JF1_NL) : "memory");
        }
        bp_off += K;
        rp_base += K4;
        if (++sin == BS) {
            __syncthreads();                                       // the compute wavefront has left batch bidx; batch bidx + 1 is in the ring
            sin = 0; ++bidx;
            ring_slot = ring_slot == 2 ? 0 : ring_slot + 1;
            rp_base = ring_slot * slot_pieces;
            if (loader) dma_batch(bidx + 2);                       // into the slot batch bidx - 1 has left
            JF_STAMP(4);
        }
    }
    if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing in flight into LDS past this point (the ring is reused below)
    __syncthreads();

    const double *dlast = delta + ((T - 1) & 1) * 256;
    if (tid == 0) {
        double best = inf;
        int slot = 0;
        const float *tlast = tdp + ((T - 1) & 1) * 256;
        for (int kk = 0; kk < K; ++kk) {
            const double v = FST32 ? (double)((float)dlast[kk] + tlast[kk]) : dlast[kk];      // (FST32: the exit arc carries the last target cost)
            if (v < best) { best = v; slot = kk; }
        }
        if (best == inf) { *path_len = 0; *cost = inf; *final_slot_p = -1; }
        else { *path_len = T; *cost = best; *final_slot_p = slot; }
    }
    if (!BPL) __threadfence();
    __syncthreads();
    // back-trace as above
    if (*final_slot_p >= 0) {
        if constexpr (BPL) {
            unsigned char *slots = reinterpret_cast<unsigned char *>(ring);       // the ring is free now (T <= its size is checked by the launcher)
            if (tid == 0) {
                int slot = *final_slot_p;
                for (int64_t t = T - 1; t >= 0; --t) {
                    slots[t] = (unsigned char)slot;
                    if (t > 0) slot = bp_lds[t * K + slot];
                }
            }
            __syncthreads();
            for (int64_t t = tid; t < T; t += (int)blockDim.x) path[t] = cand[t * K + slots[t]];
        } else if (tid == 0) {
            int slot = *final_slot_p;
            for (int64_t t = T - 1; t >= 0; --t) {
                path[t] = cand[t * K + slot];
                if (t > 0) slot = bp_global[t * K + slot];
            }
        }
    }
}

static int g_sparse_waves = 1;
void set_viterbi_sparse_waves(int w) { g_sparse_waves = w == 4 ? 4 : 1; }

static void jf_trace_dump(unsigned long long *stats, int n, int64_t T, hipStream_t s)
{
#ifdef SNK_JF_TRACE
    if (!stats || n != 1) return;
    static std::vector<unsigned long long> tr(16 * 1024);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(tr.data(), stats + 128, tr.size() * 8, hipMemcpyDeviceToHost);
    const int64_t n_t = T < 1024 ? T : 1024;
    double seg[8] = {0};
    const int last = g_sparse_waves == 1 ? 3 : 5;
    for (int64_t t = 2; t + 1 < n_t; ++t) {
        const unsigned long long *a = &tr[(size_t)t * 16], *b = &tr[(size_t)(t + 1) * 16];
        for (int i = 0; i < last; ++i) seg[i] += (double)(a[i + 1] - a[i]);
        seg[5] += (double)(b[0] - a[last]);
        seg[7] += (double)(b[0] - a[0]);
    }
    {
        std::vector<double> v;
        for (int i = 0; i <= last; ++i) {
            v.clear();
            for (int64_t t = 2; t + 1 < n_t; ++t) {
                const unsigned long long *a = &tr[(size_t)t * 16], *b = &tr[(size_t)(t + 1) * 16];
                v.push_back(i < last ? (double)(a[i + 1] - a[i]) : (double)(b[0] - a[last]));
            }
            std::sort(v.begin(), v.end());
            fprintf(stderr, "[jf-trace] segment %d: median %.0f, p90 %.0f, max %.0f\n", i, v[v.size() / 2], v[v.size() * 9 / 10], v.back());
        }
    }
    fprintf(stderr, "[jf-trace] clocks per step %.0f: segments %.0f %.0f %.0f %.0f %.0f, loop %.0f\n", seg[7] / (n_t - 3), seg[0] / (n_t - 3),
            seg[1] / (n_t - 3), seg[2] / (n_t - 3), seg[3] / (n_t - 3), seg[4] / (n_t - 3), seg[5] / (n_t - 3));
#else
    (void)stats; (void)n; (void)T; (void)s;
#endif
}

void launch_viterbi_sparse(const int64_t *cand, const void *rec, const float *Jlo, const float *JC_unw, int Jp, int Dj,
                           const double *wj, const int64_t *off, int n_utts, int first_utt, int K, int64_t n_units,
                           unsigned char *bp_global, int64_t *path, int64_t *path_len, double *cost,
                           unsigned long long *stats, hipStream_t s, const float *scale, float ceps, bool fst32)
{
    for (int u0 = 0; u0 < n_utts; u0 += DpBatch::MAX) {
        const int n = (n_utts - u0 < DpBatch::MAX) ? n_utts - u0 : DpBatch::MAX;
        DpBatch batch;
        int64_t T = 0;
        for (int i = 0; i <= n; ++i) batch.off[i] = off[u0 + i];
        for (int i = 0; i < n; ++i) T = (off[u0 + i + 1] - off[u0 + i] > T) ? off[u0 + i + 1] - off[u0 + i] : T;
        batch.first = first_utt + u0;
        if (g_sparse_waves == 1 || fst32) {
            const int nc = (K + 63) / 64;
            int bs1 = (JF1_NL * 1024) / (K * 64);                  // steps per batch: what a JF1_NL-KB slot holds
            if (bs1 < 1) bs1 = 1;
            const size_t base1 = (size_t)(3 * 256 + JF1_SQ) * 8 + 256 * 4 + 16 + JF1_PAIRS * 2 + 256 + 2 * 256 * 4 + (size_t)3 * JF1_NL * 1024;
            const size_t bp_bytes1 = (size_t)T * K;
            const bool bpl1 = base1 + bp_bytes1 + 64 <= 150 * 1024 && (size_t)T <= (size_t)3 * JF1_NL * 1024;
            const size_t shmem1 = base1 + (bpl1 ? bp_bytes1 : 0);
#define SNK_SP1F(NC_, BPL_, F_)                                                                                   \
    {                                                                                                             \
        static size_t attr_set[32] = {0};                                                                         \
        lds_attr_ensure(attr_set, 150 * 1024, [] {                                                                \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&viterbi_sparse1_kernel<NC_, BPL_, F_>),     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024)); });         \
        hipLaunchKernelGGL((viterbi_sparse1_kernel<NC_, BPL_, F_>), dim3(n), dim3(128), shmem1, s, cand,          \
                           reinterpret_cast<const JfRecord *>(rec), Jlo, JC_unw, Jp, Dj, wj, batch, K, bs1, n_units, \
                           bp_global, path, path_len, cost, stats, scale, ceps);                                  \
    }
#define SNK_SP1(NC_, BPL_) { if (fst32) SNK_SP1F(NC_, BPL_, true) else SNK_SP1F(NC_, BPL_, false) }
            if (nc == 1) { if (bpl1) SNK_SP1(1, true) else SNK_SP1(1, false) }
            else if (nc == 2) { if (bpl1) SNK_SP1(2, true) else SNK_SP1(2, false) }
            else if (nc == 3) { if (bpl1) SNK_SP1(3, true) else SNK_SP1(3, false) }
            else { if (bpl1) SNK_SP1(4, true) else SNK_SP1(4, false) }
#undef SNK_SP1
#undef SNK_SP1F
            jf_trace_dump(stats, n, T, s);
            continue;
        }
        const int nth = 64 * 5;                                    // four compute wavefronts (refinement: one per failing column) + the loader
        const int bs = K <= 104 ? 4 : 2;                           // steps per loader batch (registers: bs K / 16 per lane)
        const size_t slot_bytes = (((size_t)bs * K * 4 + 63) & ~(size_t)63) * 16;
        const size_t base = (size_t)(3 * 256 + 16) * 8 + 16 + 256 * 4 + 256 + 3 * slot_bytes;
        const size_t bp_bytes = (size_t)T * K;
        const bool bp_in_lds = base + bp_bytes + 64 <= 150 * 1024 && (size_t)T <= 3 * slot_bytes;
        const size_t shmem = base + (bp_in_lds ? bp_bytes : 0);
#define SNK_SP(BS_, BPL_)                                                                                         \
    {                                                                                                             \
        static size_t attr_set[32] = {0};                                                                   \
        lds_attr_ensure(attr_set, 150 * 1024, [] {                                                                \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&viterbi_sparse_kernel<BS_, BPL_>),          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024)); });         \
        hipLaunchKernelGGL((viterbi_sparse_kernel<BS_, BPL_>), dim3(n), dim3(nth), shmem, s, cand,                \
                           reinterpret_cast<const JfRecord *>(rec), Jlo, JC_unw, Jp, Dj, wj, batch, K, n_units,   \
                           bp_global, path, path_len, cost, stats, scale, ceps);                                  \
    }
        if (bs == 4) { if (bp_in_lds) SNK_SP(4, true) else SNK_SP(4, false) }
        else { if (bp_in_lds) SNK_SP(2, true) else SNK_SP(2, false) }
#undef SNK_SP
        jf_trace_dump(stats, n, T, s);
    }
}

}  // namespace snk
