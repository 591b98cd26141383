// Pass 1 of the sparse Viterbi path (joinfast_kernels.hip), second form -- the default, option join_lb_variant 1:
// proven float32 lower bounds of the join costs
//   make_on_the_fly_join_lattice_BLOCK_DIRECT / get_natural_distance_vectorised (script/synth_halfphone.py:3206-3322, :2942-2951)
// from a float32 copy of the WEIGHTED join rows (join_weight32_kernel, built once per set of weights: (N + 1) x Jq floats,
// padding columns 0) on the bf16 matrix pipe.  What the first form (join_lb_kernel) spends per gathered element -- a float64
// multiply by the weight, a float64 subtraction, two conversions, on the vector unit, beside float32 MFMAs that issue at
// 1/16 of the bf16 rate -- becomes one float32 subtraction and a split into two bf16 pieces.
//
//   u~ = fl32(u), u = fl64(x w) the canonical weighted value;   y = fl32(u~ - m~), m~ = u~ of the step's reference row
//   (the start vector of the first candidate of row r + 1: any row of the matrix would do);   y = h + l + rr,
//   h = bf16(y), l = bf16(y - h) (round to nearest even, v_cvt_pk_bf16_f32: |l| <= 2^-9 (1 + 2^-8) |y|, |rr| <= 2^-18 |y|);
//   G~ = Ghh + Gx,  Ghh = sum over the k-blocks of 16 columns of he.hs,  Gx = sum of he.ls + le.hs
//   (v_mfma_f32_32x32x16_bf16, float32 accumulation; with K <= 128 the two sums have accumulators of their own, so the
//   long chain -- the one whose running sum is of the size of the products -- is n_kb MFMAs, not 3 n_kb);
//   ne~, ns~ = float32 sums of squares of ye, ys (8-term chains per k-block, the k-blocks added in order).
// Bound.  (i) representation:  || (ye - ys) - (ue - us) || <= eta := 2^-24 (||ue|| + ||us||) + 2^-24 (1 + 2^-23)(||ye|| + ||ys||)
//   (the two roundings of every element; m~ cancels), so c = ||ue - us|| >= ||ye - ys|| - eta, with ||u|| <= Umax, the largest
//   row norm of the copy (measured when it is built).  (ii) || ye - ys ||^2 = ne + ns - 2 ye.ys against c2~ = ne~ + ns~ - 2 G~:
//     |G~ - ye.ys| <= cG ||ye|| ||ys||,  cG = 1.02 [ 3.02 2^-18   (what the split drops: le.ls + rre.ys + (he + le).rrs)
//                      + (n_kb + 1) 2^-20 + (2 n_kb + 1) 2^-28 + 2^-24   (two accumulator sets: every MFMA off by at most
//                        2^-20 of its |products| + |C| -- the probed property of this instruction, knn16_kernels.hip /
//                        snk_probe_mfma_bf16 --, the cross terms' sums are 2^-8 of the product of the norms; one float32 addition)
//                      or (3 n_kb + 1) 2^-20   (one set, K > 128) ],
//     ||ye|| ||ys|| <= (ne + ns) / 2,   |ne~ - ne| <= (n_kb + 12) 2^-24 ne,   six float32 roundings in the epilogue,
//   together e2 = ceps (ne~ + ns~), ceps from join_lb2_ceps: 3.4e-5 at 302 columns with two sets (7.1e-5 with one; the first
//   form's 2.2 (D + 6) 2^-24 is 4.1e-5 there).  clo = sqrt(c2~ - e2) (1 - 2^-21) - eta (1 + 1e-4), clamped at 0
//   (v_sqrt_f32, 1 ulp: the factor leaves 8; a negative radicand gives NaN, which the clamp turns into 0).  The epilogue
//   evaluates this as fma(sqrt(ne' + ns' - 2 G~), 1 - 2^-21, -(se' + ss')) with the per-row terms ne' = (1 - ceps) ne~ - 1e-30,
//   se' = c24 (sqrt(ne~) + 2 Umax) prepared once per row (9 vector-issue slots per cell instead of 15).
// One workgroup per row pair; KT = ceil(K / 32) wavefronts; wavefront w keeps the pieces of its 32 E rows in registers (the A
// operand: lane l <-> row l & 31, columns 16 kb + 8 (l >> 5) + 0..7) and stages those of its 32 S rows in LDS for everybody
// (fragment order, double buffered, one barrier per k-block).  The rows of the next PF k-blocks are in flight in registers
// (16-byte loads; a row's 64 bytes of a k-block are two lanes' 32 bytes each); the reference row sits in LDS.
#include "snk_internal.h"
#include <float.h>

namespace snk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 jf_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 jf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float jf_f32x2 __attribute__((ext_vector_type(2)));
typedef float jf_f32x16 __attribute__((ext_vector_type(16)));
#define JF2_MAXD 1024          // join columns (padded to 16) the reference row in LDS holds

__device__ __forceinline__ bool jf2_usable(int64_t id, int64_t n_units)
{
    return id >= 1 && id < n_units - 1;        // synth_halfphone.py:3238-3268
}

__global__ void __launch_bounds__(256)
join_weight32_kernel(const float *__restrict__ JC_unw, int Jp, int64_t Njc, int Dj, const double *__restrict__ wj,
                     float *__restrict__ JW, int Jq, unsigned int *__restrict__ umax_bits)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Njc) return;
    double n2 = 0.0;
    for (int c = lane; c < Jq; c += 64) {
        const float v = c < Dj ? (float)__dmul_rn((double)JC_unw[row * Jp + c], wj[c]) : 0.f;
        JW[row * Jq + c] = v;
        n2 += (double)v * (double)v;
    }
#pragma unroll
    for (int m = 1; m <= 32; m <<= 1) n2 += __shfl_xor(n2, m, 64);
    if (lane == 0) {
        const float nf = (float)sqrt(n2) * 1.000001f;          // rounded up; non-negative floats order like their bit patterns
        atomicMax(umax_bits, nf == nf ? __float_as_uint(nf) : 0x7f800000u);      // NaN: +inf (every bound becomes 0)
    }
}

void launch_join_weight32(const float *JC_unw, int Jp, int64_t Njc, int Dj, const double *wj, float *JW, int Jq,
                          unsigned int *umax_bits, hipStream_t s)
{
    (void)hipMemsetAsync(umax_bits, 0, sizeof(unsigned int), s);
    hipLaunchKernelGGL(join_weight32_kernel, dim3((unsigned)((Njc + 3) / 4)), dim3(256), 0, s, JC_unw, Jp, Njc, Dj, wj, JW, Jq,
                       umax_bits);
}

int join_lb2_pitch(int Dj) { return (Dj + 15) & ~15; }
bool join_lb2_supported(int Dj, int K) { return join_lb2_pitch(Dj) <= JF2_MAXD && K >= 1 && K <= 208; }

// one pair of float32 values into two bf16 pieces each (v_cvt_pk_bf16_f32: round to nearest even); the pair's share of the
// row's sum of squares is added to n2 (lane-wise: even and odd columns have chains of their own).  Scalar float32 arithmetic on
// purpose (and -fno-slp-vectorize in the Makefile): beside MFMAs a v_pk_add_f32 / v_pk_fma_f32 costs the issue port more than the
// two scalar instructions it replaces (MI355X_MICROARCH.md, 'price of one filler beside MFMAs'), and this kernel is issue-bound
__device__ __forceinline__ void jf_split2(float y0, float y1, unsigned int &hi, unsigned int &lo, float &n0, float &n1)
{
    n0 = __builtin_fmaf(y0, y0, n0);
    n1 = __builtin_fmaf(y1, y1, n1);
    const jf_f32x2 y = {y0, y1};
    const unsigned int hb = __builtin_bit_cast(unsigned int, __builtin_convertvector(y, jf_bf16x2));
    const float r0 = y0 - __builtin_bit_cast(float, hb << 16), r1 = y1 - __builtin_bit_cast(float, hb & 0xffff0000u);      // exact
    hi = hb;
    const jf_f32x2 rr = {r0, r1};
    lo = __builtin_bit_cast(unsigned int, __builtin_convertvector(rr, jf_bf16x2));
}

// ONESET (K <= 128 only; option join_lb_one_set): ONE accumulator set and one k-block in flight instead of two and two -- 119
// registers instead of 246 at KT = 4, four workgroups per compute unit instead of two -- at the looser error constant of the
// long chain (join_lb2_ceps: 7.1e-5 instead of 3.4e-5 at 302 columns).  The kernel touches none of its ceilings with two
// wavefronts per SIMD parked at the same barrier; four hide the gather's latency behind one another.
template <int KT, bool QUAD, bool ONESET = false>
__global__ void __launch_bounds__(64 * KT, ONESET ? 4 : 2)
join_lb2_kernel(const float *__restrict__ JW, int Jq, int n_kb, const unsigned int *__restrict__ umax_bits, float omc /* 1 - ceps, rounded down */,
                int64_t n_units, const int64_t *__restrict__ cand, int K, float *__restrict__ Jlo,
                float *__restrict__ scale_out, int Kq)
{
    constexpr bool TWO = KT <= 4 && !ONESET;                  // accumulators of their own for the cross terms
    constexpr int PF = ONESET ? 1 : (KT <= 3) ? 3 : 2;        // k-blocks in flight
    __shared__ u32x4 Bs[2][KT][2][64];                        // [buffer][S tile][hi, lo][lane]
    __shared__ __align__(16) float m_s[JF2_MAXD];
    __shared__ float ne_s[32 * KT], ns_s[32 * KT], sne_s[32 * KT], sns_s[32 * KT];
    __shared__ int smax_bits;
    const int64_t r = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row32 = lane & 31, half = lane >> 5;
    const int kk = wave * 32 + row32;
    // Kq > 0 (K > 128): the K x K matrix of a step in 2 x 2 QUADRANTS of Kq x Kq, one workgroup each (blockIdx.y): E rows
    // e0 .. e0 + Ke - 1 against S rows s0 .. s0 + Ks - 1, the step's reference row the same for all four.  Four workgroups of
    // four wavefronts with two accumulator sets each (the tighter bound) instead of one of seven with one set that leaves a
    // compute unit no room for a second workgroup, at twice the pieces split (join_lb2_quadrant below: when it pays).
    const int e0 = QUAD ? (int)(blockIdx.y >> 1) * Kq : 0, s0 = QUAD ? (int)(blockIdx.y & 1) * Kq : 0;
    const int Ke = QUAD ? (K - e0 < Kq ? K - e0 : Kq) : K, Ks = QUAD ? (K - s0 < Kq ? K - s0 : Kq) : K;
    const int64_t idE = kk < Ke ? cand[r * K + e0 + kk] : -1;
    const int64_t idS = kk < Ks ? cand[(r + 1) * K + s0 + kk] : -1;
    const int64_t id0 = cand[(r + 1) * K];
    const bool okE = jf2_usable(idE, n_units), okS = jf2_usable(idS, n_units);
    const float *const pE = JW + (okE ? idE + 1 : 0) * (int64_t)Jq + 8 * half;      // unit_end_data[a]   = JC[a+1]
    const float *const pS = JW + (okS ? idS : 0) * (int64_t)Jq + 8 * half;          // unit_start_data[b] = JC[b]
    const int DC = n_kb * 16;

    f32x4 rE[PF][2], rS[PF][2];
    auto fetch = [&](int kb, f32x4 (&e)[2], f32x4 (&s)[2]) {
        const int c = kb * 16;
        e[0] = *reinterpret_cast<const f32x4 *>(pE + c); e[1] = *reinterpret_cast<const f32x4 *>(pE + c + 4);
        s[0] = *reinterpret_cast<const f32x4 *>(pS + c); s[1] = *reinterpret_cast<const f32x4 *>(pS + c + 4);
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) fetch(p < n_kb ? p : n_kb - 1, rE[p], rS[p]);
    {
        const float *const row0 = JW + ((id0 >= 0 && id0 <= n_units) ? id0 : 0) * (int64_t)Jq;
        for (int c = tid; c < DC; c += 64 * KT) m_s[c] = row0[c];
    }
    if (tid == 0) smax_bits = 0;

    jf_f32x16 acc[KT], accx[TWO ? KT : 1];
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[j][i] = 0.f; if (TWO) accx[j][i] = 0.f; }
    float ne = 0.f, ns = 0.f;
    auto mfma = [](const u32x4 &a, const u32x4 &b, jf_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(jf_bf16x8, a), __builtin_bit_cast(jf_bf16x8, b), c, 0, 0, 0);
    };
    __syncthreads();                                          // the reference row is in LDS

    // one k-block: rows out of the ring slot, the slot refilled with the k-block `next` (the caller clamps it: a clamped
    // re-load is never consumed), pieces, this wavefront's S pieces to LDS, barrier, 3 KT MFMAs
    auto block = [&](int kb, int next, f32x4 (&e)[2], f32x4 (&s)[2]) {
        const f32x4 m0 = *reinterpret_cast<const f32x4 *>(&m_s[kb * 16 + 8 * half]);
        const f32x4 m1 = *reinterpret_cast<const f32x4 *>(&m_s[kb * 16 + 8 * half + 4]);
        const float mm[8] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
        const float ee[8] = {e[0][0], e[0][1], e[0][2], e[0][3], e[1][0], e[1][1], e[1][2], e[1][3]};
        const float ss[8] = {s[0][0], s[0][1], s[0][2], s[0][3], s[1][0], s[1][1], s[1][2], s[1][3]};
        float ye[8], ys[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ye[j] = ee[j] - mm[j]; ys[j] = ss[j] - mm[j]; }
        if (next >= 0) fetch(next, e, s);                     // (a compile-time decision at every call site)
        float te0 = 0.f, te1 = 0.f, ts0 = 0.f, ts1 = 0.f;
        u32x4 he, le, hs, ls;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned int a, b, c, d;
            jf_split2(ye[2 * j], ye[2 * j + 1], a, b, te0, te1); jf_split2(ys[2 * j], ys[2 * j + 1], c, d, ts0, ts1);
            he[j] = a; le[j] = b; hs[j] = c; ls[j] = d;
        }
        ne += te0 + te1; ns += ts0 + ts1;
        Bs[kb & 1][wave][0][lane] = hs;
        Bs[kb & 1][wave][1][lane] = ls;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const u32x4 bh = Bs[kb & 1][j][0][lane], bl = Bs[kb & 1][j][1][lane];
            if (TWO) {
                accx[j] = mfma(he, bl, accx[j]);
                accx[j] = mfma(le, bh, accx[j]);
            } else {
                acc[j] = mfma(he, bl, acc[j]);
                acc[j] = mfma(le, bh, acc[j]);
            }
            acc[j] = mfma(he, bh, acc[j]);
        }
    };
    int kb = 0;
    for (; kb + PF <= n_kb; kb += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int nx = kb + p + PF;
            block(kb + p, nx < n_kb ? nx : n_kb - 1, rE[p], rS[p]);
        }
    }
    if (kb < n_kb) {                                          // the last n_kb % PF k-blocks: nothing left to fetch
        block(kb, -1, rE[0], rS[0]);
        if (PF > 2 && kb + 1 < n_kb) block(kb + 1, -1, rE[1], rS[1]);
    }
    // row norms: the two lanes of a row hold the sums of its two column halves
    ne += __shfl_xor(ne, 32, 64);
    ns += __shfl_xor(ns, 32, 64);
    const float umax2 = 2.f * __uint_as_float(*umax_bits);
    const float c24 = 5.9604644775390625e-08f * 1.0001f;
    if (half == 0) {
        // Everything of the epilogue that depends on ONE row only is done here, once per row instead of once per cell (the
        // epilogue was a third of the kernel's vector instructions: 15 per cell, 64 cells per lane):
        //   ne' = (1 - ceps) ne - 1e-30   (so that lo2 = ne' + ns' - 2 g~ = c2~ - ceps (ne~ + ns~) - 2e-30; omc is rounded down),
        //   +inf for an unusable unit (every cell of its row / column then comes out +inf by itself: inf - finite = inf,
        //   sqrt(inf) = inf, fma(inf, 1 - 2^-21, -t) = inf);   se' = c24 (sqrt(ne) + 2 Umax) -- its share of eta.
        ne_s[kk] = okE ? __builtin_fmaf(omc, ne, -1e-30f) : __builtin_inff();
        ns_s[kk] = okS ? __builtin_fmaf(omc, ns, -1e-30f) : __builtin_inff();
        sne_s[kk] = c24 * (__builtin_amdgcn_sqrtf(ne) * 1.000001f + umax2);
        sns_s[kk] = c24 * (__builtin_amdgcn_sqrtf(ns) * 1.000001f);
        // scale of the step (margin of pass 2): the largest centred norm among the usable rows
        const float big = fmaxf(okE ? ne : 0.f, okS ? ns : 0.f);
        atomicMax(&smax_bits, __float_as_int(big));
    }
    __syncthreads();
    // ---- epilogue, branch-free: 16 KT cells per lane; stores through a buffer descriptor of the step's K x K slab (the row
    // part of the address in the scalar offset, a lane whose column lies beyond K points out of range and is dropped).
    // Overflowed float32 norms and negative lo2 end as NaN under the final maximum (v_max returns the other operand for a
    // NaN; sqrt of a negative is NaN): bound 0.
    float nev[16], sev[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int kp = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;       // row of the result = E row (crow32)
        nev[i] = ne_s[kp];
        sev[i] = sne_s[kp];
    }
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(Jlo + r * (int64_t)K * K, 0, K * K * 4, 0x00020000);
    auto cell = [&](int j, int i, float nsv, float ssv) {
        const float sum = nev[i] + nsv;                                      // (1 - ceps)(ne~ + ns~) - 2e-30, or +inf
        const float g = TWO ? acc[j][i] + accx[j][i] : acc[j][i];
        const float lo2 = __builtin_fmaf(-2.f, g, sum);                      // (2 g is exact: one rounding)
        const float sq = __builtin_amdgcn_sqrtf(lo2);                        // NaN for lo2 < 0: bound 0 below
        return __builtin_fmaxf(__builtin_fmaf(sq, 1.f - 4.76837158203125e-07f, -(sev[i] + ssv)), 0.f);
    };
    if (wave * 32 + 32 <= Ke) {                               // uniform: every row of this wavefront's tile exists
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const int k = j * 32 + row32;                       // column of the result = S row
            const float nsv = ns_s[k], ssv = sns_s[k];
            const int voff = k < Ks ? ((e0 + wave * 32 + 4 * half) * K + s0 + k) * 4 : 0x7ffffffc;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, cell(j, i, nsv, ssv)), ores, voff,
                                                      ((i & 3) + 8 * (i >> 2)) * K * 4, 0);
        }
    } else {
        // the wavefront of the last, partly filled tile (K = 100: rows 96..127, four of them candidates): result registers whose
        // row lies beyond K in BOTH halves are skipped (a uniform test per register: 12 of its 16 at K = 100)
        float nsv[KT], ssv[KT];
#pragma unroll
        for (int j = 0; j < KT; ++j) { const int k = j * 32 + row32; nsv[j] = ns_s[k]; ssv[j] = sns_s[k]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kp0 = wave * 32 + (i & 3) + 8 * (i >> 2);
            if (kp0 >= Ke) continue;
            const int kp = kp0 + 4 * half;
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const int k = j * 32 + row32;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, cell(j, i, nsv[j], ssv[j])), ores,
                                                      (kp < Ke && k < Ks) ? ((e0 + kp) * K + s0 + k) * 4 : 0x7ffffffc, 0, 0);
            }
        }
    }
    if (tid == 0) {
        const float sc = __builtin_amdgcn_sqrtf(__int_as_float(smax_bits)) * 1.000001f;
        if (QUAD) atomicMax(reinterpret_cast<unsigned int *>(scale_out) + r, __float_as_uint(sc));      // (zeroed before the launch; non-negative floats order like their bits)
        else scale_out[r] = sc;
    }
}

// process-wide switch, option join_lb_one_set (like join_lb_quadrants below)
static int g_one_set = 0;
void set_join_lb_one_set(int v) { g_one_set = v ? 1 : 0; }
int get_join_lb_one_set() { return g_one_set; }

template <int KT>
static void launch_join_lb2_t(const float *JW, int Jq, int n_kb, const unsigned int *umax_bits, float omc, int64_t n_units,
                              const int64_t *cand, int64_t R, int K, float *Jlo, float *scale, hipStream_t s, int Kq)
{
    if (Kq > 0) {
        if constexpr (KT <= 4)
            hipLaunchKernelGGL((join_lb2_kernel<KT, true>), dim3((unsigned)(R - 1), 4u), dim3(64 * KT), 0, s, JW, Jq, n_kb, umax_bits, omc,
                               n_units, cand, K, Jlo, scale, Kq);
        return;
    }
    if constexpr (KT <= 4) {
        if (g_one_set) {
            hipLaunchKernelGGL((join_lb2_kernel<KT, false, true>), dim3((unsigned)(R - 1), 1u), dim3(64 * KT), 0, s, JW, Jq, n_kb, umax_bits, omc,
                               n_units, cand, K, Jlo, scale, 0);
            return;
        }
    }
    hipLaunchKernelGGL((join_lb2_kernel<KT, false>), dim3((unsigned)(R - 1), 1u), dim3(64 * KT), 0, s, JW, Jq, n_kb, umax_bits, omc,
                       n_units, cand, K, Jlo, scale, 0);
}

// rows of a quadrant (0: the matrix in one piece).  Process-wide switch, option join_lb_quadrants (default 0).  Measured at B4
// (1.5 M units, K 200, 32 utterances per step): the pass 2.97 -> 2.2 ms per launch and, with the two accumulator sets' tighter
// bounds, half the refinements in pass 4 (8.8 -> 6.0 ms per step) -- but inside the batch pipeline the K-NN stream's persistent
// sweeps no longer find room beside four times as many workgroups of 246 registers (stage A 0.5 -> 2 ms per launch) and the step
// as a whole gets slower (9.2 -> 10.2 ms): on for callers that run the Viterbi side alone, off where the K-NN runs beside it.
static int g_quadrants = 0;
void set_join_lb_quadrants(int q) { g_quadrants = q ? 1 : 0; }
int get_join_lb_quadrants() { return g_quadrants; }
static int join_lb2_quadrant(int K) { return (g_quadrants && K > 128) ? (K + 1) / 2 : 0; }

double join_lb2_ceps(int Dj, int K)
{
    const double n_kb = (double)(join_lb2_pitch(Dj) / 16);
    const double u24 = 5.9604644775390625e-08, u20 = 9.5367431640625e-07, u18 = 3.814697265625e-06;
    const int Kt = join_lb2_quadrant(K) > 0 ? join_lb2_quadrant(K) : K;
    const bool two = (Kt + 31) / 32 <= 4 && !(g_one_set && join_lb2_quadrant(K) == 0);
    const double acc = two ? (n_kb + 1.0) * u20 + (2.0 * n_kb + 1.0) * u20 / 256.0 + u24 : (3.0 * n_kb + 1.0) * u20;
    const double cg = 1.02 * (3.02 * u18 + acc);
    const double gam = (n_kb + 12.0) * u24;
    // (six float32 roundings in the epilogue since the per-row terms are prepared once per row: (1 - ceps) ne, its sum with the
    // other row's, the fused -2 g~; 4 until round 5)
    return (gam + cg * (1.0 + gam) + 6.0 * u24) * 1.0001;
}

void launch_join_lb2(const float *JW, int Dj, const unsigned int *umax_bits, int64_t n_units, const int64_t *cand, int64_t R,
                     int K, float *Jlo, float *scale, hipStream_t s)
{
    if (R < 2) return;
    const int Jq = join_lb2_pitch(Dj), n_kb = Jq / 16;
    // 1 - ceps as the kernel multiplies the norms by it, rounded DOWN (a smaller factor is a looser, still valid bound)
    const float omc = __builtin_nextafterf((float)(1.0 - join_lb2_ceps(Dj, K)), 0.f);
    const int Kq = join_lb2_quadrant(K);
    const int kt = ((Kq > 0 ? Kq : K) + 31) / 32;
    if (Kq > 0) (void)hipMemsetAsync(scale, 0, (size_t)(R - 1) * sizeof(float), s);       // the quadrants' workgroups take the maximum
#define SNK_JLB2(KT_) launch_join_lb2_t<KT_>(JW, Jq, n_kb, umax_bits, omc, n_units, cand, R, K, Jlo, scale, s, Kq)
    switch (kt) {
    case 1: SNK_JLB2(1); break;
    case 2: SNK_JLB2(2); break;
    case 3: SNK_JLB2(3); break;
    case 4: SNK_JLB2(4); break;
    case 5: SNK_JLB2(5); break;
    case 6: SNK_JLB2(6); break;
    default: SNK_JLB2(7); break;
    }
#undef SNK_JLB2
}

}  // namespace snk
