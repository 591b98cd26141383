// Device helpers shared by the float32 greedy scans (greedy32_kernels.hip: streamed; greedy_res_kernels.hip: database
// resident in LDS): the canonical float64 total of a window, the float32 error bounds, wavefront-uniform values.
#pragma once
#include "greedy_common.h"

namespace snk {

// (value, index) order: smaller value, then smaller index -- so that equal float32 totals keep the lowest index first
__device__ __forceinline__ bool lt_vi(float va, int64_t ia, float vb, int64_t ib) { return va < vb || (va == vb && ia < ib); }

// canonical float64 squared distance of window i to the step's reference of utterance u (the oracle's order:
// join columns, then target columns epoch by epoch; separately rounded sub / mul / add)
static __device__ double g32_exact_d2(const GreedyArgs &a, int u, int64_t step, int64_t prev_row, bool prev_is_current, int64_t i)
{
    double acc_j = 0.0, acc_t = 0.0;
    const int col0 = prev_is_current ? a.cur_col0 : a.prev_col0;
    const int64_t row0 = prev_is_current ? a.cur_row0 : a.prev_row0;
    const float *xr = a.JC_unw + (a.prev_row0 + i) * a.Jp + a.prev_col0;
    const float *rr = a.JC_unw + (row0 + (prev_row >= 0 ? prev_row : 0)) * a.Jp + col0;
    for (int c = 0; c < a.jdim; ++c) {
        const double xw = __dmul_rn((double)xr[c], a.wj[a.prev_col0 + c]);
        const double ref = prev_row >= 0 ? __dmul_rn((double)rr[c], a.wj[col0 + c]) : 0.0;
        const double d = __dsub_rn(xw, ref);
        acc_j = __dadd_rn(acc_j, __dmul_rn(d, d));
    }
    for (int k = 0; k < a.nep; ++k) {
        const float *fr = a.F_unw + (i + a.ep[k]) * a.Fp;
        const double *q = a.Q + (a.q_off[u] + step * a.me + a.ep[k]) * a.Dt;
        for (int c = 0; c < a.Dt; ++c) {
            const double d = __dsub_rn(__dmul_rn((double)fr[c], a.wt[c]), q[c]);
            acc_t = __dadd_rn(acc_t, __dmul_rn(d, d));
        }
    }
    return __dadd_rn(acc_j, acc_t);
}

// The same total computed by a whole wavefront: the per-column terms fl(fl(x w - ref)^2) in parallel (coalesced
// loads), then summed by ONE lane in the canonical order -- bit-identical to g32_exact_d2, without 1 000 dependent
// memory round trips.  terms: (jdim + nep Dt) doubles of LDS private to the wavefront.
// the canonical sum of one candidate's term array: join columns, then target columns, each a chain of dependent additions.
// The LDS reads in front of the additions are not dependent: eight at a time.  Any number of lanes may run it side by side
// on different arrays (the chains of eight candidates cost the time of one).
__device__ __forceinline__ double g32_chain_sum(const double *terms, int jdim, int nt)
{
    auto chain = [&](const double *t, int n) {
        double acc = 0.0;
        int c = 0;
        for (; c + 8 <= n; c += 8) {
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = t[c + j];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __dadd_rn(acc, v[j]);
        }
        for (; c < n; ++c) acc = __dadd_rn(acc, t[c]);
        return acc;
    };
    const double acc_j = chain(terms, jdim), acc_t = chain(terms + jdim, nt);
    return __dadd_rn(acc_j, acc_t);
}

// epoch e of the window (GreedyArgs::ep, greedy_fill_args) without indexing the argument block by a lane's value: all frames of
// the window, or the first and the last one
__device__ __forceinline__ int g32_ep(const GreedyArgs &a, int e) { return a.nep == a.me ? e : (e ? a.me - 1 : 0); }

// The term arrays of up to four candidates at once: weight and reference of a column are the same for all of them, only
// the database value differs -- so the operands of all candidates are requested together (one trip to HBM for their cold
// rows instead of one per candidate) in the registers one candidate took.  terms + k * stride: array of candidate k.
static __device__ void g32_terms_multi(const GreedyArgs &a, int u, int64_t step, int64_t prev_row, bool prev_is_current,
                                const int64_t (&ids)[4], int cnt, double *terms, size_t stride, int lane)
{
    const int col0 = prev_is_current ? a.cur_col0 : a.prev_col0;
    const int64_t row0 = prev_is_current ? a.cur_row0 : a.prev_row0;
    const float *rr = a.JC_unw + (row0 + (prev_row >= 0 ? prev_row : 0)) * a.Jp + col0;
    const int ncol = a.jdim + a.nep * a.Dt;
    constexpr int NJ = 9;
    for (int base = 0; base < ncol; base += 64 * NJ) {
        float x[4][NJ], rx[NJ];
        double w[NJ], rw[NJ], qv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int idx = base + 64 * j + lane;
            rx[j] = 0.f; w[j] = 0.0; rw[j] = 0.0; qv[j] = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k][j] = 0.f;
            if (idx < a.jdim) {
                w[j] = a.wj[a.prev_col0 + idx];
                if (prev_row >= 0) { rx[j] = rr[idx]; rw[j] = a.wj[col0 + idx]; }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < cnt) x[k][j] = a.JC_unw[(a.prev_row0 + ids[k]) * a.Jp + a.prev_col0 + idx];
            } else if (idx < ncol) {
                const int t = idx - a.jdim, e = t / a.Dt, c = t - e * a.Dt;
                w[j] = a.wt[c];
                const int epe = g32_ep(a, e);
                qv[j] = a.Q[(a.q_off[u] + step * a.me + epe) * a.Dt + c];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < cnt) x[k][j] = a.F_unw[(ids[k] + epe) * a.Fp + c];
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int idx = base + 64 * j + lane;
            if (idx < ncol) {
                const double ref = idx < a.jdim ? (prev_row >= 0 ? __dmul_rn((double)rx[j], rw[j]) : 0.0) : qv[j];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < cnt) {
                        const double d = __dsub_rn(__dmul_rn((double)x[k][j], w[j]), ref);
                        terms[(size_t)k * stride + idx] = __dmul_rn(d, d);
                    }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wavefront's own LDS writes, in order
    __builtin_amdgcn_wave_barrier();
}

static __device__ double g32_exact_d2_wave(const GreedyArgs &a, int u, int64_t step, int64_t prev_row, bool prev_is_current, int64_t i,
                                    double *terms, int lane, bool terms_only = false)
{
    const int col0 = prev_is_current ? a.cur_col0 : a.prev_col0;
    const int64_t row0 = prev_is_current ? a.cur_row0 : a.prev_row0;
    const float *xr = a.JC_unw + (a.prev_row0 + i) * a.Jp + a.prev_col0;
    const float *rr = a.JC_unw + (row0 + (prev_row >= 0 ? prev_row : 0)) * a.Jp + col0;
    // all columns as one index space, 576 per round: the operands of nine columns per lane are requested together
    // (the rows of a candidate are cold: nine dependent trips to HBM, one per 64 columns, were 25 us per candidate)
    const int ncol = a.jdim + a.nep * a.Dt;
    constexpr int NJ = 9;
    for (int base = 0; base < ncol; base += 64 * NJ) {
        float x[NJ], rx[NJ];
        double w[NJ], rw[NJ], qv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int idx = base + 64 * j + lane;
            x[j] = 0.f; rx[j] = 0.f; w[j] = 0.0; rw[j] = 0.0; qv[j] = 0.0;
            if (idx < a.jdim) {
                x[j] = xr[idx]; w[j] = a.wj[a.prev_col0 + idx];
                if (prev_row >= 0) { rx[j] = rr[idx]; rw[j] = a.wj[col0 + idx]; }
            } else if (idx < ncol) {
                const int t = idx - a.jdim, k = t / a.Dt, c = t - k * a.Dt;
                const int epk = g32_ep(a, k);
                x[j] = a.F_unw[(i + epk) * a.Fp + c]; w[j] = a.wt[c];
                qv[j] = a.Q[(a.q_off[u] + step * a.me + epk) * a.Dt + c];
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int idx = base + 64 * j + lane;
            if (idx < ncol) {
                const double xw = __dmul_rn((double)x[j], w[j]);
                const double ref = idx < a.jdim ? (prev_row >= 0 ? __dmul_rn((double)rx[j], rw[j]) : 0.0) : qv[j];
                const double d = __dsub_rn(xw, ref);
                terms[idx] = __dmul_rn(d, d);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wavefront's own LDS writes, in order
    __builtin_amdgcn_wave_barrier();
    if (terms_only) return 0.0;
    double d2 = 0.0;
    if (lane == 0) d2 = g32_chain_sum(terms, a.jdim, a.nep * a.Dt);
    __builtin_amdgcn_wave_barrier();
    return __shfl(d2, 0, 64);
}

// wavefront-uniform values read from LDS live in scalar registers across the scan
__device__ __forceinline__ int64_t g32_uniform_i(int64_t v)
{
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)(unsigned long long)v);
    const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)((unsigned long long)v >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double g32_uniform_d(double v) { return __longlong_as_double(g32_uniform_i(__double_as_longlong(v))); }

// error bound of a float32 total (see the file header); V2 = squared norm of the reference vector
__device__ __forceinline__ double g32_err(double d, double V2, int ncols)
{
    const double u = 5.9604644775390625e-08;
    return 6.0 * u * sqrt(V2 * d) * 1.01 + (double)(ncols + 8) * u * d;
}

// ... of a total whose join columns were rounded to float16 first: x~ = x (1 + delta) + eta, |delta| <= 2^-11, |eta| <= 2^-25,
// so the vector of differences moves by at most D = 2^-11 max_i ||w o S'[i]|| + 2^-25 ||w|| (a.f16_delta), the total by
// E16(d) = 2 sqrt(d) D + D^2, and the float32 evaluation error applies to a total of at most d + E16(d)
__device__ __forceinline__ double g32_err16(double d, double V2, int ncols, double D)
{
    const double e16 = (2.0 * sqrt(d) * D + D * D) * 1.01;
    return g32_err(d + e16, V2, ncols) + e16;
}

// Tripwire of the float32 scans' bound (status words 8 and 9 of a launch).  Whenever a step is decided by exact totals, every
// window weighed has d~ >= M (M: the float32 minimum the bound tau was built from) and, if the bound holds, an exact total
// d >= M - (E(d) + EW).  used = (M - d) / (E(d) + EW) is the share of the bound a window consumed: > 1 is a VIOLATION (the
// scan's approximate total of some window was off by more than the proven -- for the hoisted bf16 product: probed -- bound).
// status[8] += violations, status[9] = max used (float bits; non-negative floats order like their bit patterns).
__device__ __forceinline__ void g32_trip(int64_t *status, double M, double d, double bound, float &seen)
{
    const double used = bound > 0.0 ? (M - d) / bound : 0.0;
    const float uf = used > 0.0 ? (float)used : 0.f;
    if (uf > seen) {                                            // rare after the first steps: the lane's own running maximum gates the atomics
        seen = uf;
        atomicMax(reinterpret_cast<unsigned int *>(&status[9]), __builtin_bit_cast(unsigned int, uf));
        if (used > 1.0) atomicAdd(reinterpret_cast<unsigned long long *>(&status[8]), 1ull);
    }
}

}  // namespace snk
