// Join-cost matrices and the T-step Viterbi recursion for gfx950.
//
// Replaces, for one utterance:
//   make_on_the_fly_join_lattice_BLOCK_DIRECT  (script/synth_halfphone.py:3206-3322)
//     -> get_natural_distance_vectorised       (script/synth_halfphone.py:2942-2951)
//   make_target_sausage_lattice / cost_cache_to_compiled_fst / openfst.compose /
//   openfst.shortestpath                       (script/fst_functions_wrapped.py:28-58,172-217,368,389)
//
// join_cost_kernel: all (T-1) K x K matrices in one launch, one workgroup per column pair
//   (every pair is independent).  Distances are accumulated in DIFFERENCE form, column by
//   column with separately rounded sub/mul/add -- the canonical oracle order -- because
//   naturally adjacent units must join at exactly 0.0 (E[a] == S[a+1] bit for bit), which a
//   ||e||^2+||s||^2-2e.s form cannot guarantee.  Candidate rows are gathered from HBM in
//   32-column chunks into LDS and each thread keeps an RT x RT block of pair accumulators.
// viterbi_dp_kernel: one 1024-thread workgroup walks the trellis; the K x K slab of the next
//   step is prefetched into registers while the current step reduces, back-pointers stay in LDS.
#include "snk_internal.h"
#include <float.h>

namespace snk {

__device__ __forceinline__ bool unit_usable(int64_t id, int64_t n_units)
{
    // mini = 1, maxi = data_frames - 1 and the -1 padding value (synth_halfphone.py:3238-3268)
    return id >= 1 && id < n_units - 1;
}

template <int RT, int DC>
__global__ void __launch_bounds__(64)
join_cost_kernel(const double *__restrict__ JCw, int Djpad, int64_t n_units,
                 const int64_t *__restrict__ cand, int64_t T, int K, double *__restrict__ J)
{
    // one wavefront per (column pair t, 8RT x 8RT block of the K x K matrix): fine-grained work
    // items (2396 at T=600, K=100) balance over the chip; thread (ta, tb) of the 8 x 8 lane grid
    // keeps an RT x RT block of pair accumulators
    constexpr int NR = 8 * RT;
    constexpr int DCP = DC + 1;
    __shared__ double Es[NR * DCP];
    __shared__ double Ss[NR * DCP];
    __shared__ int64_t rowE[NR], rowS[NR];
    __shared__ unsigned char okE[NR], okS[NR];

    const int64_t t = blockIdx.x;
    const int a0 = blockIdx.y * NR, b0 = blockIdx.z * NR;
    const int tid = threadIdx.x;
    for (int i = tid; i < NR; i += 64) {
        const int64_t a = (a0 + i < K) ? cand[t * K + a0 + i] : -1;
        const int64_t b = (b0 + i < K) ? cand[(t + 1) * K + b0 + i] : -1;
        const bool va = unit_usable(a, n_units), vb = unit_usable(b, n_units);
        okE[i] = va; okS[i] = vb;
        rowE[i] = va ? a + 1 : 0;      // unit_end_data[a]   = JCw[a+1]
        rowS[i] = vb ? b : 0;          // unit_start_data[b] = JCw[b]
    }
    __syncthreads();

    const int ta = tid >> 3, tb = tid & 7;
    double acc[RT][RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j) acc[i][j] = 0.0;

    // register-staged gather: the next column chunk's global loads are in flight while the
    // current chunk is being accumulated (DC doubles = 128 contiguous bytes per row and chunk)
    constexpr int PER = (NR * (DC / 2) + 63) / 64;       // double2 loads per lane and matrix
    double2 pe[PER], ps[PER];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid + k * 64;
            if (e < NR * (DC / 2)) {
                const int r = e / (DC / 2), c = (e % (DC / 2)) * 2;
                pe[k] = *reinterpret_cast<const double2 *>(JCw + rowE[r] * Djpad + c0 + c);
                ps[k] = *reinterpret_cast<const double2 *>(JCw + rowS[r] * Djpad + c0 + c);
            }
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < Djpad; c0 += DC) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid + k * 64;
            if (e < NR * (DC / 2)) {
                const int r = e / (DC / 2), c = (e % (DC / 2)) * 2;
                Es[r * DCP + c] = pe[k].x; Es[r * DCP + c + 1] = pe[k].y;
                Ss[r * DCP + c] = ps[k].x; Ss[r * DCP + c + 1] = ps[k].y;
            }
        }
        __syncthreads();
        if (c0 + DC < Djpad) fetch(c0 + DC);
#pragma unroll 2
        for (int c = 0; c < DC; ++c) {
            double ev[RT], sv[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) ev[i] = Es[(ta * RT + i) * DCP + c];
#pragma unroll
            for (int j = 0; j < RT; ++j) sv[j] = Ss[(tb * RT + j) * DCP + c];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < RT; ++j) {
                    const double d = __dsub_rn(ev[i], sv[j]);
                    acc[i][j] = __dadd_rn(acc[i][j], __dmul_rn(d, d));
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j) {
            const int al = ta * RT + i, bl = tb * RT + j;
            const int a = a0 + al, b = b0 + bl;
            if (a < K && b < K) {
                const bool ok = okE[al] && okS[bl];
                J[(t * K + a) * K + b] = ok ? __dsqrt_rn(acc[i][j]) : __builtin_inf();
            }
        }
}

template <int RT, int DC>
static void launch_join_t(const double *JCw, int Djpad, int64_t n_units, const int64_t *cand,
                          int64_t T, int K, double *J, hipStream_t s)
{
    const int nb = (K + 8 * RT - 1) / (8 * RT);
    hipLaunchKernelGGL((join_cost_kernel<RT, DC>), dim3((unsigned)(T - 1), nb, nb), dim3(64), 0, s,
                       JCw, Djpad, n_units, cand, T, K, J);
}

void launch_join_costs(const double *JCw, int Djpad, int /*Dj*/, int64_t n_units,
                       const int64_t *cand, int64_t T, int K, double *J, hipStream_t s)
{
    if (T < 2) return;
    if (K <= 16) launch_join_t<2, 16>(JCw, Djpad, n_units, cand, T, K, J, s);
    else if (K <= 32) launch_join_t<4, 16>(JCw, Djpad, n_units, cand, T, K, J, s);
    else if (K <= 40 || (K > 56 && K <= 80)) launch_join_t<5, 16>(JCw, Djpad, n_units, cand, T, K, J, s);
    else launch_join_t<7, 16>(JCw, Djpad, n_units, cand, T, K, J, s);
}

// ---------------------------------------------------------------------------
// Viterbi:  delta_0[k] = tdist[0,k]
//           delta_t[k] = tdist[t,k] + min_k' ( delta_{t-1}[k'] + J[t-1,k',k] )
// ties: lowest k', then lowest final k  (oracle/snk_oracle.py: viterbi)
// ---------------------------------------------------------------------------
#define VIT_KPP 16
// PF = true : K <= 128, each thread owns <= 16 predecessors and prefetches the next slab
// PF = false: larger K, predecessors are read from L2 inside the step
template <bool PF>
__global__ void __launch_bounds__(1024)
viterbi_dp_kernel(const int64_t *__restrict__ cand, const double *__restrict__ tdist,
                  const double *__restrict__ J, int64_t T, int K, int64_t n_units, int KP,
                  int parts, int kpp, int bp_in_lds, unsigned char *__restrict__ bp_global,
                  int64_t *__restrict__ path, int64_t *__restrict__ path_len,
                  double *__restrict__ cost)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *delta = reinterpret_cast<double *>(smem);              // [KP]
    double *pval = delta + KP;                                     // [parts][KP]
    int *parg = reinterpret_cast<int *>(pval + (size_t)parts * KP);// [parts][KP]
    unsigned char *bp = bp_in_lds
        ? reinterpret_cast<unsigned char *>(parg + (size_t)parts * KP) : bp_global;  // [T][K]
    __shared__ int final_slot;

    const int tid = threadIdx.x;
    const int k = tid % KP, part = tid / KP;
    const bool active = (k < K) && (part < parts);
    const int kp0 = part * kpp;
    const double inf = __builtin_inf();

    if (T < 2) {                      // the reference's J has no states for T < 2
        if (tid == 0) { *path_len = 0; *cost = inf; }
        return;
    }
    if (part == 0) {
        double v = inf;
        if (k < K && unit_usable(cand[k], n_units)) v = tdist[k];
        delta[k] = v;
    }
    double jreg[VIT_KPP];
    if (PF && active) {
#pragma unroll
        for (int i = 0; i < VIT_KPP; ++i) {
            const int kp = kp0 + i;
            jreg[i] = (i < kpp && kp < K) ? J[(int64_t)kp * K + k] : inf;
        }
    }
    __syncthreads();

    for (int64_t t = 1; t < T; ++t) {
        double best = inf;
        int arg = kp0;
        double td = 0.0;
        bool ok_k = false;
        double jnext[VIT_KPP];
        if (active) {
            if (PF && t + 1 < T) {     // next slab column: independent of delta
#pragma unroll
                for (int i = 0; i < VIT_KPP; ++i) {
                    const int kp = kp0 + i;
                    jnext[i] = (i < kpp && kp < K) ? J[(t * K + kp) * K + k] : inf;
                }
            }
            if (part == 0) {
                td = tdist[t * K + k];
                ok_k = unit_usable(cand[t * K + k], n_units);
            }
            if (PF) {
#pragma unroll
                for (int i = 0; i < VIT_KPP; ++i) {
                    const int kp = kp0 + i;
                    if (i < kpp && kp < K) {
                        const double tot = __dadd_rn(delta[kp], jreg[i]);
                        if (tot < best) { best = tot; arg = kp; }
                    }
                }
            } else {
                const double *Jt = J + (t - 1) * K * K;
                for (int i = 0; i < kpp; ++i) {
                    const int kp = kp0 + i;
                    if (kp < K) {
                        const double tot = __dadd_rn(delta[kp], Jt[(int64_t)kp * K + k]);
                        if (tot < best) { best = tot; arg = kp; }
                    }
                }
            }
            pval[part * KP + k] = best;
            parg[part * KP + k] = arg;
        }
        __syncthreads();
        if (active && part == 0) {
            for (int p = 1; p < parts; ++p) {
                const double v = pval[p * KP + k];
                if (v < best) { best = v; arg = parg[p * KP + k]; }
            }
            delta[k] = ok_k ? __dadd_rn(td, best) : inf;
            bp[t * K + k] = (unsigned char)arg;
        }
        if (PF && active && t + 1 < T) {
#pragma unroll
            for (int i = 0; i < VIT_KPP; ++i) jreg[i] = jnext[i];
        }
        __syncthreads();
    }

    if (tid == 0) {
        double best = inf;
        int slot = 0;
        for (int kk = 0; kk < K; ++kk)
            if (delta[kk] < best) { best = delta[kk]; slot = kk; }
        if (best == inf) { *path_len = 0; *cost = inf; final_slot = -1; }
        else { *path_len = T; *cost = best; final_slot = slot; }
    }
    if (!bp_in_lds) __threadfence();
    __syncthreads();
    if (tid == 0 && final_slot >= 0) {
        int slot = final_slot;
        for (int64_t t = T - 1; t >= 0; --t) {
            path[t] = cand[t * K + slot];
            if (t > 0) slot = bp[t * K + slot];
        }
    }
}

void launch_viterbi_dp(const int64_t *cand, const double *tdist, const double *J, int64_t T, int K,
                       int64_t n_units, unsigned char *bp_global, int64_t *path, int64_t *path_len,
                       double *cost, hipStream_t s)
{
    int KP = 64;
    while (KP < K) KP <<= 1;                // K <= 256
    const int parts = 1024 / KP;
    const int kpp = (K + parts - 1) / parts;
    const bool pf = (kpp <= VIT_KPP);
    const size_t base = (size_t)KP * 8 + (size_t)parts * KP * 12;
    const size_t bp_bytes = (size_t)T * K;
    const int bp_in_lds = (base + bp_bytes + 64 <= 150 * 1024) ? 1 : 0;
    const size_t shmem = base + (bp_in_lds ? bp_bytes : 0);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&viterbi_dp_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024));
        hipFuncSetAttribute(reinterpret_cast<const void *>(&viterbi_dp_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024));
        attr_set = true;
    }
    if (pf)
        hipLaunchKernelGGL(viterbi_dp_kernel<true>, dim3(1), dim3(1024), shmem, s, cand, tdist, J,
                           T, K, n_units, KP, parts, kpp, bp_in_lds, bp_global, path, path_len, cost);
    else
        hipLaunchKernelGGL(viterbi_dp_kernel<false>, dim3(1), dim3(1024), shmem, s, cand, tdist, J,
                           T, K, n_units, KP, parts, kpp, bp_in_lds, bp_global, path, path_len, cost);
}

}  // namespace snk
