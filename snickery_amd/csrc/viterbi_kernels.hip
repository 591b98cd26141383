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
// viterbi_dp_kernel: one workgroup per utterance walks the trellis; the K x K slabs of the next
//   steps are in flight into registers, the minimum over predecessors finishes inside the
//   wavefront, back-pointers stay in LDS.
#include "snk_internal.h"
#include <float.h>

namespace snk {

__device__ __forceinline__ bool unit_usable(int64_t id, int64_t n_units)
{
    // mini = 1, maxi = data_frames - 1 and the -1 padding value (synth_halfphone.py:3238-3268)
    return id >= 1 && id < n_units - 1;
}

// One wavefront computes an (8 RTA) x (8 RTB) block of the K x K matrix of one column pair:
// thread (ta, tb) of the 8 x 8 lane grid keeps an RTA x RTB block of pair accumulators.
template <int RTA, int RTB, int DC>
__device__ __forceinline__ void join_tile(const double *__restrict__ JCw, int Djpad, int64_t n_units,
                                          const int64_t *__restrict__ cand, int64_t t, int K, int a0, int b0,
                                          double *__restrict__ J, double *Es, double *Ss, int64_t *rowE,
                                          int64_t *rowS, unsigned char *okE, unsigned char *okS)
{
    constexpr int NRA = 8 * RTA, NRB = 8 * RTB;
    constexpr int DCP = DC + 1;
    static_assert(DC == 16, "one wave-wide double2 load covers 8 rows x 16 columns");
    const int tid = threadIdx.x;
    for (int i = tid; i < NRA; i += 64) {
        const int64_t a = (a0 + i < K) ? cand[t * K + a0 + i] : -1;
        const bool va = unit_usable(a, n_units);
        okE[i] = va;
        rowE[i] = (va ? a + 1 : 0) * Djpad;    // unit_end_data[a]   = JCw[a+1]  (element offset)
    }
    for (int i = tid; i < NRB; i += 64) {
        const int64_t b = (b0 + i < K) ? cand[(t + 1) * K + b0 + i] : -1;
        const bool vb = unit_usable(b, n_units);
        okS[i] = vb;
        rowS[i] = (vb ? b : 0) * Djpad;        // unit_start_data[b] = JCw[b]
    }
    __syncthreads();

    const int ta = tid >> 3, tb = tid & 7;
    double acc[RTA][RTB];
#pragma unroll
    for (int i = 0; i < RTA; ++i)
#pragma unroll
        for (int j = 0; j < RTB; ++j) acc[i][j] = 0.0;

    // register-staged gather: the next column chunk's global loads are in flight while the
    // current chunk is being accumulated (DC doubles = 128 contiguous bytes per row and chunk).
    // Load k of a lane covers row 8k + lane/8, columns 2*(lane%8)..+1 of the chunk: RTA (RTB)
    // unconditional, independent loads per lane and matrix.
    const int lr = tid >> 3, lc = (tid & 7) * 2;
    double2 pe[RTA], ps[RTB];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int k = 0; k < RTA; ++k)
            pe[k] = *reinterpret_cast<const double2 *>(JCw + rowE[8 * k + lr] + c0 + lc);
#pragma unroll
        for (int k = 0; k < RTB; ++k)
            ps[k] = *reinterpret_cast<const double2 *>(JCw + rowS[8 * k + lr] + c0 + lc);
    };
    fetch(0);
    for (int c0 = 0; c0 < Djpad; c0 += DC) {
#pragma unroll
        for (int k = 0; k < RTA; ++k) {
            Es[(8 * k + lr) * DCP + lc] = pe[k].x; Es[(8 * k + lr) * DCP + lc + 1] = pe[k].y;
        }
#pragma unroll
        for (int k = 0; k < RTB; ++k) {
            Ss[(8 * k + lr) * DCP + lc] = ps[k].x; Ss[(8 * k + lr) * DCP + lc + 1] = ps[k].y;
        }
        __syncthreads();
        if (c0 + DC < Djpad) fetch(c0 + DC);
#pragma unroll 2
        for (int c = 0; c < DC; ++c) {
            double ev[RTA], sv[RTB];
#pragma unroll
            for (int i = 0; i < RTA; ++i) ev[i] = Es[(ta * RTA + i) * DCP + c];
#pragma unroll
            for (int j = 0; j < RTB; ++j) sv[j] = Ss[(tb * RTB + j) * DCP + c];
#pragma unroll
            for (int i = 0; i < RTA; ++i)
#pragma unroll
                for (int j = 0; j < RTB; ++j) {
                    const double d = __dsub_rn(ev[i], sv[j]);
                    acc[i][j] = __dadd_rn(acc[i][j], __dmul_rn(d, d));
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RTA; ++i)
#pragma unroll
        for (int j = 0; j < RTB; ++j) {
            const int al = ta * RTA + i, bl = tb * RTB + j;
            const int a = a0 + al, b = b0 + bl;
            if (a < K && b < K) {
                const bool ok = okE[al] && okS[bl];
                J[(t * K + a) * K + b] = ok ? __dsqrt_rn(acc[i][j]) : __builtin_inf();
            }
        }
}

// The K candidates of a column are cut into n_tiles row blocks: n_tiles-1 of 8*RTM rows and a
// last one of 8*RTL >= the remainder (K=100: 56 + 48 rows instead of 2 x 56: 8% padding, not 25%).
// Work items are fine-grained ((T-1) * n_tiles^2 single-wave workgroups) and balance over the chip.
template <int RTM, int RTL, int DC>
__global__ void __launch_bounds__(64)
join_cost_kernel(const double *__restrict__ JCw, int Djpad, int64_t n_units,
                 const int64_t *__restrict__ cand, int64_t T, int K, int n_tiles, double *__restrict__ J)
{
    constexpr int RTX = (RTM > RTL) ? RTM : RTL;
    constexpr int NR = 8 * RTX;
    __shared__ double Es[NR * (DC + 1)];
    __shared__ double Ss[NR * (DC + 1)];
    __shared__ int64_t rowE[NR], rowS[NR];
    __shared__ unsigned char okE[NR], okS[NR];
    const int64_t t = blockIdx.x;
    const bool lastA = (int)blockIdx.y == n_tiles - 1, lastB = (int)blockIdx.z == n_tiles - 1;
    const int a0 = blockIdx.y * 8 * RTM, b0 = blockIdx.z * 8 * RTM;
#define SNK_TILE(A_, B_) join_tile<A_, B_, DC>(JCw, Djpad, n_units, cand, t, K, a0, b0, J, Es, Ss, rowE, rowS, okE, okS)
    if (RTM == RTL) SNK_TILE(RTL, RTL);
    else if (lastA && lastB) SNK_TILE(RTL, RTL);
    else if (lastA) SNK_TILE(RTL, RTM);
    else if (lastB) SNK_TILE(RTM, RTL);
    else SNK_TILE(RTM, RTM);
#undef SNK_TILE
}

template <int RTM, int RTL, int DC>
static void launch_join_t(const double *JCw, int Djpad, int64_t n_units, const int64_t *cand,
                          int64_t T, int K, int n_tiles, double *J, hipStream_t s)
{
    hipLaunchKernelGGL((join_cost_kernel<RTM, RTL, DC>), dim3((unsigned)(T - 1), n_tiles, n_tiles), dim3(64), 0, s,
                       JCw, Djpad, n_units, cand, T, K, n_tiles, J);
}

// cand holds T rows; J gets T-1 slabs (slab r: rows r and r+1).  A batch passes the concatenated
// rows of several utterances: the slab between two utterances is computed and never read.
void launch_join_costs(const double *JCw, int Djpad, int /*Dj*/, int64_t n_units,
                       const int64_t *cand, int64_t T, int K, double *J, hipStream_t s)
{
    if (T < 2) return;
    const int n_tiles = (K + 55) / 56;
    const int last = K - 56 * (n_tiles - 1);
    const int rtl = (last + 7) / 8;
    if (n_tiles == 1) {
        if (rtl <= 2) launch_join_t<2, 2, 16>(JCw, Djpad, n_units, cand, T, K, 1, J, s);
        else if (rtl <= 4) launch_join_t<4, 4, 16>(JCw, Djpad, n_units, cand, T, K, 1, J, s);
        else if (rtl == 5) launch_join_t<5, 5, 16>(JCw, Djpad, n_units, cand, T, K, 1, J, s);
        else if (rtl == 6) launch_join_t<6, 6, 16>(JCw, Djpad, n_units, cand, T, K, 1, J, s);
        else launch_join_t<7, 7, 16>(JCw, Djpad, n_units, cand, T, K, 1, J, s);
    } else {
        if (rtl <= 2) launch_join_t<7, 2, 16>(JCw, Djpad, n_units, cand, T, K, n_tiles, J, s);
        else if (rtl <= 4) launch_join_t<7, 4, 16>(JCw, Djpad, n_units, cand, T, K, n_tiles, J, s);
        else if (rtl == 5) launch_join_t<7, 5, 16>(JCw, Djpad, n_units, cand, T, K, n_tiles, J, s);
        else if (rtl == 6) launch_join_t<7, 6, 16>(JCw, Djpad, n_units, cand, T, K, n_tiles, J, s);
        else launch_join_t<7, 7, 16>(JCw, Djpad, n_units, cand, T, K, n_tiles, J, s);
    }
}

// ---------------------------------------------------------------------------
// Viterbi:  delta_0[k] = tdist[0,k]
//           delta_t[k] = tdist[t,k] + min_k' ( delta_{t-1}[k'] + J[t-1,k',k] )
// ties: lowest k', then lowest final k  (oracle/snk_oracle.py: viterbi)
//
// One workgroup per utterance, ceil(K/16) wavefronts.  A wavefront owns 16 columns k; its four
// 16-lane rows split the predecessors: row p takes k' in [p*KPM, (p+1)*KPM).  So the minimum over
// k' finishes with two cross-row exchanges inside the wavefront, and a step costs ONE barrier
// (delta is double-buffered in LDS).  Every thread keeps its KPM slab entries of the next NB steps
// in flight in registers: a single workgroup pulls 8*K*K bytes per step through one CU, and without
// several slabs in flight every step would pay a full memory round trip.  Slabs are read with
// buffer loads (one VGPR offset per thread, the slab/row part in an SGPR).  Predecessors k' >= K
// need no predicate: their delta stays +inf and an out-of-range buffer load returns 0.
// ---------------------------------------------------------------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// FST32 (option viterbi_weights 1): the float32 chain of the reference's OpenFST lattices instead of the float64 recursion
// (fst_functions_wrapped.py:47,201: every arc weight is parsed into a float32 tropical weight; :368,389: compose adds the
// T arc's and the J arc's weights, shortestpath accumulates from the start state -- oracle/snk_oracle.py _viterbi_fst32):
//           acc_0[k] = 0                                     (free epsilon entry, :195-196)
//           acc_t[k] = min_k' fl32( acc_{t-1}[k'] + fl32( fl32(tdist[t-1,k']) + fl32(J[t-1,k',k]) ) )
//           final[k] = fl32( acc_{T-1}[k] + fl32(tdist[T-1,k]) )   (the exit arc carries the last target cost)
// same tie rule.  delta then holds float32 values (widened), tdp the float32 target costs of the previous row.
template <int KPM, int NB, int NTH, bool BPL, bool FST32>
__global__ void __launch_bounds__(NTH)
viterbi_dp_kernel(const int64_t *__restrict__ cand_all, const double *__restrict__ tdist_all,
                  const double *__restrict__ J_all, const DpBatch batch, int K, int64_t n_units, int KP,
                  unsigned char *__restrict__ bp_all, int64_t *__restrict__ path_all,
                  int64_t *__restrict__ path_len_all, double *__restrict__ cost_all)
{
    // rows [off[u], off[u+1]) of the group's matrices; J row r holds the costs between rows r and
    // r+1 (the row joining two utterances is unused)
    const int64_t r0 = batch.off[blockIdx.x];
    const int64_t T = batch.off[blockIdx.x + 1] - r0;
    const int64_t *__restrict__ cand = cand_all + r0 * K;
    const double *__restrict__ tdist = tdist_all + r0 * K;
    const double *__restrict__ J = J_all + r0 * K * K;
    unsigned char *__restrict__ bp_global = bp_all + r0 * K;
    int64_t *__restrict__ path = path_all + r0;
    int64_t *__restrict__ path_len = path_len_all + batch.first + blockIdx.x;
    double *__restrict__ cost = cost_all + batch.first + blockIdx.x;

    extern __shared__ __align__(16) unsigned char smem[];
    double *delta = reinterpret_cast<double *>(smem);              // [2][KP], KP >= 4*KPM
    // back-pointers [T][K]: in LDS when they fit, else in global memory (two typed pointers: a
    // flat store would turn every later wait into a full vmcnt(0)/lgkmcnt(0))
    float *tdp = reinterpret_cast<float *>(delta + 2 * KP);        // FST32: [2][KP] float32 target costs of a row
    unsigned char *bp_lds = reinterpret_cast<unsigned char *>(delta + 2 * KP) + (FST32 ? 2 * KP * 4 : 0);
    __shared__ int final_slot;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int k = (tid >> 6) * 16 + (lane & 15);
    const int part = lane >> 4;
    const int kp0 = part * KPM;
    const bool col = k < K;
    const bool lead = col && part == 0;
    const double inf = __builtin_inf();

    if (T < 2) {                      // the reference's J has no states for T < 2
        if (tid == 0) { *path_len = 0; *cost = inf; }
        return;
    }
    for (int i = tid; i < 2 * KP; i += (int)blockDim.x) { delta[i] = inf; if constexpr (FST32) tdp[i] = 0.f; }
    __syncthreads();
    if (lead) {
        if constexpr (FST32) { delta[k] = unit_usable(cand[k], n_units) ? 0.0 : inf; tdp[k] = (float)tdist[k]; }
        else delta[k] = unit_usable(cand[k], n_units) ? tdist[k] : inf;
    }

    // idle columns (k >= K) run the same convergent code with an out-of-range offset: their loads
    // return 0 without touching memory and their results are never stored
    const int voff = col ? (kp0 * K + k) * 8 : 0x7ffffff8;
    double jb[NB][KPM];
    auto load_slab = [&](int64_t slab, double (&dst)[KPM]) {       // slab t-1 feeds step t
        // one descriptor per slab (scalar arithmetic): offsets stay 32-bit for any T
        const __amdgpu_buffer_rsrc_t jres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double *>(J + slab * K * K), 0, K * K * 8, 0x00020000);
#pragma unroll
        for (int i = 0; i < KPM; ++i)
            dst[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(jres, voff, i * K * 8, 0));
    };
    // target cost and candidate id of the column NB steps ahead: plain independent loads staged
    // like the slabs and combined at the point of use (a dependent load would force vmcnt(0) and
    // drain the slab pipeline every step)
    double td_raw[NB];
    int64_t id_raw[NB];
    auto load_target = [&](int64_t t, double &td, int64_t &id) {
        const int64_t tn = (t < T ? t : T - 1) * K + (col ? k : 0);
        td = tdist[tn];
        id = cand[tn];
    };
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        load_slab(s < T - 2 ? s : T - 2, jb[s]);
        load_target(1 + s, td_raw[s], id_raw[s]);
    }
    __syncthreads();

    // One loop, NB steps per trip, no prologue/epilogue copies of the register stages: after its
    // slab is consumed a stage is refilled with the slab NB steps ahead (index clamped to the last
    // slab: a clamped re-load is never consumed), and the up to NB-1 surplus steps of the last trip
    // run with their stores switched off.
    auto step = [&](int64_t t, double (&jr)[KPM], double &tdr, int64_t &idr) {
        const bool valid = t < T;                       // uniform
        const double *dprev = delta + ((t - 1) & 1) * KP;
        double *dcur = delta + (t & 1) * KP;
        const double td = unit_usable(idr, n_units) ? tdr : inf;
        const float td32 = (float)tdr;
        load_target(t + NB, tdr, idr);
        double best = inf;
        int arg = 0;
#pragma unroll
        for (int i = 0; i < KPM; ++i) {
            double tot;
            if constexpr (FST32) {
                const float arcw = tdp[((t - 1) & 1) * KP + kp0 + i] + (float)jr[i];        // composed arc weight
                tot = (double)((float)dprev[kp0 + i] + arcw);
            } else tot = __dadd_rn(dprev[kp0 + i], jr[i]);
            if (tot < best) { best = tot; arg = i; }
        }
        arg += kp0;
        load_slab(t - 1 + NB < T - 2 ? t - 1 + NB : T - 2, jr);
        // minimum over the four predecessor ranges (lane rows); equal values keep the lower k'
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
            const double ob = __shfl_xor(best, m, 64);
            const int oa = __shfl_xor(arg, m, 64);
            if (ob < best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        if (lead && valid) {
            if constexpr (FST32) { dcur[k] = td < inf ? best : inf; tdp[(t & 1) * KP + k] = td32; }
            else dcur[k] = __dadd_rn(td, best);
            if constexpr (BPL) bp_lds[t * K + k] = (unsigned char)arg;
            else bp_global[t * K + k] = (unsigned char)arg;
        }
        __syncthreads();
    };

    for (int64_t t = 1; t < T; t += NB) {
#pragma unroll
        for (int s = 0; s < NB; ++s) step(t + s, jb[s], td_raw[s], id_raw[s]);
    }

    const double *dlast = delta + ((T - 1) & 1) * KP;
    if (tid == 0) {
        double best = inf;
        int slot = 0;
        for (int kk = 0; kk < K; ++kk) {
            double v = dlast[kk];
            if constexpr (FST32) v = (double)((float)v + tdp[((T - 1) & 1) * KP + kk]);
            if (v < best) { best = v; slot = kk; }
        }
        if (best == inf) { *path_len = 0; *cost = inf; final_slot = -1; }
        else { *path_len = T; *cost = best; final_slot = slot; }
    }
    if (!BPL) __threadfence();
    __syncthreads();
    if (tid == 0 && final_slot >= 0) {
        int slot = final_slot;
        for (int64_t t = T - 1; t >= 0; --t) {
            path[t] = cand[t * K + slot];
            if (t > 0) {
                if constexpr (BPL) slot = bp_lds[t * K + slot];
                else slot = bp_global[t * K + slot];
            }
        }
    }
}

// Launches one workgroup per utterance; `off` (n+1 row offsets into the group's matrices) travels
// in the kernel arguments.  path_len / cost are indexed by first_utt + i.
void launch_viterbi_dp_batch(const int64_t *cand, const double *tdist, const double *J, const int64_t *off,
                             int n_utts, int first_utt, int K, int64_t n_units, unsigned char *bp_global,
                             int64_t *path, int64_t *path_len, double *cost, hipStream_t s, bool fst32)
{
    for (int u0 = 0; u0 < n_utts; u0 += DpBatch::MAX) {
        const int n = (n_utts - u0 < DpBatch::MAX) ? n_utts - u0 : DpBatch::MAX;
        DpBatch batch;
        int64_t T = 0;
        for (int i = 0; i <= n; ++i) batch.off[i] = off[u0 + i];
        for (int i = 0; i < n; ++i) T = (off[u0 + i + 1] - off[u0 + i] > T) ? off[u0 + i + 1] - off[u0 + i] : T;
        batch.first = first_utt + u0;
        // K <= 64: 4 x 16 predecessors, 4 slabs in flight; K <= 100: 4 x 25, 3 slabs; K <= 128:
        // 4 x 32, 3 slabs; K <= 208: 4 x 52 -- 13 wavefronts leave 128 registers per thread, which
        // hold a single slab
        int variant, kpm;
        if (K <= 64) { variant = 0; kpm = 16; }
        else if (K <= 100) { variant = 1; kpm = 25; }
        else if (K <= 128) { variant = 2; kpm = 32; }
        else { variant = 3; kpm = 52; }
        const int KP = 4 * kpm;
        const int nth = 64 * ((K + 15) / 16);
        const size_t base = (size_t)2 * KP * 8 + (fst32 ? (size_t)2 * KP * 4 : 0);
        const size_t bp_bytes = (size_t)T * K;
        const int bp_in_lds = (base + bp_bytes + 64 <= 150 * 1024) ? 1 : 0;
        const size_t shmem = base + (bp_in_lds ? bp_bytes : 0);
#define SNK_DP1(KPM_, NB_, NTH_, BPL_, F32_)                                                       \
    {                                                                                              \
        static size_t attr_set[32] = {0};                                                    \
        lds_attr_ensure(attr_set, 150 * 1024, [] {                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&viterbi_dp_kernel<KPM_, NB_, NTH_, BPL_, F32_>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024)); });  \
        hipLaunchKernelGGL((viterbi_dp_kernel<KPM_, NB_, NTH_, BPL_, F32_>), dim3(n), dim3(nth), shmem, s, cand, \
                           tdist, J, batch, K, n_units, KP, bp_global, path, path_len, cost);      \
    }
#define SNK_DP(KPM_, NB_, NTH_)                                                                    \
    {                                                                                              \
        if (fst32) { if (bp_in_lds) SNK_DP1(KPM_, NB_, NTH_, true, true) else SNK_DP1(KPM_, NB_, NTH_, false, true) }   \
        else { if (bp_in_lds) SNK_DP1(KPM_, NB_, NTH_, true, false) else SNK_DP1(KPM_, NB_, NTH_, false, false) }       \
    }
        if (variant == 0) SNK_DP(16, 4, 256)
        else if (variant == 1) SNK_DP(25, 3, 448)
        else if (variant == 2) SNK_DP(32, 3, 512)
        else SNK_DP(52, 1, 832)
#undef SNK_DP1
#undef SNK_DP
    }
}

void launch_viterbi_dp(const int64_t *cand, const double *tdist, const double *J, int64_t T, int K,
                       int64_t n_units, unsigned char *bp_global, int64_t *path, int64_t *path_len,
                       double *cost, hipStream_t s, bool fst32)
{
    const int64_t off[2] = {0, T};
    launch_viterbi_dp_batch(cand, tdist, J, off, 1, 0, K, n_units, bp_global, path, path_len, cost, s, fst32);
}


// Results of a batch -> page-locked host memory by a KERNEL (stores over PCIe) instead of DMA copies: a copy queued behind events
// that are minutes of GPU time away -- the batch's last recursions -- sits in the DMA engine's queue until they fire, and the
// next batch's upload of query rows (same engine family) waited behind it (B4: 3-4 ms of a 9.6 ms step inside `h2d_queries`,
// profiles/r06k_b4.log).  Five segments, 16-byte stores where aligned.
struct D2HSegs { void *dst[5]; const void *src[5]; unsigned long long bytes[5]; };
__global__ void __launch_bounds__(256)
results_to_host_kernel(D2HSegs sg)
{
    for (int i = 0; i < 5; ++i) {
        const unsigned long long n = sg.bytes[i];
        if (!n) continue;
        const unsigned long long n16 = ((((unsigned long long)sg.dst[i] | (unsigned long long)sg.src[i]) & 15ull) == 0) ? n / 16 : 0;
        const uint4 *s4 = static_cast<const uint4 *>(sg.src[i]);
        uint4 *d4 = static_cast<uint4 *>(sg.dst[i]);
        for (unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x; k < n16; k += (unsigned long long)gridDim.x * 256) d4[k] = s4[k];
        const unsigned char *s1 = static_cast<const unsigned char *>(sg.src[i]);
        unsigned char *d1 = static_cast<unsigned char *>(sg.dst[i]);
        for (unsigned long long k = n16 * 16 + (unsigned long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (unsigned long long)gridDim.x * 256) d1[k] = s1[k];
    }
}

void launch_results_to_host(void *const *dst, const void *const *src, const size_t *bytes, int n, hipStream_t s)
{
    D2HSegs sg{};
    for (int i = 0; i < 5; ++i) { sg.dst[i] = i < n ? dst[i] : nullptr; sg.src[i] = i < n ? src[i] : nullptr; sg.bytes[i] = i < n ? bytes[i] : 0; }
    hipLaunchKernelGGL(results_to_host_kernel, dim3(16), dim3(256), 0, s, sg);
}

}  // namespace snk
