// Greedy joint search, third form: the database RESIDENT in LDS for the whole launch.
//
// Replaces the same reference code as greedy_kernels.hip / greedy32_kernels.hip (greedy_joint_search,
// script/synth_simple.py:458-503; get_tree_for_greedy_search :190-229) for voices whose windowed join matrix fits
// the chip's LDS: 256 compute units x 152 KB hold 65 536 windows x 151 float32 join columns (the README demo voice,
// config/slt_simplified_mini.cfg).  With the target term hoisted into one float64 matrix product per
// utterance (greedy_hoist_kernels.hip) a step then touches no database byte outside LDS:
//
//   every workgroup owns 256 consecutive windows, one per thread; their join rows are loaded ONCE, lane-major
//   (xs[q][thread] = float4 of columns 4q .. 4q+3: conflict-free 16-byte LDS reads);
//   step:  reference row of the previous winner (one 604-byte read per workgroup) -> float32 (w, ref) table in LDS
//          -> 152 columns of float32 arithmetic per thread out of LDS + the window's hoisted target value
//          -> workgroup top-3 (three wavefront minima, no sort)
//          -> ONE 16-byte record per workgroup, two self-validating 8-byte granules {data, step tag} (agent-scope
//             stores), which EVERY workgroup gathers (256 x 16 B, agent-scope loads, polled until all tags match)
//          -> every workgroup takes the SAME decision from the same records: no deciding workgroup, no release, no
//             poll for the winner.
//   Two dependent trips through the fabric per step (records, reference row) instead of seven.
//
// The decision is greedy32_kernels.hip's: a float32 total lies within the proven E(d) of the canonical float64 total;
// only windows inside tau = M + 2 E(tau) can be the exact nearest neighbour; one such window wins outright, several
// are settled by their canonical float64 totals (lowest index on exact ties), computed by every workgroup for itself.
// A third window of ONE workgroup inside tau (mass duplicates: digital silence) is beyond what a record carries:
// the launch ends and the caller finishes the utterance on greedy32_kernels.hip's scan, as for its own undecidable
// steps.  search_epsilon >= 1e-3: the float32 minimum is the answer (see greedy32_kernels.hip).
// The records' hand-off is the form "8-byte agent-scope atomics on both sides" of a data-tagged granule on hipMalloc
// memory: nothing but the granule itself is handed over, so no fence and no drain is involved.
#include "greedy32_device.h"

namespace snk {

#define GRES_T 256                                 // threads per workgroup = windows per workgroup
#define GRES_STALL_TICKS 300000000ull              // 3 s of the 100 MHz clock
#define GRES_MAXCAND 64                            // windows inside the bound that one step settles exactly

struct GresRec { unsigned long long a, b; };
// v1 <= v2 <= v3: order-preserving uint32 images (gres_image) of the workgroup's three smallest float32 totals
// granule a: v1 | thread of the best window << 32 | thread of the second << 40 | tag << 48
// granule b: v2 | (v3 - v1 as a float32 truncated to its upper 16 bits: at or BELOW the true distance, conservative) << 32 | tag << 48

// order-preserving map float32 -> uint32 and back
__device__ __forceinline__ unsigned int gres_image(float f)
{
    const unsigned int b = __builtin_bit_cast(unsigned int, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float gres_value(unsigned int u)
{
    return __builtin_bit_cast(float, (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__device__ __forceinline__ unsigned int gres_dpp_min_u32(unsigned int v)
{
    // wavefront minimum by DPP: quads, half rows, rows, then row broadcasts; the result is in lane 63
#define GRES_DPP(ctrl_, rmask_)                                                                       \
    {                                                                                                 \
        const unsigned int o = (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl_, rmask_, 0xf, false); \
        v = o < v ? o : v;                                                                            \
    }
    GRES_DPP(0xB1, 0xf)      // quad_perm [1,0,3,2]
    GRES_DPP(0x4E, 0xf)      // quad_perm [2,3,0,1]
    GRES_DPP(0x141, 0xf)     // row_half_mirror
    GRES_DPP(0x140, 0xf)     // row_mirror
    GRES_DPP(0x142, 0xa)     // row_bcast:15 into rows 1 and 3
    GRES_DPP(0x143, 0xc)     // row_bcast:31 into rows 2 and 3
#undef GRES_DPP
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ unsigned long long gres_min_u64(unsigned long long v)
{
#pragma unroll
    for (int m = 1; m <= 32; m <<= 1) {
        const unsigned long long o = __shfl_xor(v, m, 64);
        v = o < v ? o : v;
    }
    return v;
}

// Tree sums of the float64 terms of up to four windows at once (the operands of all of them requested together: the
// rows of a candidate are cold).  Any order of adding the same terms lands within n 2^-53 of the true sum, the canonical
// chain too: a window whose tree sum is beyond (1 + 4 n 2^-53) of the smallest cannot be the canonical minimum.
static __device__ void gres_tree_sums(const GreedyArgs &a, int64_t step, int64_t prev_row, bool prev_is_current,
                                      const int64_t (&ids)[4], int cnt, double (&sum)[4], int lane)
{
    const int col0 = prev_is_current ? a.cur_col0 : a.prev_col0;
    const int64_t row0 = prev_is_current ? a.cur_row0 : a.prev_row0;
    const float *rr = a.JC_unw + (row0 + (prev_row >= 0 ? prev_row : 0)) * a.Jp + col0;
    const int ncol = a.jdim + a.nep * a.Dt;
    constexpr int NJ = 9;
    double part[4] = {0.0, 0.0, 0.0, 0.0};
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
                qv[j] = a.Q[(a.q_off[0] + step * a.me + a.ep[e]) * a.Dt + c];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < cnt) x[k][j] = a.F_unw[(ids[k] + a.ep[e]) * a.Fp + c];
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
                        part[k] += __dmul_rn(d, d);
                    }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double v = part[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        sum[k] = k < cnt ? v : DBL_MAX;
    }
}

// the canonical float64 minimum among n listed windows (lowest index on exact ties), by ONE wavefront: tree sums of all of
// them first; only the windows that the tree sums cannot tell from the smallest get their canonical chain
// M / c1 / c2 / EW: the float32 minimum and the bound E(d) = c1 sqrt(d) + c2 d (+ EW for the hoisted values) the step's tau was built
// from; report != nullptr (one workgroup): every tree sum is held against them (g32_trip, the tripwire of the scan's bound)
static __device__ int64_t gres_exact_argmin(const GreedyArgs &a, int64_t step, int64_t prev_row, const int64_t *clist, int n,
                                            double *terms, int lane, double M, double c1, double c2, double EW, int64_t *report,
                                            float &trip_seen)
{
    const int ex_cols = a.jdim + a.nep * a.Dt;
    double smin = DBL_MAX;
    for (int p0 = 0; p0 < n; p0 += 4) {
        int64_t ids[4];
        const int cnt = n - p0 < 4 ? n - p0 : 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) ids[k] = k < cnt ? g32_uniform_i(clist[p0 + k]) : 0;
        double sum[4];
        gres_tree_sums(a, step, prev_row, step > 0, ids, cnt, sum, lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < cnt && lane == 0) {
                terms[ex_cols + p0 + k] = sum[k];          // (behind the term array: up to GRES_MAXCAND sums)
                if (report) g32_trip(report, M, sum[k], c1 * sqrt(sum[k]) * (1.0 + 1e-12) + c2 * sum[k] + EW, trip_seen);
            }
            smin = sum[k] < smin ? sum[k] : smin;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const double thr = smin * (1.0 + 4.0 * (double)ex_cols * 1.1102230246251565e-16 * 1.01) + 1e-300;
    int n_surv = 0;
    int64_t only = -1;
    for (int p = 0; p < n; ++p)
        if (terms[ex_cols + p] <= thr) { ++n_surv; only = clist[p]; }
    if (n_surv == 1) return g32_uniform_i(only);
    double dbest = DBL_MAX;
    int64_t ibest = INT64_MAX;
    for (int p = 0; p < n; ++p) {
        if (!(terms[ex_cols + p] <= thr)) continue;                              // uniform
        const int64_t id = g32_uniform_i(clist[p]);
        const double d = g32_exact_d2_wave(a, 0, step, prev_row, step > 0, id, terms, lane);
        if (d < dbest || (d == dbest && id < ibest)) { dbest = d; ibest = id; }
    }
    return ibest;
}

// JQ: float4 columns of a join row the instance is compiled for (the weights live in registers, the loop over the columns
// is unrolled: the launch picks the smallest instance that holds the voice's join width; the rest is zero padding)
template <int JQ, bool EXACT>                            // EXACT: the voice's join rows have exactly JQ float4 columns (no bounds tests in the scan)
__global__ void __launch_bounds__(GRES_T, 1)
greedy_res_kernel(GreedyArgs a, int64_t nsteps, int flags, int JQ4, int tile_q, GresRec *rec, unsigned long long *rec2,
                  int64_t *path, int64_t *status, unsigned long long *trace)
{
    // optional timeline (SNK_GRES_TRACE=file): 8 stamps of the 100 MHz clock per step, workgroup 0, steps 0 .. 255
    auto stamp = [&](int64_t st, int k) {
        if (trace && st < 256 && blockIdx.x == 0 && threadIdx.x == 0) {
            trace[st * 8 + k] = __builtin_amdgcn_s_memrealtime();
            trace[2048 + st * 8 + k] = __builtin_amdgcn_s_memtime();          // shader clock
        }
    };
    const bool approx = (flags & 1) != 0;
    const bool test_stall = (flags & 256) != 0;
    const bool fenced = (flags & 512) != 0;              // cross-check mode: release / acquire fences around the hand-off (greedy32_kernels.hip)
    extern __shared__ __align__(16) char lds[];
    f32x4 *const xs = reinterpret_cast<f32x4 *>(lds);                           // [JQ4][256]
    // (written and read as floats: a float store read back through a float4 pointer is undefined behaviour, and the
    // compiler used it -- every component of a reference became the first one)
    float *const tw = reinterpret_cast<float *>(lds + (size_t)JQ4 * GRES_T * 16);    // [4 JQ4] weights (float32)
    float *const tr = tw + 4 * JQ4;                                              // [4 JQ4] references of the step
    char *const scratch = reinterpret_cast<char *>(tr + 4 * JQ4);                // 512 bytes
    double *const redd = reinterpret_cast<double *>(scratch);                    // [4] partial norms
    unsigned int *const redk = reinterpret_cast<unsigned int *>(scratch + 64);   // [4][6] wavefront top-3
    int64_t *const bcast = reinterpret_cast<int64_t *>(scratch + 192);           // [8] winner, state, tau | minimum, c1, c2, EW of the step's bound
    float trip_seen = 0.f;                                                       // tripwire of the bound (g32_trip)
    int64_t *const clist = reinterpret_cast<int64_t *>(scratch + 256);           // candidates (GRES_MAXCAND, behind: terms)
    double *const terms = reinterpret_cast<double *>(scratch + 256 + GRES_MAXCAND * 8);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned int nb = gridDim.x;
    const int64_t win = (int64_t)blockIdx.x * GRES_T + tid;
    const bool valid = win < a.Nwin;

    // ---- once: this workgroup's join rows into LDS, the float32 weights ----
    {
        const int64_t r = a.prev_row0 + (valid ? win : 0);
        const f32x4 *src = a.JT + ((size_t)(r >> 6) * tile_q) * 64 + (r & 63);
        for (int q = 0; q < JQ4; ++q) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (valid) v = src[(size_t)q * 64];
            xs[q * GRES_T + tid] = v;
        }
        if (tid < JQ4 * 4) {
            const float w = tid < a.jdim ? (float)a.wj[a.prev_col0 + tid] : 0.f;
            tw[tid] = w;
            tr[tid] = 0.f;
        }
    }
    // weight of this thread's reference column: the winner's row is read from the `current` columns, a start state's
    // from the `prev` columns (synth_simple.py:467-469,501)
    __syncthreads();
    f32x4 wreg[JQ];                                      // float32 weights of all columns: constant over the launch
#pragma unroll
    for (int q = 0; q < JQ; ++q)
        wreg[q] = q < JQ4 ? (f32x4){tw[4 * q], tw[4 * q + 1], tw[4 * q + 2], tw[4 * q + 3]} : (f32x4){0.f, 0.f, 0.f, 0.f};
    const double wref_cur = tid < a.jdim ? a.wj[a.cur_col0 + tid] : 0.0;
    const double wref_prev = tid < a.jdim ? a.wj[a.prev_col0 + tid] : 0.0;
    const int ecols = JQ4 * 4 + 6;                       // chain length of the bound: columns, the four partial sums, the hoisted value
    int64_t prev_row = a.start[0];
    const float *const Wp0 = a.W[0] + (valid ? win : 0);
    float w_next = valid ? Wp0[0] : 0.f;
    double qn2_next = a.qn2[0][0];
    const double sq_fw = sqrt(a.fwmax2);
    unsigned long long stat_rounds = 0, stat_windows = 0;
    __syncthreads();

    for (int64_t step = 0; step < nsteps; ++step) {
        const unsigned int tag = (unsigned int)(step % 65535) + 1u;
        stamp(step, 0);
        const float w_cur = w_next;
        if (step + 1 < nsteps && valid) w_next = Wp0[(size_t)(step + 1) * a.Wp];
        // ---- the step's references: row of the previous winner x weights, float32; squared norm in float64 ----
        {
            double ref = 0.0;
            if (tid < a.jdim && prev_row >= 0) {
                const bool cur = step > 0;
                const float x = a.JC_unw[((cur ? a.cur_row0 : a.prev_row0) + prev_row) * a.Jp + (cur ? a.cur_col0 : a.prev_col0) + tid];
                ref = __dmul_rn((double)x, cur ? wref_cur : wref_prev);
            }
            if (tid < JQ4 * 4) tr[tid] = (float)ref;
            double v = ref * ref;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) redd[wave] = v;
        }
        __syncthreads();
        stamp(step, 1);                                      // table built
        const double V2 = g32_uniform_d(((redd[0] + redd[1]) + redd[2]) + redd[3]);
        const double qn2 = qn2_next;                         // (requested a step ahead: a scalar-cache miss is a microsecond)
        if (step + 1 < nsteps) qn2_next = a.qn2[0][step + 1];

        // ---- scan: this thread's window (LDS) against the references (lane q of every wavefront holds float4 column q
        //      of the table: one LDS read per step, then a v_readlane per value -- as broadcast LDS reads the table was two
        //      thirds of the step's LDS traffic, and that traffic was the scan's time) and the weights (registers) ----
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
        {
            const int lq = 4 * (lane < JQ4 ? lane : 0);
            const int rl0 = __builtin_bit_cast(int, tr[lq]), rl1 = __builtin_bit_cast(int, tr[lq + 1]);
            const int rl2 = __builtin_bit_cast(int, tr[lq + 2]), rl3 = __builtin_bit_cast(int, tr[lq + 3]);
#pragma unroll
            for (int q = 0; q < JQ; ++q) {
                if (EXACT || q < JQ4) {                                      // uniform
                    const f32x4 x = xs[q * GRES_T + tid];
                    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rl0, q));
                    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rl1, q));
                    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rl2, q));
                    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rl3, q));
                    const float d0 = __builtin_fmaf(x[0], wreg[q][0], -r0), d1 = __builtin_fmaf(x[1], wreg[q][1], -r1);
                    const float d2 = __builtin_fmaf(x[2], wreg[q][2], -r2), d3 = __builtin_fmaf(x[3], wreg[q][3], -r3);
                    acc0 = __builtin_fmaf(d0, d0, acc0); acc1 = __builtin_fmaf(d1, d1, acc1);
                    acc2 = __builtin_fmaf(d2, d2, acc2); acc3 = __builtin_fmaf(d3, d3, acc3);
                }
            }
        }
        asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3));
        stamp(step, 2);                                      // scan done
        float key = valid ? ((acc0 + acc1) + (acc2 + acc3)) + w_cur : __builtin_inff();
        if (!(key == key)) key = __builtin_inff();           // a NaN total (non-finite data) never wins
        // order-preserving image of the float32 key (the hoisted value's rounding can leave a total just below zero)
        const unsigned int INF = 0xff800000u;
        const unsigned int kb = gres_image(key);
        // ---- wavefront top-3: three minima; ties keep the lowest lane = lowest window index ----
        const unsigned int m1 = gres_dpp_min_u32(kb);
        const int l1 = __builtin_ctzll(__ballot(kb == m1));
        const unsigned int kb2 = lane == l1 ? INF : kb;
        const unsigned int m2 = gres_dpp_min_u32(kb2);
        const int l2 = m2 < INF ? __builtin_ctzll(__ballot(kb2 == m2 && lane != l1)) : 0;
        const unsigned int kb3 = (lane == l2 && m2 < INF) ? INF : kb2;
        const unsigned int m3 = gres_dpp_min_u32(kb3);
        if (lane == 0) {
            redk[wave * 6 + 0] = m1; redk[wave * 6 + 1] = (unsigned int)(wave * 64 + l1);
            redk[wave * 6 + 2] = m2; redk[wave * 6 + 3] = (unsigned int)(wave * 64 + l2);
            redk[wave * 6 + 4] = m3;
        }
        __syncthreads();
        stamp(step, 3);                                      // wavefront top-3 in LDS

        if (wave == 0) {
            // ---- workgroup top-3 (the wavefronts hold ascending windows: earlier entries win ties) and the record ----
            // lane 3w + j holds wavefront w's j-th value (j = 0, 1 with its thread, j = 2 the bare third value): three minima;
            // ties keep the lowest lane = the lowest window (the wavefronts hold ascending windows)
            unsigned int cv = INF, ci = 0;
            if (lane < 12) { const int w = lane / 3, j = lane - 3 * w; cv = redk[w * 6 + 2 * j]; ci = j < 2 ? redk[w * 6 + 2 * j + 1] : 0u; }
            const unsigned int v1 = gres_dpp_min_u32(cv);
            const int p1 = __builtin_ctzll(__ballot(cv == v1));
            const unsigned int a1 = (unsigned int)__builtin_amdgcn_readlane((int)ci, p1);
            const unsigned int cv2 = lane == p1 ? INF : cv;
            const unsigned int v2 = gres_dpp_min_u32(cv2);
            const int p2 = v2 < INF ? __builtin_ctzll(__ballot(cv2 == v2 && lane != p1)) : 0;
            const unsigned int a2 = (unsigned int)__builtin_amdgcn_readlane((int)ci, p2);
            const unsigned int cv3 = (lane == p2 && v2 < INF) ? INF : cv2;
            const unsigned int v3 = gres_dpp_min_u32(cv3);
            if (lane == 0 && !(test_stall && step == 1 && blockIdx.x == 0)) {
                const unsigned long long ga = (unsigned long long)v1 | ((unsigned long long)a1 << 32) | ((unsigned long long)a2 << 40) |
                                              ((unsigned long long)tag << 48);
                // third value: its distance to the best one, truncated to eight significant bits (never above the true
                // distance; resolution where it matters: near ties)
                unsigned int d3 = 0x7f80u;                                   // +inf: no third window
                if (v3 < INF && v1 < INF) {
                    const double diff = (double)gres_value(v3) - (double)gres_value(v1);      // >= 0, exact
                    const float df = (float)diff;
                    unsigned int db = __builtin_bit_cast(unsigned int, df);
                    if ((double)df > diff) db -= 1u;                         // the conversion rounded up: one step down
                    d3 = db >> 16;
                }
                const unsigned long long gb = (unsigned long long)v2 | ((unsigned long long)d3 << 32) | ((unsigned long long)tag << 48);
                if (fenced) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                // records are double-buffered by step parity: a workgroup that has gathered step s publishes step s + 1 at
                // once, while a slower one may still be polling the records of step s -- nobody can be two steps ahead (step
                // s + 2 is published only after every record of step s + 1, hence every gather of step s, is done)
                __hip_atomic_store(&rec[(size_t)(step & 1) * nb + blockIdx.x].a, ga, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&rec[(size_t)(step & 1) * nb + blockIdx.x].b, gb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            stamp(step, 4);                                  // record published
            // constants of the bound, while the records travel: E(d) = c1 sqrt(d) + c2 d
            const double rr = sqrt(qn2) + sq_fw;
            const double EW = a.hoist_c * rr * rr;
            const double c1 = 6.0 * 5.9604644775390625e-08 * 1.01 * sqrt(V2) * (1.0 + 1e-12), c2 = (double)(ecols + 8) * 5.9604644775390625e-08;
            auto errf = [&](double d) { return c1 * sqrt(d) * (1.0 + 1e-12) + c2 * d; };       // >= g32_err(d, V2, ecols)
            // ---- gather the records of all workgroups (lane L: workgroups L, L + 64, ...) until every tag is this step's ----
            unsigned long long ra[4], rb[4];
            int state = 0;                                   // 0: go on, -1: leave (undecidable / watchdog)
            {
                const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned int b = (unsigned int)lane + 64u * q;
                        ra[q] = 0ull; rb[q] = 0ull;
                        if (b < nb) {
                            ra[q] = __hip_atomic_load(&rec[(size_t)(step & 1) * nb + b].a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            rb[q] = __hip_atomic_load(&rec[(size_t)(step & 1) * nb + b].b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = ok && (unsigned int)(ra[q] >> 48) == tag && (unsigned int)(rb[q] >> 48) == tag;
                        }
                    }
                    if (__all(ok)) break;
                    if (__builtin_amdgcn_s_memrealtime() - t_wait > GRES_STALL_TICKS) {
                        // a workgroup of the launch is not running (the device shared with another spinning launch)
                        if (lane == 0) {
                            __hip_atomic_store(&status[3], (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(status, (int64_t)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        state = -1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (fenced) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            stamp(step, 5);                                  // all records seen
            int64_t winner = -1;
            if (state == 0) {
                // ---- the decision, the same in every workgroup ----
                unsigned long long best = ~0ull;             // (v1 bits, window) of the float32 minimum: lowest index on ties
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned int b = (unsigned int)lane + 64u * q;
                    if (b < nb) {
                        const unsigned long long k = ((ra[q] & 0xffffffffull) << 32) | (unsigned long long)(b * GRES_T + (unsigned int)((ra[q] >> 32) & 0xffu));
                        best = k < best ? k : best;
                    }
                }
                best = gres_min_u64(best);
                const unsigned int mvb = (unsigned int)(best >> 32);
                const int64_t mi = (int64_t)(best & 0xffffffffull);
                const float mv = gres_value(mvb);
                if (!(mvb < INF)) {
                    state = -1;                              // nothing finite
                    if (lane == 0 && blockIdx.x == 0) __hip_atomic_store(status, (int64_t)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (approx && 4.0 * (errf((double)mv) + EW) <= 1e-3 * (double)mv) {
                    winner = mi;
                } else {
                    const double M = (double)mv + 2.0 * EW;
                    // (greedy32_kernels.hip: every iterate of tau = M + 2 E(tau) from above stays above the solution; two
                    // rounds leave 1e-10 of it to gain)
                    double tau = 2.0 * M + 64.0 * 36.0 * 3.5527136788005009e-15 * V2 + 1e-300;
                    for (int it = 0; it < 2; ++it) tau = M + 2.0 * errf(tau);
                    tau = tau * (1.0 + 1e-6) + 1e-300;
                    int nc = 0;
                    bool cov = false;
                    unsigned long long m1b[4], m2b[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned int b = (unsigned int)lane + 64u * q;
                        const float v1 = gres_value((unsigned int)ra[q]);
                        const float v2 = gres_value((unsigned int)rb[q]);
                        // (third value = best + truncated distance: at or below the workgroup's true third value)
                        const double v3 = (double)v1 + (double)__builtin_bit_cast(float, (unsigned int)((rb[q] >> 32) & 0xffffu) << 16);
                        m1b[q] = __ballot(b < nb && (double)v1 <= tau);
                        m2b[q] = __ballot(b < nb && (double)v2 <= tau);
                        nc += __popcll(m1b[q]) + __popcll(m2b[q]);
                        cov = cov || __ballot(b < nb && v3 <= tau) != 0ull;
                    }
                    if (nc == 1 && !cov) winner = mi;        // the only window inside the bound is the float32 minimum itself
                    else if (cov && nc <= GRES_MAXCAND) {
                        // a window no record carries may matter (three or more of one workgroup inside the bound: runs of
                        // near-identical frames): second round -- every workgroup publishes WHICH of its windows lie inside
                        state = 2;
                        if (lane == 0) {
                            bcast[2] = __double_as_longlong(tau);
                            // (for the tripwire of the bound in the second round: the minimum and the bound's constants)
                            bcast[3] = __double_as_longlong((double)mv); bcast[4] = __double_as_longlong(c1); bcast[5] = __double_as_longlong(c2);
                            bcast[6] = __double_as_longlong(EW);
                        }
                    } else if (cov || nc > GRES_MAXCAND) {
                        state = -1;                          // mass ties: the caller's other scan
                        if (lane == 0 && blockIdx.x == 0) {
                            // (why, for the developer: status[4..7] = candidates, third-value flag, float32 minimum, tau)
                            status[4] = nc; status[5] = cov ? 1 : 0;
                            status[6] = __double_as_longlong((double)mv); status[7] = __double_as_longlong(tau);
                        }
                        if (lane == 0 && blockIdx.x == 0) __hip_atomic_store(status, (int64_t)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        // canonical float64 totals of the windows inside the bound, lowest index on exact ties
                        int pos = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned int b = (unsigned int)lane + 64u * q;
                            const unsigned long long below = (1ull << lane) - 1ull;
                            if ((m1b[q] >> lane) & 1ull) clist[pos + __popcll(m1b[q] & below)] = (int64_t)(b * GRES_T + (unsigned int)((ra[q] >> 32) & 0xffu));
                            pos += __popcll(m1b[q]);
                            if ((m2b[q] >> lane) & 1ull) clist[pos + __popcll(m2b[q] & below)] = (int64_t)(b * GRES_T + (unsigned int)((ra[q] >> 40) & 0xffu));
                            pos += __popcll(m2b[q]);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        winner = gres_exact_argmin(a, step, prev_row, clist, nc, terms, lane, (double)mv, c1, c2, EW,
                                                   blockIdx.x == 0 ? status : nullptr, trip_seen);
                        stat_windows += (unsigned long long)nc;
                    }
                }
            }
            stamp(step, 6);                                  // decided
            if (lane == 0) { bcast[0] = winner; bcast[1] = state; }
        }
        __syncthreads();
        stamp(step, 7);
        if (bcast[1] == 2) {                                 // uniform
            // ---- second round: 256-bit membership mask of every workgroup (8 granules of 32 bits + tag), gathered by all ----
            const double tau = __longlong_as_double(bcast[2]);
            const double tm_ = __longlong_as_double(bcast[3]), tc1_ = __longlong_as_double(bcast[4]), tc2_ = __longlong_as_double(bcast[5]),
                         tew_ = __longlong_as_double(bcast[6]);
            const unsigned long long mask = __ballot(valid && (double)key <= tau);
            if (lane == 0) {
                if (fenced) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
                    __hip_atomic_store(&rec2[(size_t)blockIdx.x * 8 + 2 * wave + hh],
                                       ((mask >> (32 * hh)) & 0xffffffffull) | ((unsigned long long)tag << 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();                                 // (bcast is rewritten below)
            if (wave == 0) {
                unsigned long long g2[4][8];
                int state = 0;
                const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned int b = (unsigned int)lane + 64u * q;
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            g2[q][g] = 0ull;
                            if (b < nb) {
                                g2[q][g] = __hip_atomic_load(&rec2[(size_t)b * 8 + g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                ok = ok && (unsigned int)(g2[q][g] >> 48) == tag;
                            }
                        }
                    }
                    if (__all(ok)) break;
                    if (__builtin_amdgcn_s_memrealtime() - t_wait > GRES_STALL_TICKS) {
                        if (lane == 0) {
                            __hip_atomic_store(&status[3], (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(status, (int64_t)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        state = -1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (fenced) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                int64_t winner = -1;
                if (state == 0) {
                    int mine = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int g = 0; g < 8; ++g) mine += __popc((unsigned int)g2[q][g]);
                    int incl = mine;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int o = __shfl_up(incl, off, 64);
                        if (lane >= off) incl += o;
                    }
                    const int total = __shfl(incl, 63, 64);
                    if (total < 1 || total > GRES_MAXCAND) {
                        state = -1;                          // mass ties: the caller's other scan
                        if (lane == 0 && blockIdx.x == 0) {
                            status[4] = total; status[5] = 2; status[7] = __double_as_longlong(tau);
                            __hip_atomic_store(status, (int64_t)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    } else {
                        int pos = incl - mine;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int g = 0; g < 8; ++g) {
                                unsigned int m = (unsigned int)g2[q][g];
                                while (m) {
                                    const int bit = __builtin_ctz(m);
                                    m &= m - 1u;
                                    clist[pos++] = (int64_t)(((unsigned int)lane + 64u * q) * GRES_T + 32u * g + (unsigned int)bit);
                                }
                            }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        winner = gres_exact_argmin(a, step, prev_row, clist, total, terms, lane, tm_, tc1_, tc2_, tew_,
                                                   blockIdx.x == 0 ? status : nullptr, trip_seen);
                        stat_windows += (unsigned long long)total;
                        stat_rounds += 1;
                    }
                }
                if (lane == 0) { bcast[0] = winner; bcast[1] = state; }
            }
            __syncthreads();
        }
        if (bcast[1] < 0) break;                             // uniform: everybody leaves
        prev_row = g32_uniform_i(bcast[0]);
        if (blockIdx.x == 0 && tid == 0) path[a.out_off[0] + step] = prev_row;
    }
    if (blockIdx.x == 0 && tid == 0 && (stat_rounds | stat_windows)) {
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&status[1]), stat_rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&status[2]), stat_windows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void greedy_res_init_kernel(GresRec *rec, unsigned long long *rec2, int n, int64_t *status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        rec[i].a = 0ull; rec[i].b = 0ull;
        rec[n + i].a = 0ull; rec[n + i].b = 0ull;              // both parities (greedy_res_kernel)
        for (int g = 0; g < 8; ++g) rec2[(size_t)i * 8 + g] = 0ull;
    }
    if (i == 0) for (int w = 0; w < 16; ++w) status[w] = 0;
}

static void *gres_trace_dev = nullptr;

static size_t gres_lds_bytes(const GreedyLayout &g, int Dt)
{
    const int JQ4 = (g.jdim + 3) / 4;
    const int nep = (g.last_frame_as_target && g.me > 1) ? 2 : g.me;
    return (size_t)JQ4 * GRES_T * 16 + (size_t)JQ4 * 32 + 256 + GRES_MAXCAND * 8 + (size_t)(g.jdim + nep * Dt + GRES_MAXCAND) * 8;
}

// one workgroup per compute unit holds 256 windows: the whole windowed database must fit the chip
bool greedy_res_supported(const GreedyLayout &g, int Dt, int n_cus)
{
    const int64_t nb = (g.Nwin + GRES_T - 1) / GRES_T;
    const int cap = n_cus < 256 ? n_cus : 256;
    return nb >= 1 && nb <= cap && (g.jdim + 3) / 4 <= 38 && gres_lds_bytes(g, Dt) <= (size_t)(160 * 1024);
}

// per workgroup: the step's record (16 bytes) for each step parity and the second round's membership mask (8 granules; it
// needs no second copy: a workgroup reaches the second round of a later step only after gathering that step's first-round
// records of everybody, which are published after the masks of the earlier step were read)
size_t greedy_res_record_bytes(const GreedyLayout &g) { return (size_t)((g.Nwin + GRES_T - 1) / GRES_T) * (2 * sizeof(GresRec) + 64); }

// One utterance, all steps, one launch.  *status (device, 4 words): 0 or 1 + the first step that was not decided here,
// rounds, windows settled by exact totals, watchdog.  hst: the hoisted target term of the utterance (required).
void launch_greedy_res(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt, const float *JC_unw, int Jp,
                       int Dj, const double *wj, const float *tiles, const double *Q, int64_t q_off, int64_t nsteps,
                       int64_t out_off, int64_t start, int flags, void *rec, int64_t *status, int64_t *path,
                       const G32Hoist *hst, hipStream_t s)
{
    if (nsteps <= 0) return;
    GreedyArgs a{};
    greedy_fill_args(a, g, F_unw, Fp, Dt, wt, JC_unw, Jp, Dj, wj, Q, true, tiles);
    a.lds_mode = 0;
    a.nu = 1;
    a.hoist = 1; a.Wp = hst->Wp; a.hoist_c = hst->c; a.fwmax2 = hst->fwmax2;
    for (int u = 0; u < 6; ++u) {
        a.W[u] = u == 0 ? hst->W[0] : nullptr; a.qn2[u] = u == 0 ? hst->qn2[0] : nullptr;
        a.q_off[u] = u == 0 ? q_off : 0; a.nsteps_u[u] = u == 0 ? nsteps : 0; a.out_off[u] = u == 0 ? out_off : 0;
        a.start[u] = u == 0 ? start : -1;
    }
    const int nb = (int)((g.Nwin + GRES_T - 1) / GRES_T);
    const int JQ4 = (g.jdim + 3) / 4, tile_q = (g.jdim + GR_CC - 1) / GR_CC * 8;
    const size_t lds = gres_lds_bytes(g, Dt);
    unsigned long long *rec2 = reinterpret_cast<unsigned long long *>(reinterpret_cast<GresRec *>(rec) + 2 * nb);
    hipLaunchKernelGGL(greedy_res_init_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, reinterpret_cast<GresRec *>(rec), rec2, nb, status);
    // (every instance is given the largest LDS size once per device)
    unsigned long long *trace = nullptr;
    if (getenv("SNK_GRES_TRACE")) {
        if (!gres_trace_dev) (void)hipMalloc(&gres_trace_dev, 2 * 256 * 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(gres_trace_dev, 0, 2 * 256 * 8 * sizeof(unsigned long long), s);
        trace = reinterpret_cast<unsigned long long *>(gres_trace_dev);
    }
#define SNK_GRES(JQ_, EX_)                                                                                         \
    {                                                                                                              \
        static size_t attr[32] = {0};                                                                              \
        lds_attr_ensure(attr, (size_t)(160 * 1024), [] {                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(greedy_res_kernel<JQ_, EX_>),                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)); });          \
        hipLaunchKernelGGL((greedy_res_kernel<JQ_, EX_>), dim3(nb), dim3(GRES_T), lds, s, a, nsteps, flags, JQ4, tile_q, \
                           reinterpret_cast<GresRec *>(rec), rec2, path, status, trace);                           \
    }
    // magphase-60 join rows (151 columns: 38 float4) have their own instance; other widths the next larger one
    if (JQ4 == 38) SNK_GRES(38, true)
    else if (JQ4 <= 10) SNK_GRES(10, false) else if (JQ4 <= 20) SNK_GRES(20, false) else if (JQ4 <= 30) SNK_GRES(30, false)
    else SNK_GRES(38, false)
#undef SNK_GRES
}

// developer aid: SNK_GRES_TRACE=<file> writes the last launch's timeline of workgroup 0 (256 steps x 8 stamps) after the caller's sync
void greedy_res_trace_dump()
{
    const char *fn = getenv("SNK_GRES_TRACE");
    if (!fn || !gres_trace_dev) return;
    static unsigned long long host[2 * 256 * 8];
    if (hipMemcpy(host, gres_trace_dev, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return;
    if (FILE *f = fopen(fn, "wb")) { fwrite(host, sizeof(host), 1, f); fclose(f); }
}

}  // namespace snk
