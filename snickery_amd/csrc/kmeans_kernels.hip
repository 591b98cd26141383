// An ORDER for the units of a voice whose database order says nothing about closeness (no reference counterpart in form; in
// purpose it is what the reference's KD-trees are for -- script/synth_halfphone.py:379,393-401, synth_simple.py:229: a spatial
// index over the weighted unit vectors, rebuilt per set of weights).
//
// The K-NN filter works on tiles of 32 consecutive units and prunes a tile by its BALL (centre, radius: knn16_kernels.hip pass 0).
// In a frame-level voice consecutive units are consecutive frames and the balls are tiny.  Where they are not -- halfphone voices,
// whose consecutive units are different phone halves; any voice whose utterances were shuffled -- nearly every ball reaches every
// query, the filter falls back to sweeping the whole database (one-pass three-term sweep: 4.3 ms per 9 600 rows at B* against 0.48 ms
// with compact tiles) and nothing is pruned.  So the engine may PERMUTE the prefilter's operands: units are clustered (Lloyd's
// k-means on the weighted vectors, a few hundred units per cluster -- what fits a compute unit's LDS --, a few iterations, float32:
// a heuristic, nothing of it reaches a result), every cluster is put in the order of a nearest-neighbour CHAIN (from the member
// farthest from the centroid, always on to the nearest member not yet taken: along a stretch of speech that is its time order,
// which is what made the tiles of a frame-level voice compact in the first place) and the clusters are laid out one after the
// other; a tile of 32 consecutive positions then holds neighbours.  (Clusters alone were tried first: a tile drawn at random from
// a cluster of 256 has the cluster's radius, and the ball pass listed a quarter of all pairs again.)  Only the prefilter sees the
// permutation: operand builders read row perm[position], and the bucket kernel maps a
// survivor's position back to its unit id before anything is ranked -- the exact float64 re-rank, its (distance, id) order and
// every result are those of the database order.
//
//   km_init      centroid c = unit floor(c N / C)
//   km_assign    nearest centroid of every unit (workgroup: 128 units staged in LDS as float32, centroids in chunks of 32;
//                a thread keeps eight centroids' dot products in registers per pass over its unit's columns) + cluster sizes
//   km_scan      exclusive prefix of the sizes (one workgroup)
//   km_scatter   perm[start[c] + k] = the k-th unit of cluster c (order inside a cluster: as the atomics fall -- any order is valid)
//   km_means     centroid = mean of its members (one workgroup per cluster, members through perm); empty clusters keep theirs
//   km_chain     one workgroup per cluster: its members' rows in LDS (float32, column-major), n steps of "nearest member not yet
//                taken" (a workgroup-wide argmin per step); clusters larger than the LDS holds are chained piece by piece
#include "snk_internal.h"
#include <float.h>

namespace snk {

typedef float km_f4 __attribute__((ext_vector_type(4)));
#define KM_UB 128              // units per workgroup of km_assign
#define KM_CJ 32               // centroids per LDS chunk

__global__ void __launch_bounds__(256)
km_init_kernel(const double *__restrict__ Fw, int64_t N, int Dt, int Dpad, int C, int Dp, float *__restrict__ cen, float *__restrict__ cnorm)
{
    const int c = blockIdx.x;
    const int64_t row = (int64_t)(((double)c + 0.5) * (double)N / (double)C);
    __shared__ float red[256];
    float n2 = 0.f;
    for (int d = threadIdx.x; d < Dp; d += 256) {
        const float v = d < Dt ? (float)Fw[(row < N ? row : N - 1) * Dpad + d] : 0.f;
        cen[(int64_t)c * Dp + d] = v;
        n2 += v * v;
    }
    red[threadIdx.x] = n2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) cnorm[c] = red[0];
}

// Dp: columns padded to a multiple of 4 (float4 reads of the centroid chunk)
__global__ void __launch_bounds__(KM_UB)
km_assign_kernel(const double *__restrict__ Fw, int64_t N, int Dt, int Dpad, int C, int Dp, const float *__restrict__ cen,
                 const float *__restrict__ cnorm, int *__restrict__ assign, int *__restrict__ count)
{
    extern __shared__ __align__(16) float km_lds[];
    float *xs = km_lds;                              // [Dp][KM_UB]: column-major, a thread reads its own unit's column
    float *cs = xs + (size_t)Dp * KM_UB;             // [Dp][KM_CJ]: the chunk's centroids, eight of one column side by side
    const int tid = threadIdx.x;
    const int64_t u0 = (int64_t)blockIdx.x * KM_UB;
    // stage the units (float32).  Row by row: the threads of a row's wavefront read consecutive columns
    for (int r = 0; r < KM_UB; ++r) {
        const int64_t row = u0 + r;
        for (int d = tid; d < Dp; d += KM_UB) xs[(size_t)d * KM_UB + r] = (row < N && d < Dt) ? (float)Fw[row * Dpad + d] : 0.f;
    }
    float best = FLT_MAX;
    int arg = 0;
    for (int c0 = 0; c0 < C; c0 += KM_CJ) {
        __syncthreads();                             // (the units are staged; the previous chunk is done with)
        for (int i = tid; i < KM_CJ * Dp; i += KM_UB) {
            const int j = i / Dp, d = i - j * Dp;
            cs[(size_t)d * KM_CJ + j] = (c0 + j < C) ? cen[(int64_t)(c0 + j) * Dp + d] : 0.f;
        }
        __syncthreads();
#pragma unroll 1
        for (int jb = 0; jb < KM_CJ; jb += 8) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
            for (int d = 0; d < Dp; ++d) {
                const float x = xs[(size_t)d * KM_UB + tid];
                const km_f4 ca = *reinterpret_cast<const km_f4 *>(cs + (size_t)d * KM_CJ + jb);
                const km_f4 cb = *reinterpret_cast<const km_f4 *>(cs + (size_t)d * KM_CJ + jb + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[j] = __builtin_fmaf(x, ca[j], acc[j]); acc[4 + j] = __builtin_fmaf(x, cb[j], acc[4 + j]); }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + jb + j;
                const float score = c < C ? cnorm[c] - 2.f * acc[j] : FLT_MAX;
                if (score < best) { best = score; arg = c; }
            }
        }
    }
    if (u0 + tid < N) {
        assign[u0 + tid] = arg;
        atomicAdd(&count[arg], 1);
    }
}

__global__ void __launch_bounds__(1024)
km_scan_kernel(const int *__restrict__ count, int C, int *__restrict__ start, int *__restrict__ cursor)
{
    __shared__ int part[1024];
    const int per = (C + 1023) / 1024;
    int sum = 0;
    for (int i = 0; i < per; ++i) { const int c = threadIdx.x * per + i; if (c < C) sum += count[c]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int i = 0; i < per; ++i) {
        const int c = threadIdx.x * per + i;
        if (c < C) { start[c] = run; cursor[c] = 0; run += count[c]; }
    }
    if (threadIdx.x == 1023) start[C] = part[1023];
}

__global__ void __launch_bounds__(256)
km_scatter_kernel(const int *__restrict__ assign, int64_t N, const int *__restrict__ start, int *__restrict__ cursor, int *__restrict__ perm)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int c = assign[i];
    perm[start[c] + atomicAdd(&cursor[c], 1)] = (int)i;
}

__global__ void __launch_bounds__(256)
km_means_kernel(const double *__restrict__ Fw, int Dt, int Dpad, int Dp, const int *__restrict__ start, const int *__restrict__ perm,
                float *__restrict__ cen, float *__restrict__ cnorm, int *__restrict__ count)
{
    const int c = blockIdx.x;
    const int s0 = start[c], n = start[c + 1] - s0;
    if (threadIdx.x == 0) count[c] = 0;              // (for the next iteration's sizes)
    if (n == 0) return;                              // an empty cluster keeps its centroid
    __shared__ float red[256];
    // thread = column (Dp <= 256); members one after the other: a wavefront reads consecutive columns of one row
    float n2 = 0.f;
    const int d = threadIdx.x;
    if (d < Dp) {
        double sum = 0.0;
        if (d < Dt)
            for (int k = 0; k < n; ++k) sum += Fw[(int64_t)perm[s0 + k] * Dpad + d];
        const float v = (float)(sum / (double)n);
        cen[(int64_t)c * Dp + d] = v;
        n2 = v * v;
    }
    red[threadIdx.x] = n2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) cnorm[c] = red[0];
}

// members of a cluster whose rows fit the chain kernel's LDS at once (120 KB of float32 columns), a multiple of 32, at most 480
int kmeans_chain_capacity(int Dt)
{
    const int Dp = (Dt + 3) & ~3;
    int cap = (120 * 1024) / (Dp * 4);
    cap = cap / 32 * 32;
    return cap > 480 ? 480 : cap < 32 ? 32 : cap;
}

int kmeans_clusters(int64_t N, int Dt)
{
    int64_t c = (N / kmeans_chain_capacity(Dt) + 63) / 64 * 64;
    if (c < 64) c = 64;
    if (c > 16384) c = 16384;
    return (int)c;
}

// The order of the CLUSTERS: a nearest-neighbour chain over the centroids (one workgroup; C steps of a workgroup-wide argmin), so
// that a cluster's neighbours in the layout are its neighbours in space and the tile that straddles two clusters holds units of
// adjoining regions (laid out in any order, one tile in fourteen straddled two unrelated clusters and had their distance as its
// radius).  prev[c] = the cluster in front of c (-1: the first), ostart[c] = first position of c's segment in that order.
__global__ void __launch_bounds__(256)
km_order_kernel(const float *__restrict__ cen, const int *__restrict__ count, int C, int Dp, int *__restrict__ prev, int *__restrict__ ostart,
                unsigned char *__restrict__ taken)
{
    __shared__ float red_v[4];
    __shared__ int red_i[4], cur_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) taken[c] = 0;
    if (tid == 0) cur_s = 0;
    __syncthreads();
    int run = 0, last = -1;
    for (int step = 0; step < C; ++step) {
        const int cur = cur_s;
        if (tid == 0) { taken[cur] = 1; prev[cur] = last; ostart[cur] = run; }
        run += count[cur];
        last = cur;
        __syncthreads();
        if (step + 1 == C) break;
        float v = FLT_MAX;
        int ii = 0x7fffffff;
        for (int c = tid; c < C; c += 256) {
            if (taken[c]) continue;
            float acc = 0.f;
            for (int d = 0; d < Dp; ++d) { const float df = cen[(int64_t)c * Dp + d] - cen[(int64_t)cur * Dp + d]; acc = __builtin_fmaf(df, df, acc); }
            // non-finite rows (NaN / inf features): a distance that compares as not-less must still leave a FREE cluster in ii, or the
            // chain writes taken[0x7fffffff] and the order is no permutation; lowest index among equals, as in the reduction
            acc = acc < FLT_MAX ? acc : FLT_MAX;
            if (acc < v || (acc == v && c < ii)) { v = acc; ii = c; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(v, off, 64); const int oi = __shfl_xor(ii, off, 64);
            if (ov < v || (ov == v && oi < ii)) { v = ov; ii = oi; }
        }
        if (lane == 0) { red_v[wave] = v; red_i[wave] = ii; }
        __syncthreads();
        if (tid == 0) {
            float vv = red_v[0]; int jj = red_i[0];
            for (int w = 1; w < 4; ++w) if (red_v[w] < vv || (red_v[w] == vv && red_i[w] < jj)) { vv = red_v[w]; jj = red_i[w]; }
            cur_s = jj;
        }
        __syncthreads();
    }
}

// The chain.  xs[d][i]: column d of member i (i < m <= cap); a step = the squared distance of every free member to the current one
// (a thread takes members tid, tid + 256, ...), the workgroup's argmin, the winner's id to the output.
__global__ void __launch_bounds__(256)
km_chain_kernel(const double *__restrict__ Fw, int Dt, int Dpad, int Dp, int cap, const int *__restrict__ start, const float *__restrict__ cen,
                const int *__restrict__ perm_in, int *__restrict__ perm_out, const int *__restrict__ prev, const int *__restrict__ ostart)
{
    extern __shared__ __align__(16) float kc_lds[];
    float *xs = kc_lds;                               // [Dp][cap]
    int *ids = reinterpret_cast<int *>(xs + (size_t)Dp * cap);      // [cap] unit of member i; -1 once taken
    __shared__ float red_v[4];
    __shared__ int red_i[4], cur_s;
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s0 = start[c], n = start[c + 1] - s0, o0 = ostart[c];
    // the chain starts at the member nearest to the centroid of the cluster in front of this one (the layout runs on from there);
    // the first cluster's at the member farthest from its own centroid (an end of the stretch, not its middle)
    const int pc = prev[c];
    const float *const ref = cen + (int64_t)(pc >= 0 ? pc : c) * Dp;
    const float sign = pc >= 0 ? -1.f : 1.f;
    for (int p0 = 0; p0 < n; p0 += cap) {
        const int m = n - p0 < cap ? n - p0 : cap;
        __syncthreads();
        for (int i = tid; i < m; i += 256) ids[i] = perm_in[s0 + p0 + i];
        __syncthreads();
        // rows in: a wavefront reads consecutive columns of one member
        for (int i = wave; i < m; i += 4) {
            const int64_t u = ids[i];
            for (int d = lane; d < Dp; d += 64) xs[(size_t)d * cap + i] = d < Dt ? (float)Fw[u * Dpad + d] : 0.f;
        }
        __syncthreads();
        float bv = -FLT_MAX;
        int bi = 0;
        for (int i = tid; i < m; i += 256) {
            float acc = 0.f;
            for (int d = 0; d < Dp; ++d) { const float df = xs[(size_t)d * cap + i] - ref[d]; acc = __builtin_fmaf(df, df, acc); }
            acc *= sign;                              // nearest to the cluster in front = largest negated distance
            if (acc > bv) { bv = acc; bi = i; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64); const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = red_v[0]; int ii = red_i[0];
            for (int w = 1; w < 4; ++w) if (red_v[w] > v || (red_v[w] == v && red_i[w] < ii)) { v = red_v[w]; ii = red_i[w]; }
            cur_s = ii;
        }
        __syncthreads();
        for (int step = 0; step < m; ++step) {
            const int cur = cur_s;
            if (tid == 0) { perm_out[o0 + p0 + step] = ids[cur]; ids[cur] = -1; }
            __syncthreads();
            if (step + 1 == m) break;
            // nearest free member (members tid and tid + 256: cap <= 480)
            const int i0 = tid, i1 = tid + 256;
            const bool f0 = i0 < m && ids[i0] >= 0, f1 = i1 < m && ids[i1] >= 0;
            float a0 = 0.f, a1 = 0.f;
            for (int d = 0; d < Dp; ++d) {
                const float cv = xs[(size_t)d * cap + cur];
                const float d0 = xs[(size_t)d * cap + (f0 ? i0 : cur)] - cv, d1 = xs[(size_t)d * cap + (f1 ? i1 : cur)] - cv;
                a0 = __builtin_fmaf(d0, d0, a0); a1 = __builtin_fmaf(d1, d1, a1);
            }
            // (non-finite distances count as FLT_MAX; a lane without a free member holds no index at all: ADVICE r5)
            a0 = a0 < FLT_MAX ? a0 : FLT_MAX; a1 = a1 < FLT_MAX ? a1 : FLT_MAX;
            float v = FLT_MAX;
            int ii = 0x7fffffff;
            if (f0) { v = a0; ii = i0; }
            if (f1 && (a1 < v || ii == 0x7fffffff)) { v = a1; ii = i1; }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ov = __shfl_xor(v, off, 64); const int oi = __shfl_xor(ii, off, 64);
                if (ov < v || (ov == v && oi < ii)) { v = ov; ii = oi; }
            }
            if (lane == 0) { red_v[wave] = v; red_i[wave] = ii; }
            __syncthreads();
            if (tid == 0) {
                float vv = red_v[0]; int jj = red_i[0];
                for (int w = 1; w < 4; ++w) if (red_v[w] < vv || (red_v[w] == vv && red_i[w] < jj)) { vv = red_v[w]; jj = red_i[w]; }
                cur_s = jj;
            }
            __syncthreads();
        }
    }
}

// workspace: centroids C x Dp floats | norms C floats | assign N ints | count C + 2 ints | start C + 2 ints | cursor C ints | the
// clusters' members before the chain N ints | prev, ostart C + 2 ints each | taken C bytes
size_t kmeans_workspace_bytes(int64_t N, int Dt)
{
    const int C = kmeans_clusters(N, Dt), Dp = (Dt + 3) & ~3;
    return (size_t)C * Dp * 4 + (size_t)C * 4 + (size_t)N * 4 + (size_t)(3 * C + 8) * 4 + (size_t)N * 4 + (size_t)(3 * C + 8) * 4 + 256;
}

// km_assign keeps Dp x (KM_UB + KM_CJ) floats in LDS: widths whose tile would reach the 160 KB of a compute unit are not clustered
// (Dt 253 .. 256 asked for exactly 163 840 bytes)
bool kmeans_supported(int Dt) { return Dt >= 1 && (size_t)((Dt + 3) & ~3) * (KM_UB + KM_CJ) * sizeof(float) <= (size_t)150 * 1024; }

// Is `perm` a permutation of 0 .. N-1?  seen: N zeroed words of scratch; *bad counts the entries out of range or seen before.
__global__ void __launch_bounds__(256)
km_perm_check_kernel(const int *__restrict__ perm, int64_t N, unsigned int *__restrict__ seen, unsigned int *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int p = perm[i];
    if (p < 0 || p >= N || atomicAdd(&seen[p], 1u) != 0u) atomicAdd(bad, 1u);
}

size_t kmeans_perm_check_bytes(int64_t N) { return (size_t)(N + 1) * sizeof(unsigned int); }

void launch_perm_check(const int *perm, int64_t N, void *scratch, hipStream_t s)
{
    (void)hipMemsetAsync(scratch, 0, kmeans_perm_check_bytes(N), s);
    unsigned int *seen = static_cast<unsigned int *>(scratch);
    hipLaunchKernelGGL(km_perm_check_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, perm, N, seen + 1, seen);      // word 0: the verdict
}

void launch_kmeans_order(const double *Fw, int64_t N, int Dt, int Dpad, int iters, void *workspace, int *perm, hipStream_t s)
{
    const int C = kmeans_clusters(N, Dt), Dp = (Dt + 3) & ~3;
    char *w = static_cast<char *>(workspace);
    float *cen = reinterpret_cast<float *>(w); w += (size_t)C * Dp * 4;
    float *cnorm = reinterpret_cast<float *>(w); w += (size_t)C * 4;
    int *assign = reinterpret_cast<int *>(w); w += (size_t)N * 4;
    int *count = reinterpret_cast<int *>(w); w += (size_t)(C + 2) * 4;
    int *start = reinterpret_cast<int *>(w); w += (size_t)(C + 2) * 4;
    int *cursor = reinterpret_cast<int *>(w); w += (size_t)(C + 2) * 4;
    int *members = reinterpret_cast<int *>(w); w += (size_t)N * 4;
    int *prev = reinterpret_cast<int *>(w); w += (size_t)(C + 2) * 4;
    int *ostart = reinterpret_cast<int *>(w); w += (size_t)(C + 2) * 4;
    unsigned char *taken = reinterpret_cast<unsigned char *>(w);
    const size_t lds = ((size_t)Dp * KM_UB + (size_t)Dp * KM_CJ) * sizeof(float);
    static size_t attr[32] = {0};
    if (lds > 65536)
        lds_attr_ensure(attr, lds, [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&km_assign_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL(km_init_kernel, dim3(C), dim3(256), 0, s, Fw, N, Dt, Dpad, C, Dp, cen, cnorm);
    (void)hipMemsetAsync(count, 0, (size_t)(C + 2) * 4, s);
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(km_assign_kernel, dim3((unsigned)((N + KM_UB - 1) / KM_UB)), dim3(KM_UB), lds, s, Fw, N, Dt, Dpad, C, Dp, cen, cnorm,
                           assign, count);
        hipLaunchKernelGGL(km_scan_kernel, dim3(1), dim3(1024), 0, s, count, C, start, cursor);
        hipLaunchKernelGGL(km_scatter_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, assign, N, start, cursor, members);
        if (it + 1 < iters)
            hipLaunchKernelGGL(km_means_kernel, dim3(C), dim3(256), 0, s, Fw, Dt, Dpad, Dp, start, members, cen, cnorm, count);
    }
    // every cluster in the order of a nearest-neighbour chain
    const int cap = kmeans_chain_capacity(Dt);
    const size_t lds_c = (size_t)Dp * cap * sizeof(float) + (size_t)cap * sizeof(int);
    static size_t attr_c[32] = {0};
    if (lds_c > 65536)
        lds_attr_ensure(attr_c, lds_c, [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&km_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c); });
    hipLaunchKernelGGL(km_order_kernel, dim3(1), dim3(256), 0, s, cen, count, C, Dp, prev, ostart, taken);
    hipLaunchKernelGGL(km_chain_kernel, dim3(C), dim3(256), lds_c, s, Fw, Dt, Dpad, Dp, cap, start, cen, members, perm, prev, ostart);
}

}  // namespace snk
