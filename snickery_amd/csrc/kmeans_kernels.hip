// An ORDER for the units of a voice whose database order says nothing about closeness (no reference counterpart in form; in
// purpose it is what the reference's KD-trees are for -- script/synth_halfphone.py:379,393-401, synth_simple.py:229: a spatial
// index over the weighted unit vectors, rebuilt per set of weights).
//
// The K-NN filter works on tiles of 32 consecutive units and prunes a tile by its BALL (centre, radius: knn16_kernels.hip pass 0).
// In a frame-level voice consecutive units are consecutive frames and the balls are tiny.  Where they are not -- halfphone voices,
// whose consecutive units are different phone halves; any voice whose utterances were shuffled -- nearly every ball reaches every
// query, the filter falls back to sweeping the whole database (one-pass three-term sweep: 4.3 ms per 9 600 rows at B* against 0.48 ms
// with compact tiles) and nothing is pruned.  So the engine may PERMUTE the prefilter's operands: units are clustered (Lloyd's
// k-means on the weighted vectors, about 256 units per cluster, a few iterations, float32 -- it is a heuristic, nothing of it
// reaches a result) and laid out cluster by cluster; tiles then hold units of one cluster and their balls are as small as the
// data allows.  Only the prefilter sees the permutation: operand builders read row perm[position], and the bucket kernel maps a
// survivor's position back to its unit id before anything is ranked -- the exact float64 re-rank, its (distance, id) order and
// every result are those of the database order.
//
//   km_init      centroid c = unit floor(c N / C)
//   km_assign    nearest centroid of every unit (workgroup: 128 units staged in LDS as float32, centroids in chunks of 32;
//                a thread keeps eight centroids' dot products in registers per pass over its unit's columns) + cluster sizes
//   km_scan      exclusive prefix of the sizes (one workgroup)
//   km_scatter   perm[start[c] + k] = the k-th unit of cluster c (order inside a cluster: as the atomics fall -- any order is valid)
//   km_means     centroid = mean of its members (one workgroup per cluster, members through perm); empty clusters keep theirs
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

int kmeans_clusters(int64_t N)
{
    int64_t c = (N / 256 + 63) / 64 * 64;
    if (c < 64) c = 64;
    if (c > 8192) c = 8192;
    return (int)c;
}

// workspace: centroids C x Dp floats | norms C floats | assign N ints | count C + 1 ints | start C + 1 ints | cursor C ints
size_t kmeans_workspace_bytes(int64_t N, int Dt)
{
    const int C = kmeans_clusters(N), Dp = (Dt + 3) & ~3;
    return (size_t)C * Dp * 4 + (size_t)C * 4 + (size_t)N * 4 + (size_t)(3 * C + 8) * 4 + 256;
}

bool kmeans_supported(int Dt) { return Dt >= 1 && Dt <= 256; }

void launch_kmeans_order(const double *Fw, int64_t N, int Dt, int Dpad, int iters, void *workspace, int *perm, hipStream_t s)
{
    const int C = kmeans_clusters(N), Dp = (Dt + 3) & ~3;
    char *w = static_cast<char *>(workspace);
    float *cen = reinterpret_cast<float *>(w); w += (size_t)C * Dp * 4;
    float *cnorm = reinterpret_cast<float *>(w); w += (size_t)C * 4;
    int *assign = reinterpret_cast<int *>(w); w += (size_t)N * 4;
    int *count = reinterpret_cast<int *>(w); w += (size_t)(C + 2) * 4;
    int *start = reinterpret_cast<int *>(w); w += (size_t)(C + 2) * 4;
    int *cursor = reinterpret_cast<int *>(w);
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
        hipLaunchKernelGGL(km_scatter_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, assign, N, start, cursor, perm);
        if (it + 1 < iters)
            hipLaunchKernelGGL(km_means_kernel, dim3(C), dim3(256), 0, s, Fw, Dt, Dpad, Dp, start, perm, cen, cnorm, count);
    }
}

}  // namespace snk
