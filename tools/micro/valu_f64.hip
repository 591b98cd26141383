// Micro-benchmark: issue rate of float64 vector instructions on gfx950, per wavefront, with 1, 2 and 4
// wavefronts per SIMD.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_f64.hip -o valu_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(1024) k(double *out, const float *in, int iters)
{
    double a[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0 + threadIdx.x * 1e-9 + i; f[i] = in[(threadIdx.x + i) & 63]; }
    const double c = 1.0000001, d = 0.999999;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = __dadd_rn(a[i], c);
                if (OP == 1) a[i] = __dmul_rn(a[i], c);
                if (OP == 2) a[i] = __fma_rn(a[i], c, d);
                if (OP == 3) { a[i] = __dadd_rn(a[i], (double)f[i]); f[i] += 1.0f; }     // cvt + add (+ f32 add)
                if (OP == 4) { const double v = __dmul_rn((double)f[i], c); const double e = __dsub_rn(v, d); a[i] = __dadd_rn(a[i], __dmul_rn(e, e)); }
            }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}

template <int OP>
void run(const char *name, int ops_per_iter, double *out, float *in)
{
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd;        // one workgroup per CU
        const int iters = 4000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, in, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, in, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)iters * ops_per_iter;                  // per wavefront
        const double wave_instr_per_s = instr * 256 * 4 * waves_per_simd / (ms * 1e-3);
        printf("%-28s %d wave/SIMD: %.3f ms  %.2f Twave-instr/s  = %.1f ns per instr per SIMD  (x64 lanes: %.1f Tlane-op/s)\n",
               name, waves_per_simd, ms, wave_instr_per_s / 1e12, 1e9 * (ms * 1e-3) / (instr * waves_per_simd),
               wave_instr_per_s * 64 / 1e12);
    }
}

int main()
{
    double *out; float *in;
    hipMalloc(&out, 256 * 1024 * sizeof(double));
    hipMalloc(&in, 64 * sizeof(float));
    hipMemset(in, 0, 64 * sizeof(float));
    run<0>("v_add_f64", 32, out, in);
    run<1>("v_mul_f64", 32, out, in);
    run<2>("v_fma_f64", 32, out, in);
    run<3>("cvt_f64_f32 + add (+f32 add)", 96, out, in);
    run<4>("greedy column (cvt,mul,sub,mul,add)", 160, out, in);
    return 0;
}
