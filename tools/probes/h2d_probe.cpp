// How long does the HOST spend inside hipMemcpyAsync for a 9.4 MB host -> device copy (the query rows of a B* step), by kind of
// host memory (hipHostMalloc / hipHostRegister'ed malloc) and state of the stream (idle / busy with a kernel), and how long
// does the copy take on the device?   hipcc --offload-arch=gfx950 -O2 tools/probes/h2d_probe.cpp -o /tmp/h2d_probe && /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long cycles, int *out) { const long long t0 = clock64(); while (clock64() - t0 < cycles) {} if (out) *out = 1; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = (size_t)19200 * 61 * 8;
    void *dev = nullptr, *pinned = nullptr;
    CK(hipMalloc(&dev, bytes));
    CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
    void *reg = aligned_alloc(4096, (bytes + 4095) & ~(size_t)4095);
    memset(reg, 1, bytes); memset(pinned, 1, bytes);
    CK(hipHostRegister(reg, bytes, hipHostRegisterDefault));
    hipStream_t s_main, s_up;
    CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
    hipEvent_t e0, e1, ex;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ex, hipEventDisableTiming));
    for (int kind = 0; kind < 2; ++kind) {
        const void *src = kind ? reg : pinned;
        for (int busy = 0; busy < 3; ++busy) {
            // busy 0: the copy's stream is idle; 1: a 2 ms kernel runs on the copy's stream; 2: the kernel runs on ANOTHER stream and the
            // copy's stream is idle (what the upload stream of snk_knn_viterbi_batch_submit sees)
            double host = 0, devms = 0;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipDeviceSynchronize());
                if (busy == 1) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s_up, 4000000LL, nullptr);
                if (busy == 2) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s_main, 4000000LL, nullptr);
                CK(hipEventRecord(e0, s_up));
                const double t0 = now();
                CK(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, s_up));
                const double t1 = now();
                CK(hipEventRecord(e1, s_up));
                CK(hipEventRecord(ex, s_up));
                CK(hipStreamWaitEvent(s_main, ex, 0));
                CK(hipDeviceSynchronize());
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) { host += t1 - t0; devms += ms; }
            }
            printf("%-16s stream %-28s host inside hipMemcpyAsync %.3f ms, events around the copy %.3f ms\n", kind ? "hipHostRegister" : "hipHostMalloc",
                   busy == 0 ? "idle" : busy == 1 ? "busy (kernel in front)" : "idle, another stream busy", host / 5, devms / 5);
        }
    }
    return 0;
}
