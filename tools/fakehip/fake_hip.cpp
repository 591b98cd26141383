// A stand-in for libamdhip64 with NO device behind it: allocation bookkeeping only.  `make asan-host` links the library's
// own translation units, compiled for the host with -fsanitize=address,undefined, against this file, so that the 3 400 lines
// of host orchestration in api_*.hip (argument checks, utterance grouping, the two-batches-in-flight state machine, the
// staging ring, shard plans, the fail() paths in front of a collective) run in the CPU container where sanitizers exist
// (tests/test_host_asan.py).  "Device" memory is zero-filled host memory with the sanitizer's red zones around it, copies
// are memcpy, kernels are never run (a launch is a no-op: every device result reads as zero), streams and events complete
// at once.  Test infrastructure: nothing under snickery_amd/ refers to it.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>

static std::mutex g_m;
static std::map<void *, size_t> g_dev, g_host, g_reg;
static int g_device = 0;
static long g_launches = 0;
static dim3 g_grid, g_block;
static size_t g_shmem = 0;
static hipStream_t g_stream = nullptr;

extern "C" {

// ---- what the host side of a kernel<<<>>> call is compiled into ----
void **__hipRegisterFatBinary(const void *) { static void *h = nullptr; return &h; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned int, void *, void *, dim3 *, dim3 *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t s)
{
    g_grid = grid; g_block = block; g_shmem = shmem; g_stream = s;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *s)
{
    *grid = g_grid; *block = g_block; *shmem = g_shmem; *s = g_stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *, dim3 grid, dim3 block, void **, size_t, hipStream_t)
{
    // a launch with an empty grid or block is an error on the real runtime too
    if (!grid.x || !grid.y || !grid.z || !block.x || !block.y || !block.z) return hipErrorInvalidConfiguration;
    ++g_launches;
    return hipSuccess;
}
long fakehip_launches() { return g_launches; }

}  // extern "C"

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d != 0) return hipErrorInvalidDevice; g_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = g_device; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t *p, int d)
{
    if (d != 0) return hipErrorInvalidDevice;
    memset(p, 0, sizeof(*p));
    strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
    strcpy(p->name, "fake MI355X (no device: tools/fakehip)");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)288 << 30;
    p->warpSize = 64;
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake-hip error"; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)200 << 30; *t = (size_t)288 << 30; return hipSuccess; }

hipError_t hipMalloc(void **p, size_t n)
{
    // "device" memory: a plain host block (the sanitizer's red zones sit around it); sizes beyond what the CPU box has fail
    // like an out-of-memory device
    if (n > ((size_t)6 << 30)) { *p = nullptr; return hipErrorOutOfMemory; }
    *p = calloc(n ? n : 1, 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> l(g_m);
    g_dev[*p] = n;
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    { std::lock_guard<std::mutex> l(g_m); if (!g_dev.erase(p)) return hipErrorInvalidValue; }
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned int)
{
    *p = calloc(n ? n : 1, 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> l(g_m);
    g_host[*p] = n;
    return hipSuccess;
}
hipError_t hipHostFree(void *p)
{
    if (!p) return hipSuccess;
    { std::lock_guard<std::mutex> l(g_m); if (!g_host.erase(p)) return hipErrorInvalidValue; }
    free(p);
    return hipSuccess;
}
hipError_t hipHostRegister(void *p, size_t n, unsigned int)
{
    std::lock_guard<std::mutex> l(g_m);
    if (g_reg.count(p)) return hipErrorHostMemoryAlreadyRegistered;
    g_reg[p] = n;
    return hipSuccess;
}
hipError_t hipHostUnregister(void *p)
{
    std::lock_guard<std::mutex> l(g_m);
    return g_reg.erase(p) ? hipSuccess : hipErrorHostMemoryNotRegistered;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p)
{
    memset(a, 0, sizeof(*a));
    std::lock_guard<std::mutex> l(g_m);
    for (auto *m : {&g_host, &g_reg})
        for (auto &kv : *m)
            if ((const char *)p >= (const char *)kv.first && (const char *)p < (const char *)kv.first + kv.second) {
                a->type = hipMemoryTypeHost;
                a->hostPointer = const_cast<void *>(p);
                return hipSuccess;
            }
    return hipErrorInvalidValue;                 // pageable memory: the real runtime says so too
}

hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }

static void *token() { return malloc(8); }      // a distinct, leak-checked handle per stream / event
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int) { *s = (hipStream_t)token(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned int) { return e ? hipSuccess : hipErrorInvalidHandle; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)token(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned int) { *e = (hipEvent_t)token(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { return e ? hipSuccess : hipErrorInvalidHandle; }
hipError_t hipEventSynchronize(hipEvent_t e) { return e ? hipSuccess : hipErrorInvalidHandle; }
hipError_t hipEventQuery(hipEvent_t e) { return e ? hipSuccess : hipErrorInvalidHandle; }      // (nothing ever runs here: everything queued is done)
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = 0.f; return (a && b) ? hipSuccess : hipErrorInvalidHandle; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

// the guard allocator of SNK_GUARD=1|2 (virtual memory management): not offered here
hipError_t hipMemGetAllocationGranularity(size_t *, const hipMemAllocationProp *, hipMemAllocationGranularity_flags) { return hipErrorNotSupported; }
hipError_t hipMemAddressReserve(void **, size_t, size_t, void *, unsigned long long) { return hipErrorNotSupported; }
hipError_t hipMemAddressFree(void *, size_t) { return hipErrorNotSupported; }
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *, size_t, const hipMemAllocationProp *, unsigned long long) { return hipErrorNotSupported; }
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t) { return hipErrorNotSupported; }
hipError_t hipMemMap(void *, size_t, size_t, hipMemGenericAllocationHandle_t, unsigned long long) { return hipErrorNotSupported; }
hipError_t hipMemUnmap(void *, size_t) { return hipErrorNotSupported; }
hipError_t hipMemSetAccess(void *, size_t, const hipMemAccessDesc *, size_t) { return hipErrorNotSupported; }
