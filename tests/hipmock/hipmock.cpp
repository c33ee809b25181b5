// Test infrastructure: a stand-in for the HIP runtime so that the HOST side of libdesc_amd (planners, CSR builders,
// upload sizes, buffer hand-over to the caller) can run on a machine without a GPU under AddressSanitizer /
// ThreadSanitizer.  "Device" memory is the host heap (so the sanitizers see every copy), copies are memcpy, kernel
// launches do nothing (device-produced values read back as zeros), streams and events are tokens.  Only
// tests/test_host_sanitizers.py links this; the product never does.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {
std::mutex g_mu;
std::unordered_map<void*, size_t> g_blocks;           // live "device" blocks and their sizes
std::atomic<long> g_launches{0};
struct CallConfig { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local CallConfig g_cfg;
}  // namespace

extern "C" {

long hipmock_launch_count() { return g_launches.load(); }
size_t hipmock_live_blocks() { std::lock_guard<std::mutex> l(g_mu); return g_blocks.size(); }

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "mock error"; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
    std::memset(p, 0, sizeof *p);
    std::strcpy(p->name, "hipmock gfx950");
    p->totalGlobalMem = (size_t)16 << 30;
    p->sharedMemPerBlock = 64 << 10;
    p->maxSharedMemoryPerMultiProcessor = 160 << 10;
    p->sharedMemPerBlockOptin = 160 << 10;
    p->warpSize = 64;
    p->maxThreadsPerBlock = 1024;
    p->multiProcessorCount = 256;
    p->l2CacheSize = 4 << 20;
    p->clockRate = 2400000;
    std::strcpy(p->gcnArchName, "gfx950");
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* fr, size_t* tot) { *fr = (size_t)12 << 30; *tot = (size_t)16 << 30; return hipSuccess; }

hipError_t hipMalloc(void** p, size_t bytes) {
    void* q = std::calloc(1, bytes ? bytes : 1);
    if (!q) return hipErrorOutOfMemory;
    { std::lock_guard<std::mutex> l(g_mu); g_blocks[q] = bytes; }
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    { std::lock_guard<std::mutex> l(g_mu); if (!g_blocks.erase(p)) std::abort(); }     // free of a pointer hipMalloc never returned
    std::free(p);
    return hipSuccess;
}
hipError_t hipExtMallocWithFlags(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { *p = std::calloc(1, bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
int hipGetStreamDeviceId(hipStream_t) { return 0; }      // hipCUB's host side asks (device scans of the structure builder)
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 64; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { std::free((void*)s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { std::free((void*)e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }

hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 2; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { ++g_launches; return hipSuccess; }
// stream capture: kernels are no-ops here anyway, so a captured graph is a token
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = (hipGraph_t)std::malloc(8); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = (hipGraphExec_t)std::malloc(8); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { ++g_launches; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { std::free((void*)e); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { std::free((void*)g); return hipSuccess; }

// what the host stubs and the module constructor of a --offload-host-only object call
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) { g_cfg = {grid, block, shmem, stream}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
    *grid = g_cfg.grid; *block = g_cfg.block; *shmem = g_cfg.shmem; *stream = g_cfg.stream; return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipRegisterManagedVar(void*, void**, void*, const char*, size_t, unsigned) {}
void __hipRegisterSurface(void**, void*, char*, char*, int, int) {}
void __hipRegisterTexture(void**, void*, char*, char*, int, int, int) {}
}
