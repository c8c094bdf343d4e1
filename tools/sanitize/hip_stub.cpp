// A host-only stand-in for the HIP runtime entry points libpptoas_hip.so uses -- memory is malloc'ed, copies are
// memcpy's, kernels are NOT run, streams and events do nothing -- so that the library's HOST side (the worker thread
// of pp_fit_submit, the three-deep queue of pp_fit_enqueue / pp_fit_collect, staging blocks, deferred tails, the event
// pool) can run under ThreadSanitizer on a machine without a GPU:  LD_PRELOAD=libhip_stub.so ./tsan_driver
// (tools/sanitize/Makefile).  Not part of the product; nothing here computes a fit.
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <cstdlib>
#include <cstring>

static std::atomic<long> g_launches{0}, g_allocs{0};
static thread_local struct { dim3 g, b; size_t sh; hipStream_t s; } t_cfg;

extern "C" {
long hip_stub_launches() { return g_launches.load(); }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p);
    strcpy(p->name, "hip_stub");
    p->multiProcessorCount = 256; p->sharedMemPerBlock = 160 * 1024; p->totalGlobalMem = (size_t)288 << 30;
    p->warpSize = 64; p->maxThreadsPerBlock = 1024;
    strcpy(p->gcnArchName, "gfx950");
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) { ++g_allocs; *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)200 << 30; *t = (size_t)288 << 30; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < h; ++r) memmove((char*)d + r * dp, (const char*)s + r * sp, w);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static hipError_t new_handle(void** h) { *h = malloc(8); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return new_handle((void**)s); }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return new_handle((void**)s); }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { return new_handle((void**)s); }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { return new_handle((void**)e); }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return new_handle((void**)e); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.01f; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 2; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hip_stub"; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { ++g_launches; return hipSuccess; }
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t s) { t_cfg.g = g; t_cfg.b = b; t_cfg.sh = sh; t_cfg.s = s; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* s) { *g = t_cfg.g; *b = t_cfg.b; *sh = t_cfg.sh; *s = t_cfg.s; return hipSuccess; }
}
