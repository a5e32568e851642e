// sanitize/mock_hip/hip/hip_runtime.h -- SANITIZER HARNESS ONLY (make -C primalcr_amd/csrc tsan).
// Just enough of the HIP surface for pcr_p2p.h's HOST side (the shared-memory rendezvous, the generation barrier, the error
// flag, the closing rendezvous) to compile with g++ -fsanitize=thread and run without a GPU: allocations are host memory,
// "IPC handles" carry the pointer (the harness runs the ranks as THREADS of one process), kernels are never launched.
// Device code in the header is parsed but never called.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
typedef void* hipStream_t;
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct hipIpcMemHandle_t { char reserved[64]; };
static const unsigned hipDeviceMallocFinegrained = 1, hipIpcMemLazyEnablePeerAccess = 1;
enum { hipMemcpyDeviceToDevice = 3, hipDeviceAttributeWallClockRate = 1 };
static thread_local dim3 blockIdx, gridDim, threadIdx;

inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
inline hipError_t hipHostMalloc(void** p, size_t n) { return hipMalloc(p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static thread_local const char* g_mock_bus = "0000:05:00.0";        // the harness gives every "rank" (thread) its device
inline hipError_t hipDeviceGetPCIBusId(char* b, int n, int) { strncpy(b, g_mock_bus, (size_t)n - 1); b[n - 1] = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 100000; return hipSuccess; }
inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) { memset(h, 0, sizeof *h); memcpy(h->reserved, &p, sizeof p); return hipSuccess; }
inline hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) { memcpy(p, h.reserved, sizeof *p); return hipSuccess; }
inline hipError_t hipIpcCloseMemHandle(void*) { return hipSuccess; }
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) do { (void)(grid); (void)(stream); } while (0)
// device intrinsics the (never executed) kernels mention
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
inline long long wall_clock64() { return 0; }
inline void __builtin_amdgcn_s_sleep(int) {}
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline int __double2loint(double d) { int64_t i; memcpy(&i, &d, 8); return (int)i; }
inline int __double2hiint(double d) { int64_t i; memcpy(&i, &d, 8); return (int)(i >> 32); }
inline double __hiloint2double(int hi, int lo) { int64_t i = ((int64_t)hi << 32) | (uint32_t)lo; double d; memcpy(&d, &i, 8); return d; }
