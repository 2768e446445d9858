// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A minimal CPU stand-in for the parts of the HIP runtime and device language that
// dwgsim_amd/csrc uses, so that the *same kernel sources* can be compiled with g++ and run
// thread-for-thread on the CPU (one OS thread per GPU thread of a block, blocks executed in order).
// It exists to diff the kernels' logic against the oracle in the dev container, which has no GPU.
// It is NOT a fallback: the product library (libdwgsim_hip.so) is built by hipcc for gfx950 only and
// never contains or loads this code.  Built by tests/emu/build.sh into tests/emu/libdwgsim_emu.so.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <pthread.h>
#include <sched.h>
#include <time.h>
#include <math.h>
#include <functional>
#include <thread>
#include <vector>

#define DW_EMU 1      // (nothing under dwgsim_amd/ looks at it: the emulation enters through tests/emu/dw_intrin.hpp)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { uint4 r = {a, b, c, d}; return r; }

namespace hipemu {
struct Idx { unsigned x, y, z; };
extern thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
void sync_block();
void sync_wave();
void yield();                     // the calling lane lets the other lanes of the block run (s_sleep in a polling loop)
uint64_t *wave_buf();             // 64 x u64 exchange slots of the calling thread's wave
void *dyn_shared();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &fn, const char *kernel_name = nullptr);
}
#define threadIdx hipemu::t_threadIdx
#define blockIdx hipemu::t_blockIdx
#define blockDim hipemu::t_blockDim
#define gridDim hipemu::t_gridDim

static inline void __syncthreads() { hipemu::sync_block(); }

template <typename T> static inline T emu_shfl_from(T v, int src)
{
    static_assert(sizeof(T) <= 8, "shuffle payload");
    uint64_t *b = hipemu::wave_buf(); const int lane = (int)(threadIdx.x & 63);
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T)); b[lane] = raw;
    hipemu::sync_wave();
    T r = v; if (src >= 0 && src < 64) { uint64_t o = b[src]; memcpy(&r, &o, sizeof(T)); }
    hipemu::sync_wave();
    return r;
}
template <typename T> static inline T __shfl_up(T v, int d) { const int l = (int)(threadIdx.x & 63); return emu_shfl_from(v, l - d >= 0 ? l - d : l); }
template <typename T> static inline T __shfl_down(T v, int d) { const int l = (int)(threadIdx.x & 63); return emu_shfl_from(v, l + d < 64 ? l + d : l); }
template <typename T> static inline T __shfl_xor(T v, int m) { const int l = (int)(threadIdx.x & 63); return emu_shfl_from(v, l ^ m); }
template <typename T> static inline T __shfl(T v, int src) { return emu_shfl_from(v, src & 63); }
static inline unsigned long long __ballot(int pred)
{
    uint64_t *b = hipemu::wave_buf(); const int lane = (int)(threadIdx.x & 63);
    b[lane] = pred ? 1 : 0;
    hipemu::sync_wave();
    unsigned long long m = 0; for (int i = 0; i < 64; ++i) if (b[i]) m |= 1ull << i;
    hipemu::sync_wave();
    return m;
}
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }

static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) { unsigned long long o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o > v && !__atomic_compare_exchange_n(p, &o, v, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)); return o; }
static inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { unsigned long long o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)); return o; }
static inline uint32_t atomicMax(uint32_t *p, uint32_t v) { uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)); return o; }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, order)
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, order)
static inline void __builtin_amdgcn_s_sleep(int) { hipemu::yield(); }

// hardware transcendentals used only inside estimates with a guard band (quality_pair_lazy): libm stand-ins are at least as accurate
#ifndef __clang__
static inline uint32_t __builtin_bitreverse32(uint32_t x)
{
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4); return __builtin_bswap32(x);
}
#endif
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31)); }
static inline uint32_t __builtin_amdgcn_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3))); }
static inline float __builtin_amdgcn_logf(float x) { return log2f(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }

// ---- host runtime ----
typedef int hipError_t;
#define hipSuccess 0
typedef struct hipemu_stream *hipStream_t;
typedef struct hipemu_event { double t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
#define hipHostMallocDefault 0
struct hipDeviceProp_t { char name[64]; char gcnArchName[64]; int multiProcessorCount; size_t totalGlobalMem; };
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
static inline hipError_t hipDeviceGetPCIBusId(char *b, int n, int) { if (n > 0) b[0] = 0; return 1; }      // (no bus on the emulator: the NUMA node is unknown)
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof *p); strcpy(p->name, "cpu-simt-emulator"); strcpy(p->gcnArchName, "emu"); p->multiProcessorCount = 1; return 0; }
// (exactly n bytes, so that AddressSanitizer sees an access past the REQUESTED size: round 5's advisor found a 16-byte overrun that a size rounded up to 256 hid)
static inline hipError_t hipMalloc(void **p, size_t n) { *p = nullptr; return posix_memalign(p, 256, n ? n : 1) == 0 ? 0 : 2; }
static inline hipError_t hipFree(void *p) { free(p); return 0; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = nullptr; return posix_memalign(p, 256, n ? n : 1) == 0 ? 0 : 2; }
static inline hipError_t hipHostFree(void *p) { free(p); return 0; }
#define hipHostRegisterDefault 0
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return 0; }
static inline hipError_t hipHostUnregister(void *) { return 0; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return 0; }
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { a->type = hipMemoryTypeUnregistered; return 1; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = (hipStream_t)malloc(1); return 0; }
#define hipStreamDefault 0
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = 0; return 0; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { return hipStreamCreate(s); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)calloc(1, sizeof(hipemu_event)); return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); e->t = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventQuery(hipEvent_t) { return 0; }
#define hipErrorNotReady 600
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return 0; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipemu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); }, #kernel)
