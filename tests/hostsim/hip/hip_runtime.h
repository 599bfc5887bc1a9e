// tests/hostsim/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A workgroup simulator for CPU-only checks of kernel LOGIC: the unmodified kernel sources of
// pocketsphinx_amd/csrc (psgpu_search.hip, psgpu_flat.hip, psgpu_lm.hip) are compiled with g++ against this header
// instead of the HIP runtime, and a launch runs every workgroup on the host: one fiber per
// work-item, cooperative switches at __syncthreads() and at the cross-lane operations, so the
// barrier structure, prefix sums, list orders and table contents a kernel produces can be compared
// with the goldens where there is no GPU (tests/test_search_hostsim.py).  The order in which the
// fibers of a workgroup run between barriers can be reversed or shuffled (PSGPU_SIM_ORDER): a
// result that depends on it is a missing barrier.
//
// This is not a product path and not a portability layer: libpsgpu.so is built by hipcc for gfx950
// only and has no host fallback; nothing under pocketsphinx_amd/ knows this file exists.  What it
// cannot show: anything about timing, memory coherence between workgroups, register pressure.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

namespace hipsim {
struct Fiber;
extern thread_local Fiber *cur;
extern thread_local dim3 v_threadIdx, v_blockIdx, v_blockDim, v_gridDim;
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
void sync_block();
uint32_t xchg_wave(uint32_t v, int src_lane_of_me(int lane, int arg), int arg);   // value of lane f(lane) (own if outside 0..63)
uint64_t ballot_wave(bool pred);
}  // namespace hipsim

#define threadIdx (hipsim::v_threadIdx)
#define blockIdx (hipsim::v_blockIdx)
#define blockDim (hipsim::v_blockDim)
#define gridDim (hipsim::v_gridDim)

inline void __syncthreads() { hipsim::sync_block(); }

// ---- cross-lane (wave64) -------------------------------------------------------------------------
namespace hipsim {
inline int f_up(int lane, int d) { return lane - d; }
inline int f_down(int lane, int d) { return lane + d; }
inline int f_xor(int lane, int m) { return lane ^ m; }
inline int f_idx(int, int i) { return i & 63; }
template <typename T> inline T xchg(T v, int f(int, int), int arg)
{
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    uint32_t b; memcpy(&b, &v, 4);
    b = xchg_wave(b, f, arg);
    memcpy(&v, &b, 4);
    return v;
}
}  // namespace hipsim
template <typename T> inline T __shfl_up(T v, unsigned d) { return hipsim::xchg(v, hipsim::f_up, (int)d); }
template <typename T> inline T __shfl_down(T v, unsigned d) { return hipsim::xchg(v, hipsim::f_down, (int)d); }
template <typename T> inline T __shfl_xor(T v, int m) { return hipsim::xchg(v, hipsim::f_xor, m); }
template <typename T> inline T __shfl(T v, int i) { return hipsim::xchg(v, hipsim::f_idx, i); }
inline unsigned long long __ballot(int p) { return hipsim::ballot_wave(p != 0); }

// ---- integer / float intrinsics ------------------------------------------------------------------
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
// unfused, round-to-nearest: the sim library is built with -ffp-contract=off
inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
using std::max;
using std::min;
template <typename T> inline T __builtin_nontemporal_load(const T *p) { return *p; }

// ---- atomics (one host thread runs all fibers: plain read-modify-write) ----------------------------
template <typename T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T *p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

// ---- the slice of the runtime API the host side of those files uses -------------------------------
typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorInvalidDevice = 101 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
enum { hipHostMallocDefault = 0 };
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t)
{
    for (size_t r = 0; r < height; ++r) memcpy((char *)d + r * dpitch, (const char *)s + r * spitch, width);
    return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int *v, int, int) { *v = 2; return hipSuccess; }      // ("two compute units": launches of three utterances and more take the launch-order path)
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "host simulation"; }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipsim::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
