// psgpu_wave_dev.h -- scans over the 64 lanes of a wavefront, shared by the search kernels (psgpu_search.hip, psgpu_flat.hip).
#pragma once
#include <cstdint>

// ---- wavefront scans ------------------------------------------------------------------------------------------------
// Inclusive scans over the 64 lanes by data-parallel-primitive moves (row shifts within 16 lanes, then the two row broadcasts):
// six VALU operations instead of six trips through the LDS crossbar (__shfl_up is ds_bpermute, an LDS-latency operation, and
// a frame runs some eighty of them in sequence).  EVERY lane of the wavefront must be active.  The host build (the workgroup
// simulator) keeps the shuffle form.
struct FtAdd { static constexpr int32_t id = 0; static __device__ __forceinline__ int32_t op(int32_t a, int32_t b) { return a + b; } };
struct FtMax { static constexpr int32_t id = (int32_t)0x80000000; static __device__ __forceinline__ int32_t op(int32_t a, int32_t b) { return a > b ? a : b; } };
struct FtMin { static constexpr int32_t id = 0x7fffffff; static __device__ __forceinline__ int32_t op(int32_t a, int32_t b) { return a < b ? a : b; } };
template <typename OP>
__device__ __forceinline__ int32_t ft_wave_incl(int32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    v = OP::op(v, __builtin_amdgcn_update_dpp(OP::id, v, 0x111, 0xf, 0xf, false));      // row_shr:1
    v = OP::op(v, __builtin_amdgcn_update_dpp(OP::id, v, 0x112, 0xf, 0xf, false));      // row_shr:2
    v = OP::op(v, __builtin_amdgcn_update_dpp(OP::id, v, 0x114, 0xf, 0xf, false));      // row_shr:4
    v = OP::op(v, __builtin_amdgcn_update_dpp(OP::id, v, 0x118, 0xf, 0xf, false));      // row_shr:8
    v = OP::op(v, __builtin_amdgcn_update_dpp(OP::id, v, 0x142, 0xa, 0xf, false));      // row_bcast:15 into rows 1 and 3
    v = OP::op(v, __builtin_amdgcn_update_dpp(OP::id, v, 0x143, 0xc, 0xf, false));      // row_bcast:31 into rows 2 and 3
    return v;
#else
    const int lane = threadIdx.x & 63;
    for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(v, d); if (lane >= d) v = OP::op(v, o); }
    return v;
#endif
}
// the scan of the lanes BEFORE this one (lane 0: the identity)
template <typename OP>
__device__ __forceinline__ int32_t ft_wave_excl(int32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return ft_wave_incl<OP>(__builtin_amdgcn_update_dpp(OP::id, v, 0x138, 0xf, 0xf, false));   // wave_shr:1
#else
    int32_t s = __shfl_up(v, 1);
    if ((threadIdx.x & 63) == 0) s = OP::id;
    return ft_wave_incl<OP>(s);
#endif
}
// lane `l`'s value (l uniform over the wavefront)
__device__ __forceinline__ int32_t ft_lane(int32_t v, int l)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readlane(v, l);
#else
    return __shfl(v, l);
#endif
}
