// psgpu_internal.h -- shared by the HIP translation units of libpsgpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "psgpu.h"

void psgpu_set_error(const char *fmt, ...);
void psgpu_clear_error();

#define PSGPU_HIP(call)                                                       \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            psgpu_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,     \
                            hipGetErrorString(e_));                           \
            return (e_ == hipErrorOutOfMemory) ? PSGPU_ENOMEM                 \
                 : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice)    \
                       ? PSGPU_ENODEV : PSGPU_EHIP;                           \
        }                                                                     \
    } while (0)

#define PSGPU_REQUIRE(cond, ...)                                              \
    do {                                                                      \
        if (!(cond)) { psgpu_set_error(__VA_ARGS__); return PSGPU_EINVAL; }   \
    } while (0)

int psgpu_check_device();   // PSGPU_OK iff a gfx950 device is current

// psgpu_fe_process_utts_dev skips the upload of offsets that are unchanged since its previous call; a caller that writes the
// frame-offset buffer itself (or replaces it) says so
void psgpu_fe_offsets_dirty(psgpu_fe_t *fe);

// Reference constants (include/pocketsphinx/prim_type.h:168, hmm.h:73,84,
// tied_mgau_common.h:60,78-82)
constexpr int32_t kMaxNegInt32 = (int32_t)0x80000000;
constexpr int32_t kWorstScore = (int32_t)0xE0000000;
constexpr int kSenscrShift = 10;
constexpr int kMaxNegAscr = 96;

// A pointer that arrives inside a by-value kernel-argument struct (or is loaded from memory) is a generic pointer to the
// compiler: every access through it becomes flat_load / flat_store, which wait on both memory counters.  Device memory
// handed over by the host is global: saying so (generic -> address space 1 -> generic) lets the address-space inference
// turn those accesses into global_load / global_store.
template <typename T>
__device__ __forceinline__ T *psgpu_as_global(T *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (T *)(__attribute__((address_space(1))) T *)p;
#else
    return p;
#endif
}
