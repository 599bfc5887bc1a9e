// psgpu_core.hip -- device / memory / event plumbing of the C ABI (include/psgpu.h).
#include "psgpu_internal.h"
#include <cstring>

static thread_local char g_err[512] = "";

void psgpu_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

void psgpu_clear_error() { g_err[0] = 0; }

int psgpu_check_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        psgpu_set_error("no HIP device visible (%s); libpsgpu has no CPU fallback",
                        e == hipSuccess ? "count=0" : hipGetErrorString(e));
        return PSGPU_ENODEV;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        psgpu_set_error("hipGetDevice failed");
        return PSGPU_ENODEV;
    }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
        psgpu_set_error("hipGetDeviceProperties failed");
        return PSGPU_ENODEV;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        psgpu_set_error("device %d is %s; libpsgpu is built for gfx950 only", dev, p.gcnArchName);
        return PSGPU_ENODEV;
    }
    return PSGPU_OK;
}

extern "C" {

const char *psgpu_version(void) { return "psgpu 0.6 (gfx950): ptm, s2_semi, ms scorers + hmm_vit_eval + fwdtree / fwdflat searches"; }
int32_t psgpu_abi_version(void) { return PSGPU_ABI_VERSION; }
uint64_t psgpu_capabilities(void)
{
    return PSGPU_CAP_PTM | PSGPU_CAP_SEMI | PSGPU_CAP_MS | PSGPU_CAP_HMM | PSGPU_CAP_FE | PSGPU_CAP_FWDTREE | PSGPU_CAP_FWDFLAT
         | PSGPU_CAP_TRIE_LM | PSGPU_CAP_DECODE | PSGPU_CAP_STREAMS | PSGPU_CAP_PTM_BATCH_ANY_SHAPE | PSGPU_CAP_STREAMS_PCM | PSGPU_CAP_FEAT_TYPES | PSGPU_CAP_LM_SETS;      // (only what this build serves)
}
const char *psgpu_last_error(void) { return g_err; }

int psgpu_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        psgpu_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return PSGPU_ENODEV;
    }
    return n;
}

int psgpu_set_device(int device)
{
    PSGPU_HIP(hipSetDevice(device));
    return psgpu_check_device();
}

int psgpu_malloc(void **p, size_t bytes)
{
    PSGPU_REQUIRE(p != nullptr, "psgpu_malloc: NULL out pointer");
    PSGPU_HIP(hipMalloc(p, bytes ? bytes : 1));
    return PSGPU_OK;
}

int psgpu_free(void *p)
{
    if (p) PSGPU_HIP(hipFree(p));
    return PSGPU_OK;
}

int psgpu_get_device(void)
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) return PSGPU_ENODEV;
    return d;
}

int psgpu_host_alloc(void **p, size_t bytes)
{
    PSGPU_REQUIRE(p != nullptr, "psgpu_host_alloc: NULL argument");
    PSGPU_HIP(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
    return PSGPU_OK;
}

int psgpu_host_free(void *p)
{
    if (p) PSGPU_HIP(hipHostFree(p));
    return PSGPU_OK;
}

int psgpu_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    PSGPU_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return PSGPU_OK;
}

int psgpu_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    PSGPU_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return PSGPU_OK;
}

int psgpu_stream_sync(void *stream)
{
    PSGPU_HIP(hipStreamSynchronize((hipStream_t)stream));
    return PSGPU_OK;
}

int psgpu_stream_create_dedicated(void **stream)
{
    PSGPU_REQUIRE(stream != nullptr, "psgpu_stream_create_dedicated: NULL argument");
    *stream = nullptr;
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;
    int dev = 0;
    hipDeviceProp_t prop;
    PSGPU_HIP(hipGetDevice(&dev));
    PSGPU_HIP(hipGetDeviceProperties(&prop, dev));
    const uint32_t n_words = (uint32_t)((prop.multiProcessorCount + 31) / 32);
    uint32_t mask[64];
    PSGPU_REQUIRE(n_words >= 1 && n_words <= 64, "psgpu_stream_create_dedicated: %d compute units", prop.multiProcessorCount);
    for (uint32_t i = 0; i < n_words; ++i) mask[i] = 0xffffffffu;
    hipStream_t st;
    PSGPU_HIP(hipExtStreamCreateWithCUMask(&st, n_words, mask));
    *stream = (void *)st;
    return PSGPU_OK;
}

int psgpu_stream_destroy(void *stream)
{
    if (stream) PSGPU_HIP(hipStreamDestroy((hipStream_t)stream));
    return PSGPU_OK;
}

int psgpu_event_create(void **ev)
{
    hipEvent_t e;
    PSGPU_HIP(hipEventCreate(&e));
    *ev = (void *)e;
    return PSGPU_OK;
}

int psgpu_event_destroy(void *ev)
{
    PSGPU_HIP(hipEventDestroy((hipEvent_t)ev));
    return PSGPU_OK;
}

int psgpu_event_record(void *ev, void *stream)
{
    PSGPU_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return PSGPU_OK;
}

int psgpu_event_elapsed_ms(void *a, void *b, float *ms)
{
    PSGPU_HIP(hipEventSynchronize((hipEvent_t)b));
    PSGPU_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return PSGPU_OK;
}

}  // extern "C"
