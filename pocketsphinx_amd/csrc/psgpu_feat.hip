// psgpu_feat.hip -- whole-utterance dynamic feature computation on gfx950:
// cepstra -> "1s_c_d_dd" features with batch CMN, for batches of utterances.
//
// Replaces, bit-exactly, feat_s2mfc2feat_live(begin = end = TRUE) (reference
// src/feat/feat.c:1310 -> feat_s2mfc2feat_block_utt :1275-1306) for the feature
// type the PTM / continuous models use: cmn() (feat/cmn.c:166-208), replication
// of the first and last frame over a window of 3, feat_1s_c_d_dd_cep2feat
// (feat.c:579-622).  The subvector split of en-us (-svspec 0-12/13-25/26-38) is
// the identity on this layout.  This is the caller-side row SURVEY 8f-1; the
// MFCC front end in front of it is csrc/psgpu_fe.hip.
//
// One workgroup per utterance.  The mean must be summed in frame order in fp32
// to match the reference, so lane i (< cepsize) walks its coefficient over the
// frames sequentially (T dependent adds -- microseconds); everything else is
// element-wise and fully parallel.  HBM: 4*cepsize bytes in, 12*cepsize out per
// frame.
#include "psgpu_internal.h"

constexpr int kFeatThreads = 256;
constexpr int kFeatMaxCep = 64;

__global__ __launch_bounds__(kFeatThreads)
void feat_1s_c_d_dd_kernel(const float *__restrict__ cep, const int32_t *__restrict__ utt_off,
                           int32_t cepsize, float *__restrict__ out)
{
    __shared__ float s_mean[kFeatMaxCep];
    const int u = blockIdx.x;
    const int t0 = utt_off[u], T = utt_off[u + 1] - t0;
    if (T <= 0)
        return;
    const float *c = cep + (size_t)t0 * cepsize;
    if ((int)threadIdx.x < cepsize) {
        // cmn.c:182-198: frames with c0 < 0 are skipped, sums run in frame order
        float sum = 0.0f;
        int n = 0;
        for (int t = 0; t < T; ++t) {
            if (c[(size_t)t * cepsize] < 0.0f) continue;
            sum = __fadd_rn(sum, c[(size_t)t * cepsize + threadIdx.x]);
            ++n;
        }
        s_mean[threadIdx.x] = __fdiv_rn(sum, (float)n);
    }
    __syncthreads();
    const int total = T * cepsize;
    float *o = out + (size_t)t0 * 3 * cepsize;
    for (int e = threadIdx.x; e < total; e += kFeatThreads) {
        const int t = e / cepsize, i = e - t * cepsize;
        const float m = s_mean[i];
        // normalised cepstrum of frame t + k with the edge frames replicated (feat.c:1295-1302)
        auto at = [&](int k) {
            int tt = t + k;
            tt = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
            return __fsub_rn(c[(size_t)tt * cepsize + i], m);
        };
        float *f = o + (size_t)t * 3 * cepsize;
        f[i] = at(0);
        f[cepsize + i] = __fsub_rn(at(2), at(-2));
        f[2 * cepsize + i] = __fsub_rn(__fsub_rn(at(3), at(-1)), __fsub_rn(at(1), at(-3)));
    }
}

extern "C" {

int psgpu_feat_1s_c_d_dd_dev(const float *cep_dev, const int32_t *utt_off_dev, int32_t n_utt,
                             int32_t cepsize, float *feat_dev, void *stream)
{
    PSGPU_REQUIRE(cep_dev && utt_off_dev && feat_dev, "psgpu_feat_1s_c_d_dd_dev: NULL argument");
    PSGPU_REQUIRE(cepsize >= 1 && cepsize <= kFeatMaxCep, "cepsize %d outside 1..%d", cepsize, kFeatMaxCep);
    PSGPU_REQUIRE(n_utt >= 0, "negative utterance count");
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;
    if (n_utt == 0) return PSGPU_OK;
    hipLaunchKernelGGL(feat_1s_c_d_dd_kernel, dim3(n_utt), dim3(kFeatThreads), 0, (hipStream_t)stream,
                       cep_dev, utt_off_dev, cepsize, feat_dev);
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

int psgpu_feat_1s_c_d_dd(const float *cep, const int32_t *utt_off, int32_t n_utt, int32_t cepsize, float *feat)
{
    PSGPU_REQUIRE(cep && utt_off && feat && n_utt >= 0, "psgpu_feat_1s_c_d_dd: bad argument");
    if (n_utt == 0) return PSGPU_OK;
    const int32_t T = utt_off[n_utt];
    PSGPU_REQUIRE(T >= 0 && utt_off[0] == 0, "utt_off must start at 0");
    if (T == 0) return PSGPU_OK;
    float *dc = nullptr, *df = nullptr; int32_t *doff = nullptr;
    auto cleanup = [&]() { hipFree(dc); hipFree(df); hipFree(doff); };
#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) {                 \
        psgpu_set_error("%s -> %s", #call, hipGetErrorString(e_)); cleanup();          \
        return e_ == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP; } } while (0)
    TRY(hipMalloc((void **)&dc, (size_t)T * cepsize * sizeof(float)));
    TRY(hipMalloc((void **)&df, (size_t)T * 3 * cepsize * sizeof(float)));
    TRY(hipMalloc((void **)&doff, (size_t)(n_utt + 1) * sizeof(int32_t)));
    TRY(hipMemcpy(dc, cep, (size_t)T * cepsize * sizeof(float), hipMemcpyHostToDevice));
    TRY(hipMemcpy(doff, utt_off, (size_t)(n_utt + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    int rc = psgpu_feat_1s_c_d_dd_dev(dc, doff, n_utt, cepsize, df, nullptr);
    if (rc == PSGPU_OK) {
        TRY(hipDeviceSynchronize());
        TRY(hipMemcpy(feat, df, (size_t)T * 3 * cepsize * sizeof(float), hipMemcpyDeviceToHost));
    }
#undef TRY
    cleanup();
    return rc;
}

}  // extern "C"
