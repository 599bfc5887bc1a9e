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

// ---- live decoders: feat_s2mfc2feat_live called piece by piece (feat.c:1310-1420) with the running mean of cmn_live ---------------
//
// State of one stream, FeatLiveState words (floats unless noted), kept on the device between steps:
//   [0, C) cmn_mean   [C, 2C) sum   [2C] nframe (int)   [2C + 1] frames pending (int, 0..3)
//   [2C + 2 + k C, ..) k = 0..2: the three normalised cepstra BEFORE the next feature frame (the ring's cepbuf[curpos - 3 .. curpos - 1])
//   [2C + 2 + (3 + k) C, ..) k = 0..2: the pending ones (cepbuf[curpos ..]): received, not yet a feature frame's centre
// A step's work per stream is a list of OPS (n, flags) = the calls the reference's acmod makes for the step's audio (the host walks
// its buffer counters, psgpu_decode.hip LiveSim): flags bit 0 beginutt, bit 1 endutt, bit 2 "statistics only" (feat_update_stats at
// an utterance's end without a call).  Everything is per coefficient: lane i < C carries coefficient i of the mean, the sum, the
// window and the three outputs of a frame (c, delta, delta-delta: feat_1s_c_d_dd_cep2feat, feat.c:579-622); frames in order.
constexpr int kCmnWin = 500, kCmnWinHwm = 800;           // feat/cmn.h:146-147
__host__ __device__ constexpr int feat_live_state_words(int C) { return 2 * C + 2 + 6 * C; }

__global__ __launch_bounds__(64)
void feat_live_kernel(const float *__restrict__ cep, const int32_t *__restrict__ cep_off, const int32_t *__restrict__ ops,
                      const int32_t *__restrict__ op_off, const int32_t *__restrict__ feat_off, int32_t C,
                      float *__restrict__ state, float *__restrict__ out)
{
    const int u = blockIdx.x, i = threadIdx.x;
    if (i >= C) return;
    float *const st = state + (size_t)u * feat_live_state_words(C);
    float mean = st[i], sum = st[C + i];
    int nframe = reinterpret_cast<const int32_t *>(st)[2 * C], npend = reinterpret_cast<const int32_t *>(st)[2 * C + 1];
    float h0 = st[2 * C + 2 + i], h1 = st[2 * C + 2 + C + i], h2 = st[2 * C + 2 + 2 * C + i];
    float p0 = st[2 * C + 2 + 3 * C + i], p1 = st[2 * C + 2 + 4 * C + i], p2 = st[2 * C + 2 + 5 * C + i];
    const float *c = cep + (size_t)cep_off[u] * C;
    float *o = out + (size_t)feat_off[u] * 3 * C;
    // a normalised cepstrum arrives: the fourth pending one makes the oldest pending a feature frame's centre
    auto push = [&](float x) {
        if (npend < 3) { if (npend == 0) p0 = x; else if (npend == 1) p1 = x; else p2 = x; ++npend; return; }
        // window: h0 h1 h2 | p0 | p1 p2 x   (mfc[-3 .. 3] around p0)
        o[i] = p0;
        o[C + i] = __fsub_rn(p2, h1);                                          // mfc[2] - mfc[-2]
        o[2 * C + i] = __fsub_rn(__fsub_rn(x, h2), __fsub_rn(p1, h0));         // (mfc[3] - mfc[-1]) - (mfc[1] - mfc[-3])
        o += 3 * C;
        h0 = h1; h1 = h2; h2 = p0; p0 = p1; p1 = p2; p2 = x;
    };
    auto stats = [&](bool at_end) {            // cmn_live_shiftwin (>= HWM) / cmn_live_update (> HWM), cmn_live.c:65-117
        if (nframe <= 0) return;
        const float sf = __fdiv_rn(1.0f, (float)nframe);
        mean = __fdiv_rn(sum, (float)nframe);
        if (at_end ? nframe > kCmnWinHwm : nframe >= kCmnWinHwm) {
            sum = __fmul_rn(sum, __fmul_rn((float)kCmnWin, sf));
            nframe = kCmnWin;
        }
    };
    for (int k = op_off[u]; k < op_off[u + 1]; ++k) {
        const int n = ops[2 * k], fl = ops[2 * k + 1];
        const bool begin = fl & 1, end = fl & 2;
        if (fl & 4) { stats(true); continue; }
        if (begin) npend = 0;                                  // "empty the input buffer on start of utterance" (feat.c:1333-1335)
        for (int t = 0; t < n; ++t) {                          // cmn_live (cmn_live.c:119-150): frames with c0 < 0 pass untouched
            float x = c[(size_t)t * C + i];
            if (!(c[(size_t)t * C] < 0.0f)) { sum = __fadd_rn(sum, x); x = __fsub_rn(x, mean); ++nframe; }
            if (begin && t == 0) { h0 = h1 = h2 = x; }         // the first frame replicated into the window before it (:1360-1369)
            push(x);
        }
        c += (size_t)n * C;
        if (n > 0 && nframe > kCmnWinHwm) stats(false);
        if (end) {
            stats(true);                                       // feat_cmn: cmn_live_update when the piece ends the utterance (feat.c:930-934)
            const float last = npend == 0 ? h2 : (npend == 1 ? p0 : (npend == 2 ? p1 : p2));    // cepbuf[bufpos - 1] (:1382-1393)
            push(last); push(last); push(last);
        }
    }
    st[i] = mean; st[C + i] = sum;
    if (i == 0) { reinterpret_cast<int32_t *>(st)[2 * C] = nframe; reinterpret_cast<int32_t *>(st)[2 * C + 1] = npend; }
    st[2 * C + 2 + i] = h0; st[2 * C + 2 + C + i] = h1; st[2 * C + 2 + 2 * C + i] = h2;
    st[2 * C + 2 + 3 * C + i] = p0; st[2 * C + 2 + 4 * C + i] = p1; st[2 * C + 2 + 5 * C + i] = p2;
}

extern "C" {

int32_t psgpu_feat_live_state_words(int32_t cepsize) { return feat_live_state_words(cepsize); }

// a new decoder's state for a stream: the mean from -cmninit (cmn_live_set, cmn.c:113-146: sum = mean * CMN_WIN, nframe = CMN_WIN),
// an all-zero feature ring (feat_init's ckd_calloc)
int psgpu_feat_live_state_init(float *state_host, int32_t cepsize, const float *cmninit, int32_t n_init)
{
    PSGPU_REQUIRE(state_host && cepsize >= 1 && cepsize <= kFeatMaxCep && n_init >= 0 && n_init <= cepsize && (n_init == 0 || cmninit),
                  "psgpu_feat_live_state_init: bad argument");
    const int C = cepsize;
    for (int i = 0; i < feat_live_state_words(C); ++i) state_host[i] = 0.0f;
    for (int i = 0; i < n_init; ++i) { state_host[i] = cmninit[i]; state_host[C + i] = cmninit[i] * (float)kCmnWin; }
    reinterpret_cast<int32_t *>(state_host)[2 * C] = kCmnWin;
    reinterpret_cast<int32_t *>(state_host)[2 * C + 1] = 0;
    return PSGPU_OK;
}

int psgpu_feat_live_step_dev(const float *cep_dev, const int32_t *cep_off_dev, const int32_t *ops_dev, const int32_t *op_off_dev,
                             const int32_t *feat_off_dev, int32_t n_streams, int32_t cepsize, float *state_dev, float *feat_dev, void *stream)
{
    PSGPU_REQUIRE(cep_off_dev && ops_dev && op_off_dev && feat_off_dev && state_dev && feat_dev, "psgpu_feat_live_step_dev: NULL argument");
    PSGPU_REQUIRE(cepsize >= 1 && cepsize <= kFeatMaxCep && n_streams >= 0, "psgpu_feat_live_step_dev: bad argument");
    if (n_streams == 0) return PSGPU_OK;
    hipLaunchKernelGGL(feat_live_kernel, dim3(n_streams), dim3(64), 0, (hipStream_t)stream, cep_dev, cep_off_dev, ops_dev, op_off_dev,
                       feat_off_dev, cepsize, state_dev, feat_dev);
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

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
