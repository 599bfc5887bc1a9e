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
#include <cstdlib>
#include <cstring>
#include <vector>

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

// ---- every feature type feat_init knows (feat.c:704-915), whole utterances -----------------------------------------------------------
// A feature type is a RECIPE: output element e of a frame is one of the three forms every cep2feat function of the reference is made
// of -- a copy mfc[k1][i], a difference mfc[k1][i] - mfc[k2][i], a difference of differences (mfc[k1][i] - mfc[k2][i]) -
// (mfc[k3][i] - mfc[k4][i]) -- over the utterance's cepstra after cmn() (feat/cmn.c:166-233: batch mean, optionally unit variance) and
// agc_max (feat/agc.c:110-127), the first and last frame replicated over the window (feat_s2mfc2feat_block_utt, feat.c:1275-1306);
// then optionally the linear transform of -lda (feat_lda_transform, lda.c:140-159: out[j] = sum over k in order of in[k] * lda[j][k],
// single precision, product then sum) and the subvector projection of -svspec (feat_subvec_project, feat.c:333-352).  The host builds the
// recipe from the type's name exactly as feat_init parses it (psgpu_feat_create).  One workgroup per utterance.
struct FeatDev {
    int32_t cepsize, n_out, win, cmn, varnorm, agc, lda_out, n_sv, final_dim;
    const int32_t *ops;                  // [n_out][6] = {form 0 / 1 / 2, coefficient, k1, k2, k3, k4}
    const float *lda;                    // [lda_out][n_out] or NULL
    const int32_t *sv;                   // [n_sv] component of the (transformed) vector each output takes, or NULL
};
struct psgpu_feat_s { FeatDev d; void *blob; std::vector<int32_t> stream_len; };

constexpr int kFeatTile = 8;             // frames a workgroup holds in LDS for the transform

__global__ __launch_bounds__(kFeatThreads)
void feat_recipe_kernel(FeatDev p, const float *__restrict__ cep, const int32_t *__restrict__ utt_off, float *__restrict__ out)
{
    extern __shared__ float s_f[];       // [kFeatTile][n_out] (+ [kFeatTile][lda_out] when subvectors follow a transform)
    __shared__ float s_mean[kFeatMaxCep], s_inv[kFeatMaxCep], s_max;
    const int u = blockIdx.x, C = p.cepsize;
    const int t0 = utt_off[u], T = utt_off[u + 1] - t0;
    if (T <= 0) return;
    const float *c = cep + (size_t)t0 * C;
    const int tid = threadIdx.x;
    if (tid < C) {
        float mean = 0.0f, inv = 1.0f;
        if (p.cmn) {                                       // cmn(): frames with c0 < 0 do not count for the mean, every frame for the variance
            float sum = 0.0f; int n = 0;
            for (int t = 0; t < T; ++t) {
                if (c[(size_t)t * C] < 0.0f) continue;
                sum = __fadd_rn(sum, c[(size_t)t * C + tid]);
                ++n;
            }
            mean = __fdiv_rn(sum, (float)n);
            if (p.varnorm) {
                float var = 0.0f;
                for (int t = 0; t < T; ++t) { const float d = __fsub_rn(c[(size_t)t * C + tid], mean); var = __fadd_rn(var, __fmul_rn(d, d)); }
                inv = (float)__dsqrt_rn(__ddiv_rn((double)T, (double)var));
            }
        }
        s_mean[tid] = mean; s_inv[tid] = inv;
    }
    __syncthreads();
    auto norm = [&](float x, int i) {
        if (!p.cmn) return x;
        const float d = __fsub_rn(x, s_mean[i]);
        return p.varnorm ? __fmul_rn(d, s_inv[i]) : d;
    };
    if (p.agc) {                                           // agc_max on the normalised c0 (feat_agc follows feat_cmn, feat.c:1292-1293)
        if (tid < 64) {
            float m = -3.4e38f;
            for (int t = tid; t < T; t += 64) m = fmaxf(m, c[(size_t)t * C]);
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (tid == 0) s_max = norm(m, 0);              // (x -> norm(x) never reverses an order: the maximum's image is the images' maximum)
        }
        __syncthreads();
    }
    auto at = [&](int t, int k, int i) {
        int tt = t + k;
        tt = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
        float v = norm(c[(size_t)tt * C + i], i);
        if (p.agc && i == 0) v = __fsub_rn(v, s_max);
        return v;
    };
    auto elem = [&](int t, int e) {
        const int32_t *op = p.ops + 6 * e;
        const int i = op[1];
        if (op[0] == 0) return at(t, op[2], i);
        const float d1 = __fsub_rn(at(t, op[2], i), at(t, op[3], i));
        if (op[0] == 1) return d1;
        return __fsub_rn(d1, __fsub_rn(at(t, op[4], i), at(t, op[5], i)));
    };
    float *o = out + (size_t)t0 * p.final_dim;
    if (!p.lda && !p.sv) {
        for (int e = tid; e < T * p.n_out; e += kFeatThreads) { const int t = e / p.n_out; o[e] = elem(t, e - t * p.n_out); }
        return;
    }
    float *s_l = s_f + kFeatTile * p.n_out;
    const int mid = p.lda ? p.lda_out : p.n_out;
    for (int tb = 0; tb < T; tb += kFeatTile) {
        const int nt = min(kFeatTile, T - tb);
        for (int e = tid; e < nt * p.n_out; e += kFeatThreads) { const int t = e / p.n_out; s_f[e] = elem(tb + t, e - t * p.n_out); }
        __syncthreads();
        const float *src = s_f;
        if (p.lda) {
            for (int e = tid; e < nt * p.lda_out; e += kFeatThreads) {
                const int t = e / p.lda_out, j = e - t * p.lda_out;
                const float *f = s_f + t * p.n_out, *l = p.lda + (size_t)j * p.n_out;
                float a = 0.0f;
                for (int k = 0; k < p.n_out; ++k) a = __fadd_rn(a, __fmul_rn(f[k], l[k]));
                if (p.sv) s_l[e] = a; else o[(size_t)(tb + t) * p.final_dim + j] = a;
            }
            src = s_l;
            if (p.sv) __syncthreads();
        }
        if (p.sv)
            for (int e = tid; e < nt * p.n_sv; e += kFeatThreads) {
                const int t = e / p.n_sv, q = e - t * p.n_sv;
                o[(size_t)(tb + t) * p.final_dim + q] = src[t * mid + p.sv[q]];
            }
        __syncthreads();
    }
}

extern "C" {

// feat_init's parse of the type's name (feat.c:704-896) into a recipe.  cmn: 0 none, 1 batch ("current" / "batch": cmn()); agc: 0 none,
// 1 max.  lda [lda_out][lda_in] (lda_in = the type's dimension) or NULL; subvec [n_subvec] (components, in -svspec's order) or NULL.
int psgpu_feat_create(psgpu_feat_t **out, const char *type, int32_t cepsize, int32_t cmn, int32_t varnorm, int32_t agc, const float *lda,
                      int32_t lda_out, int32_t lda_in, const int32_t *subvec, int32_t n_subvec)
{
    PSGPU_REQUIRE(out && type, "psgpu_feat_create: NULL argument");
    PSGPU_REQUIRE(cmn >= 0 && cmn <= 1 && agc >= 0 && agc <= 1, "psgpu_feat_create: cmn none / batch and agc none / max are served (agc emax / noise are not)");
    *out = nullptr;
    if (cepsize == 0) cepsize = 13;
    PSGPU_REQUIRE(cepsize >= 1 && cepsize <= kFeatMaxCep, "psgpu_feat_create: cepsize %d outside 1..%d", cepsize, kFeatMaxCep);
    std::vector<int32_t> ops, slen;
    int win = 0;
    auto put = [&](int form, int i, int k1, int k2, int k3, int k4) { const int32_t r[6] = { form, i, k1, k2, k3, k4 }; ops.insert(ops.end(), r, r + 6); };
    const int C = cepsize;
    if (!strcmp(type, "s2_4x")) {                          // feat_s2_4x_cep2feat (feat.c:424-485): 12 cep | 12 + 12 dcep | pow | 12 ddcep
        PSGPU_REQUIRE(C == 13, "psgpu_feat_create: s2_4x features require cepsize == 13");
        win = 4; slen = { 12, 24, 3, 12 };
        for (int i = 1; i < 13; ++i) put(0, i, 0, 0, 0, 0);
        for (int i = 1; i < 13; ++i) put(1, i, 2, -2, 0, 0);
        for (int i = 1; i < 13; ++i) put(1, i, 4, -4, 0, 0);
        put(0, 0, 0, 0, 0, 0); put(1, 0, 2, -2, 0, 0); put(2, 0, 3, -1, 1, -3);
        for (int i = 1; i < 13; ++i) put(2, i, 3, -1, 1, -3);
    }
    else if (!strcmp(type, "s3_1x39") || !strcmp(type, "1s_12c_12d_3p_12dd")) {      // feat_s3_1x39_cep2feat (:487-540)
        PSGPU_REQUIRE(C == 13, "psgpu_feat_create: s3_1x39 features require cepsize == 13");
        win = 3; slen = { 39 };
        for (int i = 1; i < 13; ++i) put(0, i, 0, 0, 0, 0);
        for (int i = 1; i < 13; ++i) put(1, i, 2, -2, 0, 0);
        put(0, 0, 0, 0, 0, 0); put(1, 0, 2, -2, 0, 0); put(2, 0, 3, -1, 1, -3);
        for (int i = 1; i < 13; ++i) put(2, i, 3, -1, 1, -3);
    }
    else if (!strncmp(type, "1s_c_d_dd", 9)) {             // feat_1s_c_d_dd_cep2feat (:578-621)
        win = 3; slen = { 3 * C };
        for (int i = 0; i < C; ++i) put(0, i, 0, 0, 0, 0);
        for (int i = 0; i < C; ++i) put(1, i, 2, -2, 0, 0);
        for (int i = 0; i < C; ++i) put(2, i, 3, -1, 1, -3);
    }
    else if (!strncmp(type, "1s_c_d_ld_dd", 12)) {         // feat_1s_c_d_ld_dd_cep2feat (:624-675)
        win = 4; slen = { 4 * C };
        for (int i = 0; i < C; ++i) put(0, i, 0, 0, 0, 0);
        for (int i = 0; i < C; ++i) put(1, i, 2, -2, 0, 0);
        for (int i = 0; i < C; ++i) put(1, i, 4, -4, 0, 0);
        for (int i = 0; i < C; ++i) put(2, i, 3, -1, 1, -3);
    }
    else if (!strncmp(type, "cep_dcep", 8) || !strncmp(type, "1s_c_d", 6)) {           // feat_s3_cep_dcep (:553-576)
        win = 2; slen = { 2 * C };
        for (int i = 0; i < C; ++i) put(0, i, 0, 0, 0, 0);
        for (int i = 0; i < C; ++i) put(1, i, 2, -2, 0, 0);
    }
    else if (!strncmp(type, "cep", 3) || !strncmp(type, "1s_c", 4)) {                  // feat_s3_cep (:542-551)
        win = 0; slen = { C };
        for (int i = 0; i < C; ++i) put(0, i, 0, 0, 0, 0);
    }
    else {
        // "1s_3c" / "1s_4c" (frames concatenated), or the generic "%d,%d,...[:window]" (feat_copy, :677-700: per stream, the window's
        // frames' shares of the input vector side by side)
        std::vector<int> widths;
        if (!strncmp(type, "1s_3c", 5) || !strncmp(type, "1s_4c", 5)) { win = type[3] == '3' ? 3 : 4; widths = { C }; }
        else {
            const char *q = type;
            int tot = 0;
            for (;;) {
                char *end;
                const long v = strtol(q, &end, 10);
                PSGPU_REQUIRE(end != q && v > 0, "psgpu_feat_create: bad feature type '%s'", type);
                widths.push_back((int)v); tot += (int)v;
                if (*end == ',') { q = end + 1; continue; }
                if (*end == ':') { win = atoi(end + 1); end += strlen(end); }
                PSGPU_REQUIRE(*end == 0, "psgpu_feat_create: bad feature type '%s'", type);
                break;
            }
            PSGPU_REQUIRE(tot == C && win >= 0 && win <= 16, "psgpu_feat_create: feature type '%s' does not add up to cepsize %d", type, C);
        }
        int spos = 0;
        for (int w_ : widths) {
            for (int k = -win; k <= win; ++k) for (int i = 0; i < w_; ++i) put(0, spos + i, k, 0, 0, 0);
            slen.push_back(w_ * (2 * win + 1));
            spos += w_;
        }
    }
    const int n_out = (int)(ops.size() / 6);
    PSGPU_REQUIRE(!lda || (slen.size() == 1 && lda_in == n_out && lda_out >= 1 && lda_out <= n_out),
                  "psgpu_feat_create: a transform of %d x %d for a feature type of %zu stream(s), dimension %d (feat_read_lda, lda.c:63-138)", lda_out, lda_in,
                  slen.size(), n_out);
    const int mid = lda ? lda_out : n_out;
    for (int q = 0; q < n_subvec; ++q) PSGPU_REQUIRE(subvec && subvec[q] >= 0 && subvec[q] < mid, "psgpu_feat_create: subvector component %d outside 0..%d", subvec ? subvec[q] : -1, mid - 1);
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;
    psgpu_feat_s *f = new psgpu_feat_s();
    f->stream_len = slen;
    FeatDev &d = f->d;
    d.cepsize = C; d.n_out = n_out; d.win = win; d.cmn = cmn; d.varnorm = cmn ? varnorm : 0; d.agc = agc; d.lda_out = lda ? lda_out : 0;
    d.n_sv = n_subvec > 0 ? n_subvec : 0; d.final_dim = d.n_sv ? d.n_sv : mid;
    const size_t b_ops = 4 * ops.size(), b_lda = lda ? 4 * (size_t)lda_out * n_out : 0, b_sv = 4 * (size_t)d.n_sv;
    std::vector<uint8_t> h(b_ops + b_lda + b_sv);
    memcpy(h.data(), ops.data(), b_ops);
    if (lda) memcpy(h.data() + b_ops, lda, b_lda);
    if (d.n_sv) memcpy(h.data() + b_ops + b_lda, subvec, b_sv);
    hipError_t e = hipMalloc(&f->blob, h.size());
    if (e == hipSuccess) e = hipMemcpy(f->blob, h.data(), h.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { psgpu_set_error("psgpu_feat_create: %s", hipGetErrorString(e)); hipFree(f->blob); delete f; return e == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP; }
    const uint8_t *b = (const uint8_t *)f->blob;
    d.ops = (const int32_t *)b; d.lda = lda ? (const float *)(b + b_ops) : nullptr; d.sv = d.n_sv ? (const int32_t *)(b + b_ops + b_lda) : nullptr;
    *out = f;
    return PSGPU_OK;
}

void psgpu_feat_free(psgpu_feat_t *f) { if (f) { hipFree(f->blob); delete f; } }
int32_t psgpu_feat_out_dim(const psgpu_feat_t *f) { return f ? f->d.final_dim : 0; }
int32_t psgpu_feat_cepsize(const psgpu_feat_t *f) { return f ? f->d.cepsize : 0; }
int32_t psgpu_feat_window(const psgpu_feat_t *f) { return f ? f->d.win : 0; }

int psgpu_feat_compute_dev(const psgpu_feat_t *f, const float *cep_dev, const int32_t *utt_off_dev, int32_t n_utt, float *feat_dev, void *stream)
{
    PSGPU_REQUIRE(f && cep_dev && utt_off_dev && feat_dev && n_utt >= 0, "psgpu_feat_compute_dev: bad argument");
    if (n_utt == 0) return PSGPU_OK;
    const FeatDev &d = f->d;
    const size_t lds = (d.lda || d.sv) ? 4 * (size_t)kFeatTile * (d.n_out + (d.lda && d.sv ? d.lda_out : 0)) : 0;
    PSGPU_REQUIRE(lds <= 60 * 1024, "psgpu_feat_compute_dev: feature dimension %d too large", d.n_out);
    hipLaunchKernelGGL(feat_recipe_kernel, dim3(n_utt), dim3(kFeatThreads), lds, (hipStream_t)stream, d, cep_dev, utt_off_dev, feat_dev);
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

int psgpu_feat_compute(const psgpu_feat_t *f, const float *cep, const int32_t *utt_off, int32_t n_utt, float *feat)
{
    PSGPU_REQUIRE(f && cep && utt_off && feat && n_utt >= 0, "psgpu_feat_compute: bad argument");
    if (n_utt == 0) return PSGPU_OK;
    const int32_t T = utt_off[n_utt];
    PSGPU_REQUIRE(T >= 0 && utt_off[0] == 0, "utt_off must start at 0");
    if (T == 0) return PSGPU_OK;
    float *dc = nullptr, *df = nullptr; int32_t *doff = nullptr;
    auto cleanup = [&]() { hipFree(dc); hipFree(df); hipFree(doff); };
#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) {                 \
        psgpu_set_error("%s -> %s", #call, hipGetErrorString(e_)); cleanup();          \
        return e_ == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP; } } while (0)
    TRY(hipMalloc((void **)&dc, (size_t)T * f->d.cepsize * sizeof(float)));
    TRY(hipMalloc((void **)&df, (size_t)T * f->d.final_dim * sizeof(float)));
    TRY(hipMalloc((void **)&doff, (size_t)(n_utt + 1) * sizeof(int32_t)));
    TRY(hipMemcpy(dc, cep, (size_t)T * f->d.cepsize * sizeof(float), hipMemcpyHostToDevice));
    TRY(hipMemcpy(doff, utt_off, (size_t)(n_utt + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    int rc = psgpu_feat_compute_dev(f, dc, doff, n_utt, df, nullptr);
    if (rc == PSGPU_OK) {
        TRY(hipDeviceSynchronize());
        TRY(hipMemcpy(feat, df, (size_t)T * f->d.final_dim * sizeof(float), hipMemcpyDeviceToHost));
    }
#undef TRY
    cleanup();
    return rc;
}

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
