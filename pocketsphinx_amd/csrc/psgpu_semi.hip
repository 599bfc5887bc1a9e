// psgpu_semi.hip -- the semi-continuous scorer: device replacement of
// s2_semi_mgau_frame_eval() (reference src/s2_semi_mgau.c:836-883) with its
// full call contract (one shared codebook, n_feat streams of different
// lengths, top-N with frame-to-frame seeding, per-stream top-N beams, history
// ring topn_hist[pl_window+2] (:1301-1322), 8-bit and 4-bit clustered mixture
// weights, int16 accumulation without final normalisation).
//
// One launch per call: one workgroup of 8 wavefronts.  Wave f < n_feat owns
// stream f: up to 256 Gaussians, 4 codewords per lane (cw = k*64 + lane), the
// fp32 distances in the reference's exact op order, then the history-dependent
// top-N update emulated on wave-uniform state with ballots.  After a barrier
// all lanes score the listed senones (mixture weights from L2, 8-bit log-add
// table in LDS) and store int16 scores into a host-mapped buffer.
#include "psgpu_ptm_dev.h"
#include <cstring>
#include <cstdlib>

constexpr int kSemiMaxFeat = 8;
constexpr int kSemiMaxTopn = 8;
constexpr int kSemiK = kGenK;             // codewords per lane -> n_density <= 256
constexpr int kSemiMaxVec = 64;
constexpr int kSemiThreads = 512;
constexpr int kSemiLa = 512;

struct SemiFeat { float x[kSemiMaxVec]; };

struct SemiDev {
    const float *mean, *var, *det;
    const uint8_t *mixw, *mixw_cb, *logadd8;       // mixw_cb == nullptr: 8-bit weights
    int32_t n_feat, n_density, n_sen, topn, ds_ratio, logadd8_size, row;   // row = bytes per mixw row
    int32_t featlen[kSemiMaxFeat], featoff[kSemiMaxFeat], foff[kSemiMaxFeat];
    int32_t beam[kSemiMaxFeat];
};

struct psgpu_semi_model_s {
    SemiDev d;
    float *mean, *var, *det;
    uint8_t *mixw, *mixw_cb, *logadd8;
    int32_t veclen;
    // batched entry scratch: lists of every frame
    uint8_t *b_cw, *b_n; int32_t *b_sc; int64_t b_cap;
};

struct psgpu_semi_state_s {
    psgpu_semi_model_t *m;
    int32_t n_hist;
    int32_t *hist_cw, *hist_sc;      // [n_hist][n_feat][topn]
    int32_t *hist_n;                 // [n_hist][n_feat]   (topn_hist_n)
    uint16_t *h_list, *d_list;
    int16_t *h_out, *d_out;
    hipStream_t stream;
    int32_t cur;
    uint32_t seq;
};

template <int N>
__global__ __launch_bounds__(kSemiThreads)
void semi_frame_kernel(SemiDev p, SemiFeat fa, int32_t fresh, int32_t do_scan, int32_t compall,
                       int32_t n_list, const uint16_t *__restrict__ list,
                       const int32_t *__restrict__ prev_cw,
                       int32_t *__restrict__ cur_cw, int32_t *__restrict__ cur_sc,
                       int32_t *__restrict__ cur_n, int16_t *__restrict__ out,
                       uint32_t *__restrict__ done_word, uint32_t seq)
{
    __shared__ uint8_t s_la[kSemiLa];
    __shared__ int32_t s_cw[kSemiMaxFeat * N], s_sc[kSemiMaxFeat * N], s_n[kSemiMaxFeat];
    __shared__ uint8_t s_cb[16];
    extern __shared__ __attribute__((aligned(16))) int16_t s_out[];       // [n_sen]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < kSemiLa; i += kSemiThreads)
        s_la[i] = (i < p.logadd8_size) ? p.logadd8[i] : 0;
    if (tid < 16) s_cb[tid] = p.mixw_cb ? p.mixw_cb[tid] : 0;

    if (wave < p.n_feat) {
        const int f = wave;
        TopN<N> L;
        int32_t cnt;
        if (fresh) {
            const int len = p.featlen[f];
            const float *mean = p.mean + p.foff[f], *var = p.var + p.foff[f];
            const float *det = p.det + (size_t)f * p.n_density;
            const float *x = fa.x + p.featoff[f];
            float d[kSemiK], dp[kSemiK];
#pragma unroll
            for (int k = 0; k < kSemiK; ++k) {
                const int cw = min(k * 64 + lane, p.n_density - 1);     // clamp; masked in the scan
                const float *m = mean + (size_t)cw * len, *v = var + (size_t)cw * len;
                float acc = det[cw], prev = acc;
                for (int j = 0; j < len; ++j) {
                    prev = acc;
                    acc = gau_step(acc, x[j], m[j], v[j]);
                }
                d[k] = acc; dp[k] = prev;
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                L.cw[i] = __builtin_amdgcn_readfirstlane(prev_cw[f * N + i]);
                L.sc[i] = kMaxNegInt32;
            }
            generic_frame_step<N, true>(L, d, dp, lane, p.n_density, do_scan != 0);
            // mgau_norm (:185-203): own best as the norm, beam cut leaves the tail raw
            const int32_t norm = L.sc[0] >> kSenscrShift;
            cnt = N;
            bool cut = false;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (!cut) {
                    int32_t v = (int32_t)(0u - ((uint32_t)(L.sc[j] >> kSenscrShift) - (uint32_t)norm));
                    if (v > kMaxNegAscr) v = kMaxNegAscr;
                    L.sc[j] = v;
                    if (p.beam[f] && v > p.beam[f]) { cnt = j; cut = true; }
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < N; ++i) { cur_cw[f * N + i] = L.cw[i]; cur_sc[f * N + i] = L.sc[i]; }
                cur_n[f] = cnt;
            }
        }
        else {
#pragma unroll
            for (int i = 0; i < N; ++i) { L.cw[i] = cur_cw[f * N + i]; L.sc[i] = cur_sc[f * N + i]; }
            cnt = cur_n[f];
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) { s_cw[f * N + i] = L.cw[i]; s_sc[f * N + i] = L.sc[i]; }
            s_n[f] = cnt;
        }
    }
    // memset(senone_scores, 0) (:846)
    for (int i = tid; i < p.n_sen; i += kSemiThreads) s_out[i] = 0;
    __syncthreads();

    const bool four = p.mixw_cb != nullptr;
    const int n = compall ? (four ? (p.n_sen & ~1) : p.n_sen) : n_list;
    for (int i = tid; i < n; i += kSemiThreads) {
        const int sen = compall ? i : list[i];
        int32_t acc = 0;
        for (int f = 0; f < p.n_feat; ++f) {
            const int cnt = s_n[f];
            // the unrolled 4-bit kernels (:446-741) keep mixw_cb + score in uint8
            const bool wrap = four && !compall && cnt >= 1 && cnt <= 6;
            int32_t tmp = 0;
            for (int k = 0; k < max(cnt, 1); ++k) {
                const int cw = s_cw[f * N + k];
                int32_t w;
                if (four) {
                    const int b = p.mixw[((size_t)f * p.n_density + cw) * p.row + (sen >> 1)];
                    w = s_cb[(sen & 1) ? (b >> 4) : (b & 0x0f)];
                }
                else
                    w = p.mixw[((size_t)f * p.n_density + cw) * p.row + sen];
                int32_t y = w + s_sc[f * N + k];
                if (wrap) y &= 0xff;
                if (k == 0) tmp = y;
                else {
                    // fast_logmath_add (tied_mgau_common.h:106-125)
                    const int32_t lo_ = min(tmp, y);
                    const uint32_t dd = (uint32_t)(max(tmp, y) - lo_);
                    tmp = lo_ - (dd < (uint32_t)kSemiLa ? (int32_t)s_la[dd] : 0);
                }
            }
            acc = (int32_t)(int16_t)(acc + tmp);          // senone_scores[sen] += tmp, int16
        }
        s_out[sen] = (int16_t)acc;
    }
    __syncthreads();
    for (int i = tid; i < p.n_sen; i += kSemiThreads) out[i] = s_out[i];
    // completion word behind the scores (host-mapped memory), polled by the host
    __threadfence_system();
    __syncthreads();
    if (tid == 0)
        __hip_atomic_store(done_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}


// ---------------------------------------------------------------------------
// batched entry: whole utterances, compallsen.  The top-N of a stream depends on
// the previous frame's list (eval_topn re-scores the carried codewords, eval_cb's
// acceptance threshold starts from them), so the unit of parallelism is the
// (utterance, stream) chain: one wavefront walks its frames in order with exactly
// the per-call procedure (generic_frame_step: codewords on lanes, wave-uniform
// list state), n_utt x n_feat waves in flight.  Lists (normalised, with the
// per-stream beam count) go to HBM; a second kernel, one workgroup per frame,
// scores all senones.  Each utterance starts from the lists of a freshly
// initialised scorer (s2_semi_mgau_init, s2_semi_mgau.c:1305-1322); frames count
// from 0 within the utterance for the down-sampling rule (:173-175).
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256)
void semi_chain_kernel(SemiDev p, const float *__restrict__ feats, int32_t veclen,
                       const int32_t *__restrict__ utt_off, int32_t n_utt,
                       uint8_t *__restrict__ l_cw, int32_t *__restrict__ l_sc, uint8_t *__restrict__ l_n,
                       const uint8_t *__restrict__ seed_in, uint8_t *__restrict__ seed_out, uint8_t *__restrict__ slot_out,
                       int32_t n_hist, const int32_t *__restrict__ frame_base)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wave >= n_utt * p.n_feat) return;
    const int u = wave / p.n_feat, f = wave - u * p.n_feat;
    const int t0 = utt_off[u], T = utt_off[u + 1] - t0;
    const int len = p.featlen[f];
    const float *mean = p.mean + p.foff[f], *var = p.var + p.foff[f];
    const float *det = p.det + (size_t)f * p.n_density;
    TopN<N> L;
    // psgpu_semi_score_batch_carry_dev: the utterance goes on (or a session's next one begins) from carried codeword lists -- the
    // previous frame's, which s2_semi_mgau_frame_eval copies before mgau_dist (s2_semi_mgau.c:853-860); scores are re-derived -- and its
    // frames count from `base` for the down-sampling rule and the history slot
    const size_t so = ((size_t)u * p.n_feat + f) * N;
#pragma unroll
    for (int i = 0; i < N; ++i) { L.cw[i] = seed_in ? (int32_t)seed_in[so + i] : i; L.sc[i] = kMaxNegInt32; }
    const int base = frame_base ? frame_base[u] : 0;
    float dt[kSemiK];
#pragma unroll
    for (int k = 0; k < kSemiK; ++k) dt[k] = det[min(k * 64 + lane, p.n_density - 1)];
    for (int t = 0; t < T; ++t) {
        const float *x = feats + (size_t)(t0 + t) * veclen + p.featoff[f];
        float d[kSemiK], dp[kSemiK];
#pragma unroll
        for (int k = 0; k < kSemiK; ++k) {
            const int cw = min(k * 64 + lane, p.n_density - 1);         // clamp; masked in the scan
            const float *m = mean + (size_t)cw * len, *v = var + (size_t)cw * len;
            float acc = dt[k], prev = acc;
            for (int j = 0; j < len; ++j) {
                prev = acc;
                acc = gau_step(acc, x[j], m[j], v[j]);
            }
            d[k] = acc; dp[k] = prev;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) L.sc[i] = kMaxNegInt32;             // carried codewords, scores re-derived
        generic_frame_step<N, true>(L, d, dp, lane, p.n_density, ((base + t) % p.ds_ratio) == 0);
        if (lane == 0) {
            // what later frames start from: the utterance's last lists (the next call's first frame), and slot n_hist - 1 of the
            // reference's ring (the next utterance's first frame, :855-858) = the lists of the last frame ts with ts % n_hist == n_hist - 1
            if (seed_out && t == T - 1) {
#pragma unroll
                for (int i = 0; i < N; ++i) seed_out[so + i] = (uint8_t)L.cw[i];
            }
            if (slot_out && (base + t) % n_hist == n_hist - 1) {
#pragma unroll
                for (int i = 0; i < N; ++i) slot_out[so + i] = (uint8_t)L.cw[i];
            }
        }
        // mgau_norm (:185-203)
        const int32_t norm = L.sc[0] >> kSenscrShift;
        int cnt = N;
        bool cut = false;
        int32_t v_[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            v_[j] = L.sc[j];
            if (!cut) {
                int32_t v = (int32_t)(0u - ((uint32_t)(L.sc[j] >> kSenscrShift) - (uint32_t)norm));
                if (v > kMaxNegAscr) v = kMaxNegAscr;
                v_[j] = v;
                if (p.beam[f] && v > p.beam[f]) { cnt = j; cut = true; }
            }
        }
        if (lane == 0) {
            const size_t o = ((size_t)(t0 + t) * p.n_feat + f) * N;
#pragma unroll
            for (int i = 0; i < N; ++i) { l_cw[o + i] = (uint8_t)L.cw[i]; l_sc[o + i] = v_[i]; }
            l_n[(size_t)(t0 + t) * p.n_feat + f] = (uint8_t)cnt;
        }
    }
}

template <int N>
__global__ __launch_bounds__(256)
void semi_senone_batch_kernel(SemiDev p, const uint8_t *__restrict__ l_cw, const int32_t *__restrict__ l_sc,
                              const uint8_t *__restrict__ l_n, int16_t *__restrict__ out)
{
    __shared__ uint8_t s_la[kSemiLa];
    __shared__ int32_t s_cw[kSemiMaxFeat * N], s_sc[kSemiMaxFeat * N], s_n[kSemiMaxFeat];
    __shared__ uint8_t s_cb[16];
    const int tid = threadIdx.x, frame = blockIdx.x;
    for (int i = tid; i < kSemiLa; i += 256)
        s_la[i] = (i < p.logadd8_size) ? p.logadd8[i] : 0;
    if (tid < 16) s_cb[tid] = p.mixw_cb ? p.mixw_cb[tid] : 0;
    if (tid < p.n_feat * N) {
        s_cw[tid] = l_cw[(size_t)frame * p.n_feat * N + tid];
        s_sc[tid] = l_sc[(size_t)frame * p.n_feat * N + tid];
    }
    if (tid < p.n_feat) s_n[tid] = l_n[(size_t)frame * p.n_feat + tid];
    __syncthreads();
    const bool four = p.mixw_cb != nullptr;
    const int n = four ? (p.n_sen & ~1) : p.n_sen;                       // compallsen loops (:446-741)
    int16_t *o = out + (size_t)frame * p.n_sen;
    for (int sen = tid; sen < p.n_sen; sen += 256) {
        int32_t acc = 0;
        if (sen < n)
            for (int f = 0; f < p.n_feat; ++f) {
                const int cnt = s_n[f];
                int32_t tmp = 0;
                for (int k = 0; k < max(cnt, 1); ++k) {
                    const int cw = s_cw[f * N + k];
                    int32_t w;
                    if (four) {
                        const int b = p.mixw[((size_t)f * p.n_density + cw) * p.row + (sen >> 1)];
                        w = s_cb[(sen & 1) ? (b >> 4) : (b & 0x0f)];
                    }
                    else
                        w = p.mixw[((size_t)f * p.n_density + cw) * p.row + sen];
                    const int32_t y = w + s_sc[f * N + k];
                    if (k == 0) tmp = y;
                    else {
                        const int32_t lo_ = min(tmp, y);
                        const uint32_t dd = (uint32_t)(max(tmp, y) - lo_);
                        tmp = lo_ - (dd < (uint32_t)kSemiLa ? (int32_t)s_la[dd] : 0);
                    }
                }
                acc = (int32_t)(int16_t)(acc + tmp);
            }
        o[sen] = (int16_t)acc;
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T>
static int up(T **dst, const T *src, size_t n)
{
    PSGPU_HIP(hipMalloc((void **)dst, n * sizeof(T) ? n * sizeof(T) : 1));
    PSGPU_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PSGPU_OK;
}

extern "C" {

int psgpu_semi_model_create(psgpu_semi_model_t **out, int32_t n_feat, int32_t n_density,
                            const int32_t *featlen, int32_t n_sen, int32_t topn, int32_t ds_ratio,
                            const uint8_t *topn_beam,
                            const float *mean, const float *var, const float *det,
                            const uint8_t *mixw, const uint8_t *mixw_cb,
                            const uint8_t *logadd8, int32_t logadd8_size)
{
    PSGPU_REQUIRE(out && featlen && mean && var && det && mixw && logadd8,
                  "psgpu_semi_model_create: NULL argument");
    PSGPU_REQUIRE(n_feat >= 1 && n_feat <= kSemiMaxFeat, "n_feat %d outside 1..%d", n_feat, kSemiMaxFeat);
    PSGPU_REQUIRE(n_density >= 1 && n_density <= 64 * kSemiK, "n_density %d outside 1..%d", n_density, 64 * kSemiK);
    PSGPU_REQUIRE(topn >= 1 && topn <= kSemiMaxTopn && topn <= n_density, "topn %d outside 1..%d", topn, kSemiMaxTopn);
    PSGPU_REQUIRE(ds_ratio >= 1, "ds_ratio %d < 1", ds_ratio);
    PSGPU_REQUIRE(n_sen > 0 && n_sen < 65535, "n_sen %d outside 1..65534", n_sen);
    PSGPU_REQUIRE(logadd8_size >= 256, "log-add table has %d < 256 entries", logadd8_size);
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;
    psgpu_semi_model_t *m = new psgpu_semi_model_t();
    memset(m, 0, sizeof *m);
    SemiDev &d = m->d;
    d.n_feat = n_feat; d.n_density = n_density; d.n_sen = n_sen; d.topn = topn;
    d.ds_ratio = ds_ratio; d.logadd8_size = logadd8_size;
    d.row = mixw_cb ? (n_sen + 1) / 2 : n_sen;
    int32_t o = 0;
    for (int f = 0; f < n_feat; ++f) {
        PSGPU_REQUIRE(featlen[f] >= 1, "stream %d has length %d", f, featlen[f]);
        d.featlen[f] = featlen[f]; d.featoff[f] = m->veclen; m->veclen += featlen[f];
        d.foff[f] = o; o += n_density * featlen[f];
        d.beam[f] = topn_beam ? topn_beam[f] : 0;
    }
    if (m->veclen > kSemiMaxVec) {
        psgpu_set_error("feature vector of %d floats exceeds %d", m->veclen, kSemiMaxVec);
        delete m;
        return PSGPU_EINVAL;
    }
    if ((rc = up(&m->mean, mean, (size_t)o)) || (rc = up(&m->var, var, (size_t)o)) ||
        (rc = up(&m->det, det, (size_t)n_feat * n_density)) ||
        (rc = up(&m->mixw, mixw, (size_t)n_feat * n_density * d.row)) ||
        (rc = up(&m->logadd8, logadd8, (size_t)logadd8_size)) ||
        (mixw_cb && (rc = up(&m->mixw_cb, mixw_cb, (size_t)16)))) {
        psgpu_semi_model_free(m);
        return rc;
    }
    d.mean = m->mean; d.var = m->var; d.det = m->det; d.mixw = m->mixw;
    d.mixw_cb = m->mixw_cb; d.logadd8 = m->logadd8;
    *out = m;
    return PSGPU_OK;
}

void psgpu_semi_model_free(psgpu_semi_model_t *m)
{
    if (!m) return;
    hipFree(m->mean); hipFree(m->var); hipFree(m->det);
    hipFree(m->mixw); hipFree(m->mixw_cb); hipFree(m->logadd8);
    hipFree(m->b_cw); hipFree(m->b_n); hipFree(m->b_sc);
    delete m;
}

int psgpu_semi_state_reset(psgpu_semi_state_t *s)
{
    PSGPU_REQUIRE(s != nullptr, "psgpu_semi_state_reset: NULL state");
    // s2_semi_mgau_init (:1305-1322): codeword k / WORST_DIST, counts 0
    const SemiDev &d = s->m->d;
    const size_t n = (size_t)s->n_hist * d.n_feat * d.topn;
    int32_t *cw = (int32_t *)malloc(n * sizeof(int32_t)), *sc = (int32_t *)malloc(n * sizeof(int32_t));
    if (!cw || !sc) { free(cw); free(sc); psgpu_set_error("out of host memory"); return PSGPU_ENOMEM; }
    for (size_t i = 0; i < n; ++i) { cw[i] = (int32_t)(i % d.topn); sc[i] = kMaxNegInt32; }
    hipError_t e1 = hipMemcpy(s->hist_cw, cw, n * sizeof(int32_t), hipMemcpyHostToDevice);
    hipError_t e2 = hipMemcpy(s->hist_sc, sc, n * sizeof(int32_t), hipMemcpyHostToDevice);
    hipError_t e3 = hipMemset(s->hist_n, 0, (size_t)s->n_hist * d.n_feat * sizeof(int32_t));
    free(cw); free(sc);
    PSGPU_HIP(e1); PSGPU_HIP(e2); PSGPU_HIP(e3);
    s->cur = 0;
    return PSGPU_OK;
}

int psgpu_semi_state_create(psgpu_semi_state_t **out, psgpu_semi_model_t *m, int32_t n_topn_hist)
{
    PSGPU_REQUIRE(out && m, "psgpu_semi_state_create: NULL argument");
    PSGPU_REQUIRE(n_topn_hist >= 1 && n_topn_hist <= 64, "n_topn_hist %d outside 1..64", n_topn_hist);
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;
    psgpu_semi_state_t *s = new psgpu_semi_state_t();
    memset(s, 0, sizeof *s);
    s->m = m; s->n_hist = n_topn_hist;
    const SemiDev &d = m->d;
    const size_t n = (size_t)n_topn_hist * d.n_feat * d.topn;
    hipError_t e = hipMalloc((void **)&s->hist_cw, n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&s->hist_sc, n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&s->hist_n, (size_t)n_topn_hist * d.n_feat * sizeof(int32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_list, (size_t)d.n_sen * sizeof(uint16_t), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_out, ((size_t)d.n_sen * sizeof(int16_t) + 15) / 16 * 16 + 16, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&s->d_list, s->h_list, 0);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&s->d_out, s->h_out, 0);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        psgpu_set_error("psgpu_semi_state_create: %s", hipGetErrorString(e));
        psgpu_semi_state_free(s);
        return e == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP;
    }
    rc = psgpu_semi_state_reset(s);
    if (rc != PSGPU_OK) { psgpu_semi_state_free(s); return rc; }
    *out = s;
    return PSGPU_OK;
}

void psgpu_semi_state_free(psgpu_semi_state_t *s)
{
    if (!s) return;
    if (s->stream) hipStreamDestroy(s->stream);
    hipFree(s->hist_cw); hipFree(s->hist_sc); hipFree(s->hist_n);
    if (s->h_list) hipHostFree(s->h_list);
    if (s->h_out) hipHostFree(s->h_out);
    delete s;
}

int psgpu_semi_frame_eval(psgpu_semi_state_t *s, int16_t *senscr,
                          const uint8_t *senone_active, int32_t n_senone_active,
                          const float *feat, int32_t frame, int32_t frame_idx, int32_t compallsen)
{
    PSGPU_REQUIRE(s && senscr && feat, "psgpu_semi_frame_eval: NULL argument");
    PSGPU_REQUIRE(frame >= 0, "negative frame %d", frame);
    PSGPU_REQUIRE(compallsen || n_senone_active == 0 || senone_active, "active list missing");
    psgpu_semi_model_t *m = s->m;
    const SemiDev &d = m->d;
    const int slot = frame % s->n_hist;                        // s2_semi_mgau.c:849-850
    const int prev = (slot == 0) ? s->n_hist - 1 : slot - 1;
    const size_t sl = (size_t)d.n_feat * d.topn;
    const int fresh = frame >= frame_idx;                      // :853
    s->cur = slot;
    int n_list = 0;
    if (!compallsen) {
        int sen = 0;
        for (int i = 0; i < n_senone_active; ++i) {
            sen += senone_active[i];
            if (sen >= d.n_sen) {
                psgpu_set_error("active list runs past n_sen (%d >= %d)", sen, d.n_sen);
                return PSGPU_EINVAL;
            }
            s->h_list[n_list++] = (uint16_t)sen;
        }
    }
    SemiFeat fa;
    memset(&fa, 0, sizeof fa);
    memcpy(fa.x, feat, (size_t)m->veclen * sizeof(float));
    const size_t smem = (((size_t)d.n_sen * 2 + 15) / 16) * 16;
    const size_t done_off = ((size_t)d.n_sen * sizeof(int16_t) + 15) / 16 * 16;
    volatile uint32_t *h_done = reinterpret_cast<volatile uint32_t *>(reinterpret_cast<char *>(s->h_out) + done_off);
    uint32_t *d_done = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s->d_out) + done_off);
    const uint32_t seq = ++s->seq ? s->seq : ++s->seq;
#define PSGPU_SEMI_CASE(NN) case NN: hipLaunchKernelGGL((semi_frame_kernel<NN>), dim3(1), dim3(kSemiThreads), smem, s->stream, \
        d, fa, (int32_t)fresh, (int32_t)(frame % d.ds_ratio == 0), (int32_t)(compallsen != 0), (int32_t)n_list,          \
        (const uint16_t *)s->d_list, (const int32_t *)(s->hist_cw + prev * sl), s->hist_cw + slot * sl,                  \
        s->hist_sc + slot * sl, s->hist_n + (size_t)slot * d.n_feat, s->d_out, d_done, seq); break;
    switch (d.topn) {
        PSGPU_SEMI_CASE(1) PSGPU_SEMI_CASE(2) PSGPU_SEMI_CASE(3) PSGPU_SEMI_CASE(4)
        PSGPU_SEMI_CASE(5) PSGPU_SEMI_CASE(6) PSGPU_SEMI_CASE(7) default: PSGPU_SEMI_CASE(8)
    }
#undef PSGPU_SEMI_CASE
    PSGPU_HIP(hipGetLastError());
    {
        bool done = false;
        for (long i = 0; i < 200000000L; ++i) {
            if (*h_done == seq) { done = true; break; }
            __builtin_ia32_pause();
        }
        if (!done) PSGPU_HIP(hipStreamSynchronize(s->stream));
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    memcpy(senscr, s->h_out, (size_t)d.n_sen * sizeof(int16_t));
    return PSGPU_OK;
}

int32_t psgpu_semi_n_sen(const psgpu_semi_model_t *m) { return m ? m->d.n_sen : 0; }
int32_t psgpu_semi_veclen(const psgpu_semi_model_t *m) { return m ? m->veclen : 0; }

int32_t psgpu_semi_n_feat(const psgpu_semi_model_t *m) { return m ? m->d.n_feat : 0; }
int32_t psgpu_semi_topn(const psgpu_semi_model_t *m) { return m ? m->d.topn : 0; }

int psgpu_semi_score_batch_dev(psgpu_semi_model_t *m, const float *feats_dev, const int32_t *utt_off_dev,
                               int32_t n_utt, int32_t total_frames, int16_t *senscr_dev, void *stream)
{
    return psgpu_semi_score_batch_carry_dev(m, feats_dev, utt_off_dev, n_utt, total_frames, nullptr, nullptr, nullptr, 1, nullptr, senscr_dev, stream);
}

int psgpu_semi_score_batch_carry_dev(psgpu_semi_model_t *m, const float *feats_dev, const int32_t *utt_off_dev, int32_t n_utt,
                                     int32_t total_frames, const uint8_t *seed_in_dev, uint8_t *seed_out_dev, uint8_t *slot_out_dev,
                                     int32_t n_hist, const int32_t *frame_base_dev, int16_t *senscr_dev, void *stream)
{
    PSGPU_REQUIRE(m && n_utt >= 0 && total_frames >= 0 && n_hist >= 1, "psgpu_semi_score_batch_dev: bad argument");
    PSGPU_REQUIRE(!seed_in_dev || (seed_in_dev != seed_out_dev && seed_in_dev != slot_out_dev),
                  "psgpu_semi_score_batch_carry_dev: the lists carried in must not be the buffer of a carry-out");
    if (n_utt == 0 || total_frames == 0) return PSGPU_OK;
    PSGPU_REQUIRE(feats_dev && utt_off_dev && senscr_dev, "psgpu_semi_score_batch_dev: NULL device buffer");
    const SemiDev &d = m->d;
    hipStream_t st = (hipStream_t)stream;
    if (total_frames > m->b_cap) {
        PSGPU_HIP(hipFree(m->b_cw)); PSGPU_HIP(hipFree(m->b_sc)); PSGPU_HIP(hipFree(m->b_n));
        m->b_cw = nullptr; m->b_sc = nullptr; m->b_n = nullptr; m->b_cap = 0;
        const size_t ne = (size_t)total_frames * d.n_feat * d.topn;
        PSGPU_HIP(hipMalloc((void **)&m->b_cw, ne));
        PSGPU_HIP(hipMalloc((void **)&m->b_sc, ne * sizeof(int32_t)));
        PSGPU_HIP(hipMalloc((void **)&m->b_n, (size_t)total_frames * d.n_feat));
        m->b_cap = total_frames;
    }
    const int waves = n_utt * d.n_feat;
#define PSGPU_SEMI_B(NN) case NN:                                                                                     \
        hipLaunchKernelGGL((semi_chain_kernel<NN>), dim3((waves + 3) / 4), dim3(256), 0, st, d, feats_dev, m->veclen, \
                           utt_off_dev, n_utt, m->b_cw, m->b_sc, m->b_n, seed_in_dev, seed_out_dev, slot_out_dev, n_hist,     \
                           frame_base_dev);                                                                           \
        hipLaunchKernelGGL((semi_senone_batch_kernel<NN>), dim3(total_frames), dim3(256), 0, st, d,                   \
                           (const uint8_t *)m->b_cw, (const int32_t *)m->b_sc, (const uint8_t *)m->b_n, senscr_dev);  \
        break;
    switch (d.topn) {
        PSGPU_SEMI_B(1) PSGPU_SEMI_B(2) PSGPU_SEMI_B(3) PSGPU_SEMI_B(4)
        PSGPU_SEMI_B(5) PSGPU_SEMI_B(6) PSGPU_SEMI_B(7) default: PSGPU_SEMI_B(8)
    }
#undef PSGPU_SEMI_B
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

int psgpu_semi_score_batch(psgpu_semi_model_t *m, const float *feats, const int32_t *utt_off, int32_t n_utt,
                           int16_t *senscr)
{
    PSGPU_REQUIRE(m && utt_off && n_utt >= 0, "psgpu_semi_score_batch: bad argument");
    if (n_utt == 0) return PSGPU_OK;
    const int32_t T = utt_off[n_utt];
    PSGPU_REQUIRE(utt_off[0] == 0 && T >= 0, "utt_off must start at 0");
    for (int u = 0; u < n_utt; ++u) PSGPU_REQUIRE(utt_off[u + 1] >= utt_off[u], "utt_off must be non-decreasing");
    if (T == 0) return PSGPU_OK;
    PSGPU_REQUIRE(feats && senscr, "psgpu_semi_score_batch: NULL buffer");
    const SemiDev &d = m->d;
    float *df = nullptr; int32_t *doff = nullptr; int16_t *ds = nullptr;
    auto cleanup = [&]() { hipFree(df); hipFree(doff); hipFree(ds); };
#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) {                 \
        psgpu_set_error("%s -> %s", #call, hipGetErrorString(e_)); cleanup();          \
        return e_ == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP; } } while (0)
    TRY(hipMalloc((void **)&df, sizeof(float) * (size_t)T * m->veclen));
    TRY(hipMalloc((void **)&doff, sizeof(int32_t) * ((size_t)n_utt + 1)));
    TRY(hipMalloc((void **)&ds, sizeof(int16_t) * (size_t)T * d.n_sen));
    TRY(hipMemcpy(df, feats, sizeof(float) * (size_t)T * m->veclen, hipMemcpyHostToDevice));
    TRY(hipMemcpy(doff, utt_off, sizeof(int32_t) * ((size_t)n_utt + 1), hipMemcpyHostToDevice));
    int rc = psgpu_semi_score_batch_dev(m, df, doff, n_utt, T, ds, nullptr);
    if (rc == PSGPU_OK) {
        TRY(hipDeviceSynchronize());
        TRY(hipMemcpy(senscr, ds, sizeof(int16_t) * (size_t)T * d.n_sen, hipMemcpyDeviceToHost));
    }
#undef TRY
    cleanup();
    return rc;
}

int psgpu_semi_state_get_topn(psgpu_semi_state_t *s, int32_t slot, int32_t *cw, int32_t *score, int32_t *n_used)
{
    PSGPU_REQUIRE(s != nullptr, "psgpu_semi_state_get_topn: NULL state");
    if (slot < 0) slot = s->cur;
    PSGPU_REQUIRE(slot < s->n_hist, "slot %d outside the %d-slot ring", slot, s->n_hist);
    const SemiDev &d = s->m->d;
    const size_t sl = (size_t)d.n_feat * d.topn;
    PSGPU_HIP(hipStreamSynchronize(s->stream));
    if (cw) PSGPU_HIP(hipMemcpy(cw, s->hist_cw + slot * sl, sl * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (score) PSGPU_HIP(hipMemcpy(score, s->hist_sc + slot * sl, sl * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n_used) PSGPU_HIP(hipMemcpy(n_used, s->hist_n + (size_t)slot * d.n_feat, d.n_feat * sizeof(int32_t), hipMemcpyDeviceToHost));
    return PSGPU_OK;
}

int psgpu_semi_state_set_topn(psgpu_semi_state_t *s, int32_t slot, const int32_t *cw, const int32_t *score,
                              const int32_t *n_used)
{
    PSGPU_REQUIRE(s && cw && score && n_used, "psgpu_semi_state_set_topn: NULL argument");
    PSGPU_REQUIRE(slot >= 0 && slot < s->n_hist, "slot %d outside the %d-slot ring", slot, s->n_hist);
    const SemiDev &d = s->m->d;
    const size_t sl = (size_t)d.n_feat * d.topn;
    for (size_t i = 0; i < sl; ++i)
        PSGPU_REQUIRE(cw[i] >= 0 && cw[i] < d.n_density, "codeword %d outside the codebook", cw[i]);
    PSGPU_HIP(hipStreamSynchronize(s->stream));
    PSGPU_HIP(hipMemcpy(s->hist_cw + slot * sl, cw, sl * sizeof(int32_t), hipMemcpyHostToDevice));
    PSGPU_HIP(hipMemcpy(s->hist_sc + slot * sl, score, sl * sizeof(int32_t), hipMemcpyHostToDevice));
    PSGPU_HIP(hipMemcpy(s->hist_n + (size_t)slot * d.n_feat, n_used, d.n_feat * sizeof(int32_t), hipMemcpyHostToDevice));
    return PSGPU_OK;
}

}  // extern "C"
