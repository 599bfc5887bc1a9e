// psgpu_ptm.hip -- phonetically-tied-mixture senone scoring on gfx950.
//
// Replaces, bit-exactly, the per-frame work of ptm_mgau_frame_eval()
// (reference src/ptm_mgau.c:408-454) for whole batches of utterances:
//
//   kernel 1  ptm_chain_kernel     eval_topn + eval_cb (ptm_mgau.c:87-226)
//       One 64-lane wavefront owns one (utterance, codebook, stream) "chain"
//       and marches over the utterance's frames.  The chain's 128 Gaussians
//       live in VGPRs for the whole utterance (2 codewords per lane), the
//       frame's 13-float stream vector arrives through the scalar cache, the
//       128 fp32 distances are computed with the reference's exact
//       sub/mul/mul/sub order (no FMA contraction: SURVEY F5), and the
//       history-dependent top-N update (seed re-score, threshold scan in
//       codeword order, insert-ahead-of-equals, skip-if-present) is emulated
//       with wave ballots on wave-uniform (scalar) list state.
//
//   kernel 2  ptm_senone_kernel    ptm_mgau_codebook_norm + ptm_mgau_senone_eval
//       (ptm_mgau.c:265-295, :326-403).  One workgroup per frame: normalise
//       the 126 top-N lists in LDS, then each lane gathers the uint8 mixture
//       weights of its senones (coalesced along the senone axis), log-adds
//       them through the 256-entry table in LDS, and the block min-reduces
//       and stores int16 scores.
//
// The model (3.7 MB) stays resident in L2 / Infinity Cache; HBM traffic is the
// feature rows in and the int16 score rows out (+ the top-N lists between the
// two kernels).  See DESIGN.md for the roofline discussion.
#include "psgpu_internal.h"
#include <vector>

struct psgpu_ptm_model_s {
    int32_t n_mgau, n_feat, n_density, n_sen, topn, ds_ratio, veclen, n_chain;
    int32_t featlen[16];
    int32_t featoff[16];
    int32_t uniform_len;          // featlen if all streams are equal, else 0
    int device;
    float *mean, *var, *det;      // device
    int64_t *cboff;               // device: float offset of (mgau, feat) block
    uint8_t *mixw, *sen2cb, *logadd8;
    int32_t logadd8_size;
};

struct PtmDev {
    const float *mean, *var, *det;
    const uint8_t *mixw, *sen2cb, *logadd8;
    int32_t n_mgau, n_feat, n_density, n_sen, veclen, n_chain, ds_ratio, logadd8_size;
};

static PtmDev dev_view(const psgpu_ptm_model_t *m)
{
    PtmDev p;
    p.mean = m->mean; p.var = m->var; p.det = m->det;
    p.mixw = m->mixw; p.sen2cb = m->sen2cb; p.logadd8 = m->logadd8;
    p.n_mgau = m->n_mgau; p.n_feat = m->n_feat; p.n_density = m->n_density;
    p.n_sen = m->n_sen; p.veclen = m->veclen; p.n_chain = m->n_chain;
    p.ds_ratio = m->ds_ratio; p.logadd8_size = m->logadd8_size;
    return p;
}

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------

// float -> int32 exactly as the reference does (ptm_mgau.c:129-132, :220-223)
__device__ __forceinline__ int32_t dist_to_int(float d)
{
    return (d < (float)kMaxNegInt32) ? kMaxNegInt32 : (int32_t)d;
}

// one dimension of the Gaussian distance, rounded after every operation
// (ptm_mgau.c:64-69 COMPUTE_GMM_MAP / COMPUTE_GMM_REDUCE)
__device__ __forceinline__ float gau_step(float d, float x, float m, float v)
{
    float diff = __fsub_rn(x, m);
    float sq = __fmul_rn(diff, diff);
    float c = __fmul_rn(sq, v);
    return __fsub_rn(d, c);
}

__device__ __forceinline__ float lane_value(float v, int lane)
{
    return __builtin_bit_cast(float,
        __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// max over the 64 lanes of a wavefront, returned wave-uniform.  Quad
// butterflies and row rotations run on the DPP path (no LDS traffic); the four
// 16-lane rows are combined on the scalar unit.
__device__ __forceinline__ int32_t wave_max_i32(int32_t v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));  // row_ror:4
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));  // row_ror:8
    const int32_t r0 = __builtin_amdgcn_readlane(v, 0);
    const int32_t r1 = __builtin_amdgcn_readlane(v, 16);
    const int32_t r2 = __builtin_amdgcn_readlane(v, 32);
    const int32_t r3 = __builtin_amdgcn_readlane(v, 48);
    return max(max(r0, r1), max(r2, r3));
}

// Wave-uniform top-N list.
template <int N>
struct TopN {
    int32_t cw[N];
    int32_t sc[N];
};

// ---------------------------------------------------------------------------
// kernel 1: top-N chains, specialised for 128 densities (2 per lane) and a
// compile-time stream length.
// ---------------------------------------------------------------------------
template <int LEN, int N>
__global__ __launch_bounds__(256)
void ptm_chain_kernel(PtmDev p, const float *__restrict__ feats,
                      const int32_t *__restrict__ utt_off, int32_t n_utt,
                      uint8_t *__restrict__ seed_cw,
                      int32_t *__restrict__ topn_score, uint8_t *__restrict__ topn_cw)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(
        (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int n_chain = p.n_chain;
    if (wave >= n_utt * n_chain)
        return;
    const int utt = wave / n_chain;
    const int chain = wave - utt * n_chain;
    const int f = chain % p.n_feat;
    const int t0 = utt_off[utt];
    const int T = utt_off[utt + 1] - t0;

    // Gaussian parameters of codewords `lane` and `lane + 64`
    float m0[LEN], v0[LEN], m1[LEN], v1[LEN];
    {
        const float *mp = p.mean + ((size_t)chain * 128 + lane) * LEN;
        const float *vp = p.var + ((size_t)chain * 128 + lane) * LEN;
#pragma unroll
        for (int j = 0; j < LEN; ++j) {
            m0[j] = mp[j];
            v0[j] = vp[j];
            m1[j] = mp[64 * LEN + j];
            v1[j] = vp[64 * LEN + j];
        }
    }
    const float det0 = p.det[(size_t)chain * 128 + lane];
    const float det1 = p.det[(size_t)chain * 128 + lane + 64];

    // seed list (ptm_mgau.c:790-793 for a fresh decoder)
    TopN<N> L;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        L.cw[i] = seed_cw ? (int32_t)seed_cw[((size_t)utt * n_chain + chain) * N + i] : i;
        L.sc[i] = kMaxNegInt32;
    }

    const float *xrow = feats + (size_t)t0 * p.veclen + f * LEN;
    float xn[LEN];
    if (T > 0) {
#pragma unroll
        for (int j = 0; j < LEN; ++j) xn[j] = xrow[j];
    }

    for (int t = 0; t < T; ++t) {
        float x[LEN];
#pragma unroll
        for (int j = 0; j < LEN; ++j) x[j] = xn[j];
        if (t + 1 < T) {                       // prefetch the next frame's vector
            const float *nx = xrow + (size_t)(t + 1) * p.veclen;
#pragma unroll
            for (int j = 0; j < LEN; ++j) xn[j] = nx[j];
        }

        // all 128 distances, reference operation order
        float d0 = det0, d1 = det1;
#pragma unroll
        for (int j = 0; j < LEN; ++j) {
            d0 = gau_step(d0, x[j], m0[j], v0[j]);
            d1 = gau_step(d1, x[j], m1[j], v1[j]);
        }

        const bool scan_frame = (p.ds_ratio == 1) || ((t % p.ds_ratio) == 0);

        // ---- fast path.  If the four largest truncated scores of the
        // codebook are pairwise distinct and strictly above the fifth, the
        // reference's seed/scan procedure ends with exactly those four in
        // descending order whatever the seeds were ("top-N
        // closed form"), so they are extracted with four wave-wide
        // unique-maximum rounds.  Any tie falls through to the exact
        // emulation below.
        bool exact = !scan_frame;
        if (scan_frame) {
            int32_t s0 = dist_to_int(d0), s1 = dist_to_int(d1);
            TopN<N> F;
#pragma unroll
            for (int r = 0; r < N; ++r) {
                const int32_t mx = wave_max_i32(max(s0, s1));
                const bool e0 = (s0 == mx), e1 = (s1 == mx);
                const unsigned long long b0 = __ballot(e0), b1 = __ballot(e1);
                if (__popcll(b0) + __popcll(b1) != 1)
                    exact = true;
                F.sc[r] = mx;
                F.cw[r] = b0 ? (__ffsll((long long)b0) - 1) : (64 + __ffsll((long long)b1) - 1);
                s0 = e0 ? kMaxNegInt32 : s0;
                s1 = e1 ? kMaxNegInt32 : s1;
            }
            if (!exact) L = F;
        }

        if (exact) {
        // ---- eval_topn: re-score the carried codewords, stable insertion
        // sort, descending, strict '>' (ptm_mgau.c:71-85, :87-136)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = L.cw[i];
            const float d = (c < 64) ? lane_value(d0, c) : lane_value(d1, c - 64);
            L.sc[i] = dist_to_int(d);
#pragma unroll
            for (int j = i; j > 0; --j) {
                if (L.sc[j] > L.sc[j - 1]) {
                    int32_t ts = L.sc[j]; L.sc[j] = L.sc[j - 1]; L.sc[j - 1] = ts;
                    int32_t tc = L.cw[j]; L.cw[j] = L.cw[j - 1]; L.cw[j - 1] = tc;
                }
            }
        }

        // ---- eval_cb: scan codewords in index order against the moving
        // threshold (ptm_mgau.c:151-226).  Only frames that are multiples of
        // the downsampling ratio are scanned (:242).
        if (scan_frame) {
            int pos = 0;                        // next codeword index to look at
            for (;;) {
                const float th = (float)L.sc[N - 1];
                bool in0 = false, in1 = false;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    in0 |= (L.cw[i] == lane);
                    in1 |= (L.cw[i] == lane + 64);
                }
                unsigned long long b0 = __ballot(d0 >= th && !in0);
                unsigned long long b1 = __ballot(d1 >= th && !in1);
                if (pos >= 64) {
                    b0 = 0;
                    b1 = (pos >= 128) ? 0ull : (b1 & (~0ull << (pos - 64)));
                }
                else
                    b0 &= (~0ull << pos);
                if ((b0 | b1) == 0)
                    break;
                const int c = b0 ? (__ffsll((long long)b0) - 1)
                                 : (64 + __ffsll((long long)b1) - 1);
                const float d = (c < 64) ? lane_value(d0, c) : lane_value(d1, c - 64);
                const int32_t s = dist_to_int(d);
                // insertion_sort_cb (:140-149): ahead of equal scores, worst drops
                int q = N - 1;
#pragma unroll
                for (int k = N - 1; k > 0; --k) {
                    if (q == k && s >= L.sc[k - 1]) {
                        L.sc[k] = L.sc[k - 1];
                        L.cw[k] = L.cw[k - 1];
                        q = k - 1;
                    }
                }
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    if (q == k) { L.sc[k] = s; L.cw[k] = c; }
                }
                pos = c + 1;
            }
        }
        }   // exact

        // ---- publish the raw list of this frame
        if (lane == 0) {
            const size_t o = ((size_t)(t0 + t) * n_chain + chain) * N;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                topn_score[o + i] = L.sc[i];
                topn_cw[o + i] = (uint8_t)L.cw[i];
            }
        }
    }

    if (seed_cw && lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i)
            seed_cw[((size_t)utt * n_chain + chain) * N + i] = (uint8_t)L.cw[i];
    }
}

// ---------------------------------------------------------------------------
// kernel 2: normalise + mixture-weight log-sum, one workgroup per frame
// ---------------------------------------------------------------------------
constexpr int kSenThreads = 256;

template <int N>
__global__ __launch_bounds__(kSenThreads)
void ptm_senone_kernel(PtmDev p, const int32_t *__restrict__ topn_score,
                       const uint8_t *__restrict__ topn_cw,
                       int16_t *__restrict__ senscr, int32_t *__restrict__ best_out,
                       uint32_t flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: [n_sen] int16 scores | [n_chain*N] cw | [n_chain*N] norm score |
    //         [256] log-add table | [16] int norm | [8] int red
    int16_t *s_out = reinterpret_cast<int16_t *>(smem);
    const int out_bytes = ((p.n_sen * 2 + 15) / 16) * 16;
    const int list_bytes = ((p.n_chain * N + 15) / 16) * 16;
    uint8_t *s_cw = smem + out_bytes;
    uint8_t *s_sc = s_cw + list_bytes;
    uint8_t *s_la = s_sc + list_bytes;
    int32_t *s_norm = reinterpret_cast<int32_t *>(s_la + 256);
    int32_t *s_red = s_norm + 16;

    const int tid = threadIdx.x;
    const int frame = blockIdx.x;
    const int n_ent = p.n_chain * N;
    const size_t lo = (size_t)frame * n_ent;

    if (tid < p.n_feat) s_norm[tid] = kWorstScore;
    if (tid < 8) s_red[tid] = 0x7fffffff;
    for (int i = tid; i < 256; i += kSenThreads)
        s_la[i] = (i < p.logadd8_size) ? p.logadd8[i] : 0;
    __syncthreads();

    // ptm_mgau_codebook_norm (:265-295): norm[f] = max over codebooks of
    // (best score >> 10)
    for (int i = tid; i < p.n_chain; i += kSenThreads) {
        const int f = i % p.n_feat;
        atomicMax(&s_norm[f], topn_score[lo + (size_t)i * N] >> kSenscrShift);
    }
    __syncthreads();
    for (int i = tid; i < n_ent; i += kSenThreads) {
        const int f = (i / N) % p.n_feat;
        int32_t v = topn_score[lo + i] >> kSenscrShift;
        v = -(v - s_norm[f]);
        if (v > kMaxNegAscr) v = kMaxNegAscr;
        s_sc[i] = (uint8_t)v;
        s_cw[i] = topn_cw[lo + i];
    }
    __syncthreads();

    // ptm_mgau_senone_eval (:326-403)
    int32_t mybest = 0x7fffffff;
    for (int s = tid; s < p.n_sen; s += kSenThreads) {
        const int cb = p.sen2cb[s];
        int32_t ascore = 0;
        for (int f = 0; f < p.n_feat; ++f) {
            const int li = (cb * p.n_feat + f) * N;
            const uint8_t *wrow = p.mixw + (size_t)f * p.n_density * p.n_sen + s;
            int32_t fden = 0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int32_t y = (int32_t)wrow[(size_t)s_cw[li + k] * p.n_sen] + s_sc[li + k];
                if (k == 0)
                    fden = y;
                else {
                    // fast_logmath_add (tied_mgau_common.h:106-125)
                    const int32_t lo_ = min(fden, y);
                    const int32_t d = max(fden, y) - lo_;
                    fden = lo_ - (d < 256 ? (int32_t)s_la[d] : 0);
                }
            }
            ascore += fden;
        }
        s_out[s] = (int16_t)ascore;
        mybest = min(mybest, ascore);
    }
    // block minimum
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        mybest = min(mybest, __shfl_xor(mybest, off));
    if ((tid & 63) == 0) atomicMin(&s_red[0], mybest);
    __syncthreads();
    const int32_t best = s_red[0];
    if (best_out && tid == 0) best_out[frame] = best;
    const int32_t sub = (flags & PSGPU_PTM_RAW_SCORES) ? 0 : best;
    int16_t *orow = senscr + (size_t)frame * p.n_sen;
    for (int s = tid; s < p.n_sen; s += kSenThreads)
        orow[s] = (int16_t)(s_out[s] - sub);    // int16 store as in :398-400
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T>
static int upload(T **dst, const T *src, size_t n)
{
    PSGPU_HIP(hipMalloc((void **)dst, n * sizeof(T) ? n * sizeof(T) : 1));
    PSGPU_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PSGPU_OK;
}

extern "C" {

int psgpu_ptm_model_create(psgpu_ptm_model_t **out,
                           int32_t n_mgau, int32_t n_feat, int32_t n_density,
                           const int32_t *featlen, int32_t n_sen, int32_t topn,
                           int32_t ds_ratio,
                           const float *mean, const float *var, const float *det,
                           const uint8_t *mixw, const uint8_t *sen2cb,
                           const uint8_t *logadd8, int32_t logadd8_size)
{
    PSGPU_REQUIRE(out && featlen && mean && var && det && mixw && sen2cb && logadd8,
                  "psgpu_ptm_model_create: NULL argument");
    PSGPU_REQUIRE(n_mgau > 0 && n_mgau <= 256, "n_mgau %d outside 1..256 (ptm_mgau.c:838)", n_mgau);
    PSGPU_REQUIRE(n_feat > 0 && n_feat <= 16, "n_feat %d outside 1..16", n_feat);
    PSGPU_REQUIRE(n_density == 128, "n_density %d: this build handles 128-density PTM codebooks", n_density);
    PSGPU_REQUIRE(topn == 4, "topn %d: this build handles topn = 4", topn);
    PSGPU_REQUIRE(ds_ratio >= 1, "ds_ratio %d < 1", ds_ratio);
    PSGPU_REQUIRE(n_sen > 0 && n_sen < 32768, "n_sen %d outside 1..32767", n_sen);
    PSGPU_REQUIRE(logadd8_size >= 256, "log-add table has %d < 256 entries (logmath.c:112)", logadd8_size);
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;

    psgpu_ptm_model_t *m = new psgpu_ptm_model_t();
    m->n_mgau = n_mgau; m->n_feat = n_feat; m->n_density = n_density;
    m->n_sen = n_sen; m->topn = topn; m->ds_ratio = ds_ratio;
    m->n_chain = n_mgau * n_feat;
    m->veclen = 0; m->uniform_len = featlen[0];
    for (int f = 0; f < n_feat; ++f) {
        m->featlen[f] = featlen[f];
        m->featoff[f] = m->veclen;
        m->veclen += featlen[f];
        if (featlen[f] != featlen[0]) m->uniform_len = 0;
    }
    if (m->uniform_len != 13) {
        psgpu_set_error("stream lengths must all be 13 in this build (got %d...)", featlen[0]);
        delete m;
        return PSGPU_EINVAL;
    }
    hipGetDevice(&m->device);
    const size_t npar = (size_t)n_mgau * n_density * m->veclen;
    if ((rc = upload(&m->mean, mean, npar)) || (rc = upload(&m->var, var, npar)) ||
        (rc = upload(&m->det, det, (size_t)m->n_chain * n_density)) ||
        (rc = upload(&m->mixw, mixw, (size_t)n_feat * n_density * n_sen)) ||
        (rc = upload(&m->sen2cb, sen2cb, (size_t)n_sen)) ||
        (rc = upload(&m->logadd8, logadd8, (size_t)logadd8_size))) {
        psgpu_ptm_model_free(m);
        return rc;
    }
    m->logadd8_size = logadd8_size;
    *out = m;
    return PSGPU_OK;
}

void psgpu_ptm_model_free(psgpu_ptm_model_t *m)
{
    if (!m) return;
    hipFree(m->mean); hipFree(m->var); hipFree(m->det);
    hipFree(m->mixw); hipFree(m->sen2cb); hipFree(m->logadd8);
    delete m;
}

int32_t psgpu_ptm_n_sen(const psgpu_ptm_model_t *m) { return m->n_sen; }
int32_t psgpu_ptm_n_chain(const psgpu_ptm_model_t *m) { return m->n_chain; }
int32_t psgpu_ptm_veclen(const psgpu_ptm_model_t *m) { return m->veclen; }
int32_t psgpu_ptm_topn(const psgpu_ptm_model_t *m) { return m->topn; }

int psgpu_ptm_topn_dev(psgpu_ptm_model_t *m, const float *feats_dev,
                       const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                       uint8_t *seed_cw_dev, int32_t *topn_score_dev,
                       uint8_t *topn_cw_dev, void *stream)
{
    PSGPU_REQUIRE(m && feats_dev && utt_off_dev && topn_score_dev && topn_cw_dev,
                  "psgpu_ptm_topn_dev: NULL argument");
    PSGPU_REQUIRE(n_utt >= 0 && total_frames >= 0, "negative sizes");
    if (n_utt == 0 || total_frames == 0) return PSGPU_OK;
    const long long waves = (long long)n_utt * m->n_chain;
    const int blocks = (int)((waves + 3) / 4);
    hipLaunchKernelGGL((ptm_chain_kernel<13, 4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       dev_view(m), feats_dev, utt_off_dev, n_utt, seed_cw_dev,
                       topn_score_dev, topn_cw_dev);
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

int psgpu_ptm_senone_dev(psgpu_ptm_model_t *m, int32_t total_frames,
                         const int32_t *topn_score_dev, const uint8_t *topn_cw_dev,
                         int16_t *senscr_dev, int32_t *best_dev, uint32_t flags,
                         void *stream)
{
    PSGPU_REQUIRE(m && topn_score_dev && topn_cw_dev && senscr_dev,
                  "psgpu_ptm_senone_dev: NULL argument");
    if (total_frames <= 0) return PSGPU_OK;
    const int out_bytes = ((m->n_sen * 2 + 15) / 16) * 16;
    const int list_bytes = ((m->n_chain * 4 + 15) / 16) * 16;
    const size_t smem = (size_t)out_bytes + 2 * list_bytes + 256 + 16 * 4 + 8 * 4;
    hipLaunchKernelGGL((ptm_senone_kernel<4>), dim3(total_frames), dim3(kSenThreads), smem,
                       (hipStream_t)stream, dev_view(m), topn_score_dev, topn_cw_dev,
                       senscr_dev, best_dev, flags);
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

int psgpu_ptm_score_batch_dev(psgpu_ptm_model_t *m,
                              const float *feats_dev, const int32_t *utt_off_dev,
                              int32_t n_utt, int32_t total_frames,
                              uint8_t *seed_cw_dev,
                              int32_t *topn_score_dev, uint8_t *topn_cw_dev,
                              int16_t *senscr_dev, int32_t *best_dev,
                              uint32_t flags, void *stream)
{
    int rc = psgpu_ptm_topn_dev(m, feats_dev, utt_off_dev, n_utt, total_frames, seed_cw_dev,
                                topn_score_dev, topn_cw_dev, stream);
    if (rc != PSGPU_OK || senscr_dev == nullptr) return rc;
    return psgpu_ptm_senone_dev(m, total_frames, topn_score_dev, topn_cw_dev, senscr_dev,
                                best_dev, flags, stream);
}

int psgpu_ptm_score_batch(psgpu_ptm_model_t *m,
                          const float *feats, const int32_t *utt_off, int32_t n_utt,
                          uint8_t *seed_cw,
                          int32_t *topn_score, uint8_t *topn_cw,
                          int16_t *senscr, int32_t *best, uint32_t flags)
{
    PSGPU_REQUIRE(m && feats && utt_off && n_utt >= 0, "psgpu_ptm_score_batch: bad argument");
    if (n_utt == 0) return PSGPU_OK;
    const int32_t T = utt_off[n_utt];
    PSGPU_REQUIRE(T >= 0 && utt_off[0] == 0, "utt_off must start at 0 and be non-decreasing");
    for (int u = 0; u < n_utt; ++u)
        PSGPU_REQUIRE(utt_off[u + 1] >= utt_off[u], "utt_off must be non-decreasing");
    if (T == 0) return PSGPU_OK;
    const size_t n_ent = (size_t)T * m->n_chain * m->topn;
    float *d_feat = nullptr; int32_t *d_off = nullptr, *d_sc = nullptr, *d_best = nullptr;
    uint8_t *d_seed = nullptr, *d_cw = nullptr; int16_t *d_scr = nullptr;
    int rc = PSGPU_OK;
    auto cleanup = [&]() {
        hipFree(d_feat); hipFree(d_off); hipFree(d_sc); hipFree(d_best);
        hipFree(d_seed); hipFree(d_cw); hipFree(d_scr);
    };
#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) {                 \
        psgpu_set_error("%s -> %s", #call, hipGetErrorString(e_)); cleanup();          \
        return e_ == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP; } } while (0)
    TRY(hipMalloc((void **)&d_feat, (size_t)T * m->veclen * sizeof(float)));
    TRY(hipMalloc((void **)&d_off, (size_t)(n_utt + 1) * sizeof(int32_t)));
    TRY(hipMalloc((void **)&d_sc, n_ent * sizeof(int32_t)));
    TRY(hipMalloc((void **)&d_cw, n_ent));
    TRY(hipMemcpy(d_feat, feats, (size_t)T * m->veclen * sizeof(float), hipMemcpyHostToDevice));
    TRY(hipMemcpy(d_off, utt_off, (size_t)(n_utt + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    if (seed_cw) {
        TRY(hipMalloc((void **)&d_seed, (size_t)n_utt * m->n_chain * m->topn));
        TRY(hipMemcpy(d_seed, seed_cw, (size_t)n_utt * m->n_chain * m->topn, hipMemcpyHostToDevice));
    }
    if (senscr) TRY(hipMalloc((void **)&d_scr, (size_t)T * m->n_sen * sizeof(int16_t)));
    if (best) TRY(hipMalloc((void **)&d_best, (size_t)T * sizeof(int32_t)));
    rc = psgpu_ptm_score_batch_dev(m, d_feat, d_off, n_utt, T, d_seed, d_sc, d_cw, d_scr, d_best,
                                   flags, nullptr);
    if (rc == PSGPU_OK) {
        TRY(hipDeviceSynchronize());
        if (topn_score) TRY(hipMemcpy(topn_score, d_sc, n_ent * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (topn_cw) TRY(hipMemcpy(topn_cw, d_cw, n_ent, hipMemcpyDeviceToHost));
        if (senscr) TRY(hipMemcpy(senscr, d_scr, (size_t)T * m->n_sen * sizeof(int16_t), hipMemcpyDeviceToHost));
        if (best) TRY(hipMemcpy(best, d_best, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (seed_cw) TRY(hipMemcpy(seed_cw, d_seed, (size_t)n_utt * m->n_chain * m->topn, hipMemcpyDeviceToHost));
    }
#undef TRY
    cleanup();
    return rc;
}

}  // extern "C"
