// psgpu_ptm_frame.hip -- the stateful, per-call PTM scorer: a literal device
// replacement of ptm_mgau_frame_eval() (reference src/ptm_mgau.c:408-454)
// including everything the batched path may ignore: the history ring
// hist[n_fast_hist] (:884-890), the `frame >= frame_idx` rule (:430), the
// active-codebook subset derived from the active senone list (:297-321),
// seeds-only re-scoring of inactive codebooks (:237-239 vs :246-251), the
// in-place normalisation of active codebooks only (:265-295), the persistent
// overwrite with MAX_NEG_ASCR (:353-364) and the best-score subtraction over
// ALL n_sen entries (:398-400).  This is what the ps_mgau_t vtable shim
// (integration/psgpu_mgau_shim.c) calls once per acmod_score().
//
// All state lives in HBM; a call is 1-2 kernel launches on the state's own
// stream, the scores leave through a host-mapped (pinned) buffer.
#include "psgpu_ptm_dev.h"
#include <cstring>
#include <cstdlib>

constexpr int kMaxVec = 64;            // feature vector travels as a kernel argument
struct FeatArg { float x[kMaxVec]; };
struct MaskArg { uint32_t w[8]; };     // active codebooks, n_mgau <= 256 (ptm_mgau.c:838)

struct psgpu_ptm_state_s {
    psgpu_ptm_model_t *m;
    int32_t n_hist;
    int32_t *hist_cw;        // [n_hist][n_chain][topn]
    int32_t *hist_sc;        // [n_hist][n_chain][topn]  raw or normalised, as the reference's slot
    uint32_t *hist_active;   // [n_hist][8]
    uint16_t *h_list;        // pinned + mapped: absolute senone ids of the active list
    uint16_t *d_list;        // device alias of h_list
    int16_t *h_out;          // pinned + mapped: n_sen scores, then a completion word
    int16_t *d_out;
    uint32_t seq;            // call counter: the kernel stores it behind the scores when done
    hipStream_t stream;
    int32_t cur;             // slot of the most recent call (s->f)
    // ---- look-ahead cache (psgpu_ptm_state_lookahead): the frames the caller
    // announced, scored in ONE batched pass when the first of them is asked for
    struct {
        int valid, computed, flushed;
        int c0, cn, cap;         // announced frames c0 .. c0+cn-1
        int next_fresh;          // next frame the cache may serve as a fresh evaluation
        int max_fresh;           // last frame served fresh from the cache (c0-1: none)
        int limit;               // frames >= limit are no longer covered (history diverged)
        float *h_feat, *d_feat;  // [cn][veclen] (pinned / device)
        int32_t *d_off, *d_sc, *d_best, *h_best;
        uint8_t *d_cw, *d_seed;
        int16_t *d_raw, *h_raw;  // [cn][n_sen] un-normalised scores (pinned host copy)
    } la;
    long la_served, la_batches;
};

// look-ahead: seed lists of the batched pass = codewords of a ring slot
__global__ void la_seed_kernel(const int32_t *__restrict__ slot_cw, uint8_t *__restrict__ seed, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) seed[i] = (uint8_t)slot_cw[i];
}

// look-ahead: write the lists of cache frame t (chain-major batch buffers) into
// ring slot t % n_hist exactly as a fresh all-codebooks-active evaluation would
// leave them: codewords, scores normalised by ptm_mgau_codebook_norm
// (ptm_mgau.c:265-295), every codebook active.  One workgroup per frame.
__global__ __launch_bounds__(256)
void la_flush_kernel(PtmDev p, const int32_t *__restrict__ sc, const uint32_t *__restrict__ cw,
                     int32_t cn, int32_t c0, int32_t t_first, int32_t n_hist,
                     int32_t *__restrict__ hist_cw, int32_t *__restrict__ hist_sc,
                     uint32_t *__restrict__ hist_active)
{
    __shared__ int32_t s_norm[16];
    const int t = t_first + blockIdx.x;
    const int slot = t % n_hist;
    const int rel = t - c0;
    if (threadIdx.x < 16) s_norm[threadIdx.x] = kWorstScore;
    __syncthreads();
    for (int c = threadIdx.x; c < p.n_chain; c += blockDim.x)
        atomicMax(&s_norm[c % p.n_feat], sc[((size_t)c * cn + rel) * 4] >> kSenscrShift);
    __syncthreads();
    for (int c = threadIdx.x; c < p.n_chain; c += blockDim.x) {
        const int4 v = *reinterpret_cast<const int4 *>(sc + ((size_t)c * cn + rel) * 4);
        const uint32_t w = cw[(size_t)c * cn + rel];
        const int32_t norm = s_norm[c % p.n_feat];
        const int32_t raw[4] = {v.x, v.y, v.z, v.w};
        int32_t *ocw = hist_cw + ((size_t)slot * p.n_chain + c) * 4;
        int32_t *osc = hist_sc + ((size_t)slot * p.n_chain + c) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int32_t x = (int32_t)(0u - ((uint32_t)(raw[k] >> kSenscrShift) - (uint32_t)norm));
            if (x > kMaxNegAscr) x = kMaxNegAscr;
            ocw[k] = (int32_t)((w >> (8 * k)) & 0xff);
            osc[k] = x;
        }
    }
    if (threadIdx.x < 8) hist_active[(size_t)slot * 8 + threadIdx.x] = 0xffffffffu;
}

// ---------------------------------------------------------------------------
// kernel A: one wavefront per (codebook, stream) chain, one frame.
//   copy the previous slot's codewords (ptm_mgau.c:435-441), eval_topn for
//   every chain (:237-239), eval_cb for chains of active codebooks on scan
//   frames (:242-251).  Raw lists go to the current slot.
// ---------------------------------------------------------------------------
template <int LEN>
__global__ __launch_bounds__(256)
void ptm_frame_topn_kernel(PtmDev p, FeatArg fa, MaskArg mask, int32_t do_scan,
                           const int32_t *__restrict__ prev_cw,
                           int32_t *__restrict__ cur_cw, int32_t *__restrict__ cur_sc,
                           uint32_t *__restrict__ cur_active)
{
    constexpr int N = 4;
    const int lane = threadIdx.x & 63;
    const int chain = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (blockIdx.x == 0 && threadIdx.x < 8)
        cur_active[threadIdx.x] = mask.w[threadIdx.x];
    if (chain >= p.n_chain)
        return;
    const int cb = chain / p.n_feat;
    const int f = chain - cb * p.n_feat;
    const bool active = (mask.w[cb >> 5] >> (cb & 31)) & 1u;

    const float *mp = p.mean + ((size_t)chain * 128 + lane) * LEN;
    const float *vp = p.var + ((size_t)chain * 128 + lane) * LEN;
    float d0 = p.det[(size_t)chain * 128 + lane];
    float d1 = p.det[(size_t)chain * 128 + lane + 64];
#pragma unroll
    for (int j = 0; j < LEN; ++j) {
        const float xj = fa.x[f * LEN + j];
        d0 = gau_step(d0, xj, mp[j], vp[j]);
        d1 = gau_step(d1, xj, mp[64 * LEN + j], vp[64 * LEN + j]);
    }
    TopN<N> L;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        L.cw[i] = __builtin_amdgcn_readfirstlane(prev_cw[(size_t)chain * N + i]);
        L.sc[i] = kMaxNegInt32;
    }
    const bool scan = active && do_scan;
    if (!(scan && closed_form_top4(L, d0, d1, 127 - lane)))
        exact_frame_step<N>(L, d0, d1, lane, scan);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            cur_cw[(size_t)chain * N + i] = L.cw[i];
            cur_sc[(size_t)chain * N + i] = L.sc[i];
        }
    }
}

// kernel A, any shape: n_density <= 256 (4 codewords per lane), any stream
// lengths, top-N 1..8 -- the exact sequential procedure only (no closed form).
template <int N>
__global__ __launch_bounds__(256)
void ptm_frame_topn_generic(PtmDev p, FeatArg fa, MaskArg mask, int32_t do_scan,
                            const int32_t *__restrict__ prev_cw,
                            int32_t *__restrict__ cur_cw, int32_t *__restrict__ cur_sc,
                            uint32_t *__restrict__ cur_active)
{
    const int lane = threadIdx.x & 63;
    const int chain = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (blockIdx.x == 0 && threadIdx.x < 8)
        cur_active[threadIdx.x] = mask.w[threadIdx.x];
    if (chain >= p.n_chain)
        return;
    const int cb = chain / p.n_feat;
    const int f = chain - cb * p.n_feat;
    const bool active = (mask.w[cb >> 5] >> (cb & 31)) & 1u;
    const int len = p.featlen[f];
    // packed [mgau][feat][density][featlen[f]] (ms_gauden.c:211-221)
    const size_t base = (size_t)cb * p.n_density * p.veclen + (size_t)p.n_density * p.featoff[f];
    const float *mean = p.mean + base, *var = p.var + base;
    const float *det = p.det + (size_t)chain * p.n_density;
    const float *x = fa.x + p.featoff[f];
    float d[kGenK];
#pragma unroll
    for (int k = 0; k < kGenK; ++k) {
        const int cw = min(k * 64 + lane, p.n_density - 1);
        const float *m = mean + (size_t)cw * len, *v = var + (size_t)cw * len;
        float acc = det[cw];
        for (int j = 0; j < len; ++j)
            acc = gau_step(acc, x[j], m[j], v[j]);
        d[k] = acc;
    }
    TopN<N> L;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        L.cw[i] = __builtin_amdgcn_readfirstlane(prev_cw[(size_t)chain * N + i]);
        L.sc[i] = kMaxNegInt32;
    }
    generic_frame_step<N, false>(L, d, d, lane, p.n_density, active && do_scan);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            cur_cw[(size_t)chain * N + i] = L.cw[i];
            cur_sc[(size_t)chain * N + i] = L.sc[i];
        }
    }
}

// ---------------------------------------------------------------------------
// kernel B: one workgroup.  ptm_mgau_codebook_norm (only when the slot was
// just evaluated) + ptm_mgau_senone_eval over the listed senones.
// ---------------------------------------------------------------------------
constexpr int kFrameThreads = 1024;
constexpr int kFrameLa = 512;

template <int N>
__global__ __launch_bounds__(kFrameThreads)
void ptm_frame_senone_kernel(PtmDev p, int32_t fresh, int32_t compall, int32_t n_list,
                             const uint16_t *__restrict__ list,
                             const int32_t *__restrict__ cur_cw, int32_t *__restrict__ cur_sc,
                             const uint32_t *__restrict__ cur_active,
                             int16_t *__restrict__ out, uint32_t *__restrict__ done_word, uint32_t seq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int16_t *s_out = reinterpret_cast<int16_t *>(smem);                    // [n_sen]
    const int out_bytes = ((p.n_sen * 2 + 15) / 16) * 16;
    int32_t *s_sc = reinterpret_cast<int32_t *>(smem + out_bytes);         // [n_chain * N]
    uint8_t *s_cw = reinterpret_cast<uint8_t *>(s_sc + p.n_chain * N);     // [n_chain * N]
    __shared__ uint8_t s_la[kFrameLa];
    __shared__ int32_t s_norm[16];
    __shared__ uint32_t s_active[8], s_touch[8];
    __shared__ int32_t s_best;

    const int tid = threadIdx.x;
    const int n_ent = p.n_chain * N;
    if (tid < 16) s_norm[tid] = kWorstScore;
    if (tid < 8) { s_active[tid] = cur_active[tid]; s_touch[tid] = 0; }
    if (tid == 0) s_best = 0x7fffffff;
    for (int i = tid; i < kFrameLa; i += kFrameThreads)
        s_la[i] = (i < p.logadd8_size) ? p.logadd8[i] : 0;
    for (int i = tid; i < n_ent; i += kFrameThreads) {
        s_sc[i] = cur_sc[i];
        s_cw[i] = (uint8_t)cur_cw[i];
    }
    if (compall)
        for (int i = tid; i < p.n_sen; i += kFrameThreads)
            s_out[i] = 0;                                  // memset (:333); list mode: the host fills the rest
    __syncthreads();

    if (fresh) {
        // ptm_mgau_codebook_norm (:265-295), active codebooks only
        for (int c = tid; c < p.n_chain; c += kFrameThreads) {
            const int cb = c / p.n_feat, f = c - cb * p.n_feat;
            if ((s_active[cb >> 5] >> (cb & 31)) & 1u)
                atomicMax(&s_norm[f], s_sc[c * N] >> kSenscrShift);
        }
        __syncthreads();
        for (int i = tid; i < n_ent; i += kFrameThreads) {
            const int c = i / N;
            const int cb = c / p.n_feat, f = c - cb * p.n_feat;
            if ((s_active[cb >> 5] >> (cb & 31)) & 1u) {
                // int arithmetic wraps exactly as the reference's
                int32_t v = (int32_t)((uint32_t)(s_sc[i] >> kSenscrShift) - (uint32_t)s_norm[f]);
                v = (int32_t)(0u - (uint32_t)v);
                if (v > kMaxNegAscr) v = kMaxNegAscr;
                s_sc[i] = v;
            }
        }
        __syncthreads();
    }

    // listed senones of inactive codebooks force that codebook's scores to
    // MAX_NEG_ASCR, persistently in the slot (:353-364)
    const int n = compall ? p.n_sen : n_list;
    for (int i = tid; i < n; i += kFrameThreads) {
        const int sen = compall ? i : list[i];
        const int cb = p.sen2cb[sen];
        if (!((s_active[cb >> 5] >> (cb & 31)) & 1u))
            atomicOr(&s_touch[cb >> 5], 1u << (cb & 31));
    }
    __syncthreads();
    for (int i = tid; i < n_ent; i += kFrameThreads) {
        const int cb = (i / N) / p.n_feat;
        if ((s_touch[cb >> 5] >> (cb & 31)) & 1u)
            s_sc[i] = kMaxNegAscr;
    }
    __syncthreads();
    // the slot keeps what the reference would leave in it
    for (int i = tid; i < n_ent; i += kFrameThreads)
        cur_sc[i] = s_sc[i];

    // ptm_mgau_senone_eval (:326-403)
    int32_t mybest = 0x7fffffff;
    for (int i = tid; i < n; i += kFrameThreads) {
        const int sen = compall ? i : list[i];
        const int cb = p.sen2cb[sen];
        int32_t ascore = 0;
        for (int f = 0; f < p.n_feat; ++f) {
            const int li = (cb * p.n_feat + f) * N;
            const uint8_t *wrow = p.mixw + (size_t)f * p.n_density * p.n_sen + sen;
            int32_t w[N];
#pragma unroll
            for (int k = 0; k < N; ++k)
                w[k] = wrow[(size_t)s_cw[li + k] * p.n_sen];
            int32_t fden = w[0] + s_sc[li];
#pragma unroll
            for (int k = 1; k < N; ++k) {
                // fast_logmath_add (tied_mgau_common.h:106-125)
                const int32_t y = w[k] + s_sc[li + k];
                const int32_t lo_ = min(fden, y);
                const uint32_t d = (uint32_t)(max(fden, y) - lo_);
                fden = lo_ - (d < (uint32_t)kFrameLa ? (int32_t)s_la[d] : 0);
            }
            ascore += fden;
        }
        s_out[compall ? sen : i] = (int16_t)ascore;       // list mode: list order, compact
        mybest = min(mybest, ascore);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        mybest = min(mybest, __shfl_xor(mybest, off));
    if ((tid & 63) == 0) atomicMin(&s_best, mybest);
    __syncthreads();
    const uint32_t best = (uint32_t)s_best;
    // compallsen: the whole row; list mode: the n listed scores in list order -- every other
    // entry of senone_scores is 0 - best (:398-400), which the host fills in from done_word[1]
    const int n_out = compall ? p.n_sen : n;
    for (int i = tid; i < n_out; i += kFrameThreads)
        out[i] = (int16_t)(uint16_t)((uint32_t)(int32_t)s_out[i] - best);   // int16 -= int (:398-400)
    if (tid == 0) done_word[1] = best;
    // completion word behind the scores (host-mapped memory): the host polls it
    // instead of paying for a stream synchronisation per call
    __threadfence_system();
    __syncthreads();
    if (tid == 0)
        __hip_atomic_store(done_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static void la_release(psgpu_ptm_state_t *s)
{
    if (s->la.h_feat) hipHostFree(s->la.h_feat);
    if (s->la.h_raw) hipHostFree(s->la.h_raw);
    if (s->la.h_best) hipHostFree(s->la.h_best);
    hipFree(s->la.d_feat); hipFree(s->la.d_off); hipFree(s->la.d_sc); hipFree(s->la.d_cw);
    hipFree(s->la.d_seed); hipFree(s->la.d_raw); hipFree(s->la.d_best);
    memset(&s->la, 0, sizeof s->la);
}

// Bring the ring up to date with the frames the cache served as fresh
// evaluations (the reference writes slot t % H on every fresh call).
static int la_flush(psgpu_ptm_state_t *s)
{
    if (!s->la.valid || s->la.flushed || s->la.max_fresh < s->la.c0)
        return PSGPU_OK;
    const int t_last = s->la.max_fresh;
    int t_first = t_last - s->n_hist + 1;
    if (t_first < s->la.c0) t_first = s->la.c0;
    hipLaunchKernelGGL(la_flush_kernel, dim3(t_last - t_first + 1), dim3(256), 0, s->stream, dev_view(s->m),
                       (const int32_t *)s->la.d_sc, reinterpret_cast<const uint32_t *>(s->la.d_cw),
                       s->la.cn, s->la.c0, t_first, s->n_hist, s->hist_cw, s->hist_sc, s->hist_active);
    PSGPU_HIP(hipGetLastError());
    s->la.flushed = 1;
    return PSGPU_OK;
}

// One batched pass over the announced frames, seeded with the ring slot of frame c0 - 1.
static int la_compute(psgpu_ptm_state_t *s)
{
    psgpu_ptm_model_t *m = s->m;
    const int cn = s->la.cn, prev = (s->la.c0 % s->n_hist == 0) ? s->n_hist - 1 : (s->la.c0 % s->n_hist) - 1;
    const size_t slot_len = (size_t)m->n_chain * m->topn;
    const int32_t off[2] = {0, cn};
    PSGPU_HIP(hipMemcpyAsync(s->la.d_feat, s->la.h_feat, (size_t)cn * m->veclen * sizeof(float), hipMemcpyHostToDevice, s->stream));
    PSGPU_HIP(hipMemcpyAsync(s->la.d_off, off, sizeof off, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(la_seed_kernel, dim3((unsigned)((slot_len + 255) / 256)), dim3(256), 0, s->stream,
                       (const int32_t *)(s->hist_cw + prev * slot_len), s->la.d_seed, (int)slot_len);
    PSGPU_HIP(hipGetLastError());
    int rc = psgpu_ptm_score_batch_dev(m, s->la.d_feat, s->la.d_off, 1, cn, s->la.d_seed, nullptr, s->la.d_sc,
                                       s->la.d_cw, s->la.d_raw, s->la.d_best, PSGPU_PTM_RAW_SCORES, s->stream);
    if (rc != PSGPU_OK) return rc;
    PSGPU_HIP(hipMemcpyAsync(s->la.h_raw, s->la.d_raw, (size_t)cn * m->n_sen * sizeof(int16_t), hipMemcpyDeviceToHost, s->stream));
    PSGPU_HIP(hipMemcpyAsync(s->la.h_best, s->la.d_best, (size_t)cn * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
    PSGPU_HIP(hipStreamSynchronize(s->stream));
    s->la.computed = 1;
    ++s->la_batches;
    return PSGPU_OK;
}

// Scores of one call from the cached un-normalised row: listed senones get
// raw - min over the list, every other entry -min (ptm_mgau.c:393-400).
static void la_serve(psgpu_ptm_state_t *s, int frame, int16_t *senscr, int compallsen, int n_list)
{
    const psgpu_ptm_model_t *m = s->m;
    const int16_t *raw = s->la.h_raw + (size_t)(frame - s->la.c0) * m->n_sen;
    if (compallsen) {
        const uint32_t best = (uint32_t)s->la.h_best[frame - s->la.c0];
        for (int i = 0; i < m->n_sen; ++i)
            senscr[i] = (int16_t)(uint16_t)((uint32_t)(int32_t)raw[i] - best);
    }
    else {
        int32_t bs = 0x7fffffff;
        for (int i = 0; i < n_list; ++i) {
            const int32_t v = raw[s->h_list[i]];
            if (v < bs) bs = v;
        }
        const uint32_t best = (uint32_t)bs;
        const int16_t rest = (int16_t)(uint16_t)(0u - best);
        for (int i = 0; i < m->n_sen; ++i) senscr[i] = rest;
        for (int i = 0; i < n_list; ++i) {
            const int sen = s->h_list[i];
            senscr[sen] = (int16_t)(uint16_t)((uint32_t)(int32_t)raw[sen] - best);
        }
    }
    ++s->la_served;
}

extern "C" {

int psgpu_ptm_state_reset(psgpu_ptm_state_t *s)
{
    PSGPU_REQUIRE(s != nullptr, "psgpu_ptm_state_reset: NULL state");
    // ptm_mgau_reset_fast_hist (ptm_mgau.c:777-802): cw = 0..N-1,
    // score = WORST_DIST, every codebook active, in every slot
    const psgpu_ptm_model_t *m = s->m;
    const size_t n = (size_t)s->n_hist * m->n_chain * m->topn;
    int32_t *cw = (int32_t *)malloc(n * sizeof(int32_t));
    int32_t *sc = (int32_t *)malloc(n * sizeof(int32_t));
    if (!cw || !sc) { free(cw); free(sc); psgpu_set_error("out of host memory"); return PSGPU_ENOMEM; }
    for (size_t i = 0; i < n; ++i) { cw[i] = (int32_t)(i % m->topn); sc[i] = kMaxNegInt32; }
    uint32_t act[64 * 8];
    memset(act, 0xff, sizeof act);
    hipError_t e1 = hipMemcpy(s->hist_cw, cw, n * sizeof(int32_t), hipMemcpyHostToDevice);
    hipError_t e2 = hipMemcpy(s->hist_sc, sc, n * sizeof(int32_t), hipMemcpyHostToDevice);
    hipError_t e3 = hipMemcpy(s->hist_active, act, (size_t)s->n_hist * 8 * sizeof(uint32_t), hipMemcpyHostToDevice);
    free(cw); free(sc);
    PSGPU_HIP(e1); PSGPU_HIP(e2); PSGPU_HIP(e3);
    s->cur = 0;
    s->la.valid = 0;
    return PSGPU_OK;
}

int psgpu_ptm_state_create(psgpu_ptm_state_t **out, psgpu_ptm_model_t *m, int32_t n_fast_hist)
{
    PSGPU_REQUIRE(out && m, "psgpu_ptm_state_create: NULL argument");
    PSGPU_REQUIRE(n_fast_hist >= 1 && n_fast_hist <= 64, "n_fast_hist %d outside 1..64", n_fast_hist);
    PSGPU_REQUIRE(m->veclen <= kMaxVec, "feature vector of %d floats exceeds %d", m->veclen, kMaxVec);
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;
    psgpu_ptm_state_t *s = new psgpu_ptm_state_t();
    memset(s, 0, sizeof *s);
    s->m = m; s->n_hist = n_fast_hist;
    const size_t n = (size_t)n_fast_hist * m->n_chain * m->topn;
    hipError_t e = hipMalloc((void **)&s->hist_cw, n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&s->hist_sc, n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&s->hist_active, (size_t)n_fast_hist * 8 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_list, (size_t)m->n_sen * sizeof(uint16_t), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_out, ((size_t)m->n_sen * sizeof(int16_t) + 15) / 16 * 16 + 16, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&s->d_list, s->h_list, 0);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&s->d_out, s->h_out, 0);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        psgpu_set_error("psgpu_ptm_state_create: %s", hipGetErrorString(e));
        psgpu_ptm_state_free(s);
        return e == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP;
    }
    rc = psgpu_ptm_state_reset(s);
    if (rc != PSGPU_OK) { psgpu_ptm_state_free(s); return rc; }
    *out = s;
    return PSGPU_OK;
}

void psgpu_ptm_state_free(psgpu_ptm_state_t *s)
{
    if (!s) return;
    if (s->stream) hipStreamDestroy(s->stream);
    hipFree(s->hist_cw); hipFree(s->hist_sc); hipFree(s->hist_active);
    if (s->h_list) hipHostFree(s->h_list);
    if (s->h_out) hipHostFree(s->h_out);
    la_release(s);
    delete s;
}

int psgpu_ptm_state_lookahead(psgpu_ptm_state_t *s, const float *feats, int32_t frame0, int32_t n_frames)
{
    PSGPU_REQUIRE(s && (feats || n_frames == 0) && frame0 >= 0 && n_frames >= 0,
                  "psgpu_ptm_state_lookahead: bad argument");
    psgpu_ptm_model_t *m = s->m;
    int rc = la_flush(s);                // the ring must be exact before the cache is replaced
    if (rc != PSGPU_OK) return rc;
    s->la.valid = 0;
    // the batched kernels are specialised (en-us shape) and count down-sampling from the
    // first frame of the pass, the reference from absolute frame numbers (ptm_mgau.c:242)
    if (n_frames == 0 || !m->fast_shape || frame0 % m->ds_ratio != 0)
        return PSGPU_OK;
    if (n_frames > s->la.cap) {
        la_release(s);
        const size_t n = (size_t)n_frames, ne = n * m->n_chain * m->topn;
        hipError_t e = hipHostMalloc((void **)&s->la.h_feat, n * m->veclen * sizeof(float), hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s->la.d_feat, n * m->veclen * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&s->la.d_off, 2 * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&s->la.d_sc, ne * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&s->la.d_cw, ne);
        if (e == hipSuccess) e = hipMalloc((void **)&s->la.d_seed, (size_t)m->n_chain * m->topn);
        if (e == hipSuccess) e = hipMalloc((void **)&s->la.d_raw, n * m->n_sen * sizeof(int16_t));
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->la.h_raw, n * m->n_sen * sizeof(int16_t), hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s->la.d_best, n * sizeof(int32_t));
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->la.h_best, n * sizeof(int32_t), hipHostMallocDefault);
        if (e != hipSuccess) {
            la_release(s);
            psgpu_set_error("psgpu_ptm_state_lookahead: %s", hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP;
        }
        s->la.cap = n_frames;
    }
    memcpy(s->la.h_feat, feats, (size_t)n_frames * m->veclen * sizeof(float));
    s->la.valid = 1; s->la.computed = 0; s->la.flushed = 1;
    s->la.c0 = frame0; s->la.cn = n_frames;
    s->la.next_fresh = frame0; s->la.max_fresh = frame0 - 1; s->la.limit = frame0 + n_frames;
    return PSGPU_OK;
}

int psgpu_ptm_state_lookahead_rows(psgpu_ptm_state_t *s, const int16_t **raw_dev, const int32_t **best_dev,
                                   int32_t *frame0, int32_t *n_frames)
{
    PSGPU_REQUIRE(s && raw_dev && best_dev && frame0 && n_frames, "psgpu_ptm_state_lookahead_rows: NULL argument");
    if (!s->la.valid || s->la.next_fresh != s->la.c0) {
        psgpu_set_error("no look-ahead cache at its first frame");
        return PSGPU_ESTATE;
    }
    if (!s->la.computed) {
        const int rc = la_compute(s);
        if (rc != PSGPU_OK) return rc;
    }
    *raw_dev = s->la.d_raw; *best_dev = s->la.d_best; *frame0 = s->la.c0; *n_frames = s->la.cn;
    return PSGPU_OK;
}

int psgpu_ptm_state_mark_fresh(psgpu_ptm_state_t *s, int32_t frame)
{
    PSGPU_REQUIRE(s != nullptr, "psgpu_ptm_state_mark_fresh: NULL state");
    if (!(s->la.valid && s->la.computed && frame >= s->la.c0 && frame < s->la.limit && frame == s->la.next_fresh)) {
        psgpu_set_error("frame %d is not the next fresh frame of a computed look-ahead cache", frame);
        return PSGPU_ESTATE;
    }
    s->cur = frame % s->n_hist;
    s->la.max_fresh = frame; s->la.next_fresh = frame + 1; s->la.flushed = 0;
    return PSGPU_OK;
}

int psgpu_ptm_state_lookahead_stats(psgpu_ptm_state_t *s, int64_t *served, int64_t *batches)
{
    PSGPU_REQUIRE(s != nullptr, "psgpu_ptm_state_lookahead_stats: NULL state");
    if (served) *served = s->la_served;
    if (batches) *batches = s->la_batches;
    return PSGPU_OK;
}

int psgpu_ptm_frame_eval(psgpu_ptm_state_t *s, int16_t *senscr,
                         const uint8_t *senone_active, int32_t n_senone_active,
                         const float *feat, int32_t frame, int32_t frame_idx,
                         int32_t compallsen)
{
    PSGPU_REQUIRE(s && senscr && feat, "psgpu_ptm_frame_eval: NULL argument");
    PSGPU_REQUIRE(frame >= 0, "negative frame %d", frame);
    PSGPU_REQUIRE(compallsen || n_senone_active == 0 || senone_active,
                  "active list missing (compallsen is off)");
    psgpu_ptm_model_t *m = s->m;
    const PtmDev pv = dev_view(m);
    const int slot = frame % s->n_hist;                      // ptm_mgau.c:425-426
    const size_t slot_len = (size_t)m->n_chain * m->topn;
    int32_t *cur_cw = s->hist_cw + slot * slot_len;
    int32_t *cur_sc = s->hist_sc + slot * slot_len;
    uint32_t *cur_act = s->hist_active + (size_t)slot * 8;
    const int fresh = frame >= frame_idx;                    // :430
    s->cur = slot;

    // active list: uint8 deltas -> absolute ids (the list acmod_flags2list
    // built, acmod.c:1223-1275), and the codebooks it touches (:297-321)
    MaskArg mask;
    memset(&mask, 0, sizeof mask);
    int n_list = 0;
    if (compallsen) {
        memset(&mask, 0xff, sizeof mask);
    }
    else {
        int sen = 0;
        for (int i = 0; i < n_senone_active; ++i) {
            sen += senone_active[i];
            if (sen >= m->n_sen) {
                psgpu_set_error("active list runs past n_sen (%d >= %d)", sen, m->n_sen);
                return PSGPU_EINVAL;
            }
            s->h_list[n_list++] = (uint16_t)sen;
        }
    }
    if (s->la.valid) {
        // Look-ahead cache.  A frame may be served from it when it is one of the
        // announced frames, its feature vector is the announced one, and -- for
        // a fresh evaluation -- every codebook is active and frames arrive in
        // order: then the slot the reference would compute is the all-active
        // list of the batched pass, whose history is the same sequential one.
        const int rel = frame - s->la.c0;
        const bool in_range = rel >= 0 && frame < s->la.limit &&
            memcmp(feat, s->la.h_feat + (size_t)rel * m->veclen, (size_t)m->veclen * sizeof(float)) == 0;
        bool all_cb = compallsen != 0;
        if (!all_cb && in_range && fresh) {
            uint32_t seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int cnt = 0;
            for (int i = 0; i < n_list && cnt < m->n_mgau; ++i) {
                const int cb = m->h_sen2cb[s->h_list[i]];
                if (!((seen[cb >> 5] >> (cb & 31)) & 1u)) { seen[cb >> 5] |= 1u << (cb & 31); ++cnt; }
            }
            all_cb = (cnt == m->n_mgau);
        }
        if (in_range && fresh && all_cb && frame == s->la.next_fresh) {
            if (!s->la.computed) {
                const int rc = la_compute(s);
                if (rc != PSGPU_OK) return rc;
            }
            la_serve(s, frame, senscr, compallsen, n_list);
            s->la.max_fresh = frame; s->la.next_fresh = frame + 1; s->la.flushed = 0;
            return PSGPU_OK;
        }
        if (in_range && !fresh && frame <= s->la.max_fresh && frame > s->la.max_fresh - s->n_hist) {
            la_serve(s, frame, senscr, compallsen, n_list);     // slot reuse: any list
            return PSGPU_OK;
        }
        // per-call from here: the ring must hold what the cache stood for
        const int rc = la_flush(s);
        if (rc != PSGPU_OK) return rc;
        if (fresh) {                      // the history diverges from the batched pass at this frame
            if (frame < s->la.limit) s->la.limit = frame;
            s->la.next_fresh = 0x7fffffff;
        }
    }
    if (fresh) {
        if (!compallsen) {
            // sen2cb lookups happen on the device copy; the host only needs the
            // codebook set, which the model keeps a host mirror of
            for (int i = 0; i < n_list; ++i) {
                const int cb = m->h_sen2cb[s->h_list[i]];
                mask.w[cb >> 5] |= 1u << (cb & 31);
            }
        }
        const int prev = (slot == 0) ? s->n_hist - 1 : slot - 1;
        FeatArg fa;
        memset(&fa, 0, sizeof fa);
        memcpy(fa.x, feat, (size_t)m->veclen * sizeof(float));
        const int blocks = (m->n_chain + 3) / 4;
        const int32_t do_scan = (int32_t)(frame % m->ds_ratio == 0);
        const int32_t *prev_cw = s->hist_cw + prev * slot_len;
        if (m->fast_shape)
            hipLaunchKernelGGL((ptm_frame_topn_kernel<13>), dim3(blocks), dim3(256), 0, s->stream,
                               pv, fa, mask, do_scan, prev_cw, cur_cw, cur_sc, cur_act);
        else {
#define PSGPU_GEN_CASE(NN) case NN: hipLaunchKernelGGL((ptm_frame_topn_generic<NN>), dim3(blocks), dim3(256), 0, \
                s->stream, pv, fa, mask, do_scan, prev_cw, cur_cw, cur_sc, cur_act); break;
            switch (m->topn) {
                PSGPU_GEN_CASE(1) PSGPU_GEN_CASE(2) PSGPU_GEN_CASE(3) PSGPU_GEN_CASE(4)
                PSGPU_GEN_CASE(5) PSGPU_GEN_CASE(6) PSGPU_GEN_CASE(7) default: PSGPU_GEN_CASE(8)
            }
#undef PSGPU_GEN_CASE
        }
        PSGPU_HIP(hipGetLastError());
    }
    const size_t smem = (((size_t)m->n_sen * 2 + 15) / 16) * 16 + slot_len * 5;
    const size_t done_off = ((size_t)m->n_sen * sizeof(int16_t) + 15) / 16 * 16;
    volatile uint32_t *h_done = reinterpret_cast<volatile uint32_t *>(reinterpret_cast<char *>(s->h_out) + done_off);
    uint32_t *d_done = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s->d_out) + done_off);
    const uint32_t seq = ++s->seq ? s->seq : ++s->seq;          // never 0
#define PSGPU_SENF_CASE(NN) case NN: hipLaunchKernelGGL((ptm_frame_senone_kernel<NN>), dim3(1), dim3(kFrameThreads), \
        smem, s->stream, pv, (int32_t)fresh, (int32_t)(compallsen != 0), (int32_t)n_list,                         \
        (const uint16_t *)s->d_list, (const int32_t *)cur_cw, cur_sc, (const uint32_t *)cur_act, s->d_out,           \
        d_done, seq); break;
    switch (m->topn) {
        PSGPU_SENF_CASE(1) PSGPU_SENF_CASE(2) PSGPU_SENF_CASE(3) PSGPU_SENF_CASE(4)
        PSGPU_SENF_CASE(5) PSGPU_SENF_CASE(6) PSGPU_SENF_CASE(7) default: PSGPU_SENF_CASE(8)
    }
#undef PSGPU_SENF_CASE
    PSGPU_HIP(hipGetLastError());
    {
        // wait for the completion word; fall back to a stream sync (which also reports
        // launch failures) if it does not show up within a generous spin budget
        static const int spin_only = [] { const char *e = getenv("PSGPU_NO_POLL"); return e ? !atoi(e) : 1; }();
        bool done = false;
        if (spin_only) {
            for (long i = 0; i < 200000000L; ++i) {
                if (*h_done == seq) { done = true; break; }
                __builtin_ia32_pause();
            }
        }
        if (!done)
            PSGPU_HIP(hipStreamSynchronize(s->stream));
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    if (compallsen)
        memcpy(senscr, s->h_out, (size_t)m->n_sen * sizeof(int16_t));
    else {
        const int16_t rest = (int16_t)(uint16_t)(0u - h_done[1]);
        for (int i = 0; i < m->n_sen; ++i) senscr[i] = rest;
        for (int i = 0; i < n_list; ++i) senscr[s->h_list[i]] = s->h_out[i];
    }
    return PSGPU_OK;
}

int psgpu_ptm_state_get_topn(psgpu_ptm_state_t *s, int32_t slot, int32_t *cw, int32_t *score,
                             uint8_t *mgau_active)
{
    PSGPU_REQUIRE(s != nullptr, "psgpu_ptm_state_get_topn: NULL state");
    if (slot < 0) slot = s->cur;
    PSGPU_REQUIRE(slot < s->n_hist, "slot %d outside the %d-slot ring", slot, s->n_hist);
    const size_t slot_len = (size_t)s->m->n_chain * s->m->topn;
    { const int rc = la_flush(s); if (rc != PSGPU_OK) return rc; }
    PSGPU_HIP(hipStreamSynchronize(s->stream));
    if (cw) PSGPU_HIP(hipMemcpy(cw, s->hist_cw + slot * slot_len, slot_len * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (score) PSGPU_HIP(hipMemcpy(score, s->hist_sc + slot * slot_len, slot_len * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (mgau_active) {
        uint32_t act[8];
        PSGPU_HIP(hipMemcpy(act, s->hist_active + (size_t)slot * 8, sizeof act, hipMemcpyDeviceToHost));
        for (int cb = 0; cb < s->m->n_mgau; ++cb)
            mgau_active[cb] = (act[cb >> 5] >> (cb & 31)) & 1u;
    }
    return PSGPU_OK;
}

int psgpu_ptm_state_set_topn(psgpu_ptm_state_t *s, int32_t slot, const int32_t *cw,
                             const int32_t *score, const uint8_t *mgau_active)
{
    PSGPU_REQUIRE(s && cw && score, "psgpu_ptm_state_set_topn: NULL argument");
    PSGPU_REQUIRE(slot >= 0 && slot < s->n_hist, "slot %d outside the %d-slot ring", slot, s->n_hist);
    const psgpu_ptm_model_t *m = s->m;
    const size_t slot_len = (size_t)m->n_chain * m->topn;
    for (size_t i = 0; i < slot_len; ++i)
        PSGPU_REQUIRE(cw[i] >= 0 && cw[i] < m->n_density, "codeword %d outside the codebook", cw[i]);
    { const int rc = la_flush(s); if (rc != PSGPU_OK) return rc; }
    s->la.valid = 0;
    uint32_t act[8];
    memset(act, mgau_active ? 0 : 0xff, sizeof act);
    if (mgau_active)
        for (int cb = 0; cb < m->n_mgau; ++cb)
            if (mgau_active[cb]) act[cb >> 5] |= 1u << (cb & 31);
    PSGPU_HIP(hipStreamSynchronize(s->stream));
    PSGPU_HIP(hipMemcpy(s->hist_cw + slot * slot_len, cw, slot_len * sizeof(int32_t), hipMemcpyHostToDevice));
    PSGPU_HIP(hipMemcpy(s->hist_sc + slot * slot_len, score, slot_len * sizeof(int32_t), hipMemcpyHostToDevice));
    PSGPU_HIP(hipMemcpy(s->hist_active + (size_t)slot * 8, act, sizeof act, hipMemcpyHostToDevice));
    return PSGPU_OK;
}

}  // extern "C"
