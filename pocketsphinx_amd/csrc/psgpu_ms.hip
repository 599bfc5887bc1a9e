// psgpu_ms.hip -- the multi-stream / continuous-density scorer: device
// replacement of ms_cont_mgau_frame_eval() (reference src/ms_mgau.c:191-282)
// = gauden_dist()/compute_dist() (ms_gauden.c:424-509) + senone_eval()
// (ms_senone.c:357-407) + the best-score normalisation.
//
// Differences from the tied-mixture scorers that shape the kernels:
//   * top-N is STATELESS per frame (list reset to WORST_DIST, ms_gauden.c:438)
//     and keeps FLOAT distances; the sequential scan's result is the N largest
//     (dist, id) pairs in descending lexicographic order among densities with
//     dist >= WORST_DIST, so it is selected with N wave-wide arg-max rounds on
//     an order-preserving 64-bit key -- no history, frames are independent;
//   * unfilled list slots keep WORST_DIST and whatever id the slot held before
//     (the reference never clears ids): the per-call state keeps the lists in
//     HBM between calls; the batch entry reports such frames instead of
//     guessing (PSGPU_ESTATE);
//   * n_top >= n_density skips selection: every density, in index order
//     (compute_dist_all, :377-417);
//   * senone scores: fden = ((int32)dist + 1023) >> 10, 16-bit-domain
//     logmath_add through the senone log-add table (util/logmath.c:401-446),
//     `/ aw`, int16 clamp BEFORE and AFTER the best-score subtraction, and only
//     LISTED senones are written (unlisted entries of senscr keep stale values).
//
// Kernel 1: one wavefront per (frame, codebook, stream); densities on lanes
// (k*64 + lane, n_density <= 256).  Kernel 2: one workgroup per frame.
#include "psgpu_internal.h"
#include <cstring>
#include <cstdlib>
#include <vector>

constexpr int kMsMaxFeat = 8;
constexpr int kMsMaxTopn = 8;
constexpr int kMsK = 4;
constexpr int kMsLaLds = 4096;           // log-add entries kept in LDS (int32)
constexpr int kMsStage = 4096;           // (codebook, stream, rank) entries staged per frame

struct MsDev {
    const float *mean, *var, *det;
    const uint8_t *pdf;                  // [n_sen][n_feat][n_density]
    const uint8_t *pdf_t;                // [n_feat][n_density][n_sen] when senones share codebooks, else nullptr
    const int32_t *sen2mgau;
    const int32_t *logadd;               // widened to int32
    int32_t n_mgau, n_feat, n_density, n_sen, topn, aw, veclen, logadd_size, log_zero;
    int32_t featlen[kMsMaxFeat], featoff[kMsMaxFeat];
    const int64_t *cboff;                // [n_mgau * n_feat] float offset into mean/var
};

struct psgpu_ms_model_s {
    MsDev d;
    float *mean, *var, *det;
    uint8_t *pdf, *pdf_t;
    int32_t *sen2mgau, *logadd;
    int64_t *cboff;
    int32_t *h_sen2mgau;
    // per-call state (persistent lists, ms_mgau.c:150-152 msg->dist)
    int32_t *list_id; float *list_dist;  // [n_mgau][n_feat][topn]
    uint8_t *h_active, *d_active;        // mapped: active codebooks
    float *h_feat, *d_feat;              // mapped: one frame
    uint16_t *h_list, *d_list;           // mapped: listed senone ids
    int16_t *h_out, *d_out;              // mapped: scores of listed senones (list order) / all
    int32_t *d_flag;
    int32_t *d_best; int32_t cap_best;   // per-frame minima of the fused continuous path
    bool cont;                           // one stream, senone i owns codebook i, topn < n_density
    // look-ahead (psgpu_ms_lookahead): raw rows of announced frames on the host, their lists on the device
    struct {
        int valid, frame0, n, cap;
        bool dirty;                      // calls were served: the per-call lists lag behind
        float *h_feats, *d_feats;        // [cap][veclen]
        int16_t *h_rows, *d_rows;        // [cap][n_sen] raw (first clamp only); h_rows pinned
        int32_t *d_ids; float *d_dist;   // [n_mgau][n_feat][n][topn]
        int32_t *h_last, *d_last;        // [n_mgau]: frame whose list a served call left in codebook c, or -1
        int64_t served, batches;
    } la;
    hipStream_t stream;
    uint32_t seq;
};

// Wave-uniform model parameters read through the constant address space: the backend
// then issues scalar loads (SGPR operands, no VGPRs, no vector-memory traffic) even when
// the kernel has stores in flight that it cannot prove disjoint from the tables.
typedef const float __attribute__((address_space(4))) kfloat;
typedef const uint8_t __attribute__((address_space(4))) kbyte;
__device__ __forceinline__ kfloat *as_k(const float *p) { return (kfloat *)(uintptr_t)p; }
__device__ __forceinline__ kbyte *as_k(const uint8_t *p) { return (kbyte *)(uintptr_t)p; }

// order-preserving map float -> uint32 (larger float = larger key)
__device__ __forceinline__ uint32_t fkey(float f)
{
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off);
        v = (o > v) ? o : v;
    }
    return v;
}

// ---------------------------------------------------------------------------
// kernel 1: top-N float distances of one (frame, codebook, stream)
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256)
void ms_topn_kernel(MsDev p, const float *__restrict__ feats, int32_t n_frames,
                    const uint8_t *__restrict__ active,           // [n_mgau] or nullptr (all)
                    int32_t *__restrict__ list_id, float *__restrict__ list_dist,
                    int32_t batch,                                // 0: per-call state (stale ids are meaningful)
                    int32_t *__restrict__ flag)
{
    const int lane = threadIdx.x & 63;
    const long long wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6));
    const int per_frame = p.n_mgau * p.n_feat;
    if (wave >= (long long)n_frames * per_frame)
        return;
    const int t = (int)(wave / per_frame);
    const int c = (int)(wave - (long long)t * per_frame);
    const int m = c / p.n_feat, f = c - m * p.n_feat;
    if (active && !active[m])
        return;
    const int len = p.featlen[f];
    const float *mean = p.mean + p.cboff[c], *var = p.var + p.cboff[c];
    const float *det = p.det + (size_t)c * p.n_density;
    const float *x = feats + (size_t)t * p.veclen + p.featoff[f];
    float d[kMsK];
#pragma unroll
    for (int k = 0; k < kMsK; ++k) {
        const int cw = min(k * 64 + lane, p.n_density - 1);
        const float *mp = mean + (size_t)cw * len, *vp = var + (size_t)cw * len;
        float acc = det[cw];
        for (int j = 0; j < len; ++j) {
            // dval -= diff * diff * v[i]  (ms_gauden.c:405-408, :452-455): every op rounded, no FMA
            const float diff = __fsub_rn(x[j], mp[j]);
            acc = __fsub_rn(acc, __fmul_rn(__fmul_rn(diff, diff), vp[j]));
        }
        d[k] = acc;
    }
    // lists are codebook-major: [codebook*stream][frame][N]
    int32_t *oid = list_id + ((size_t)c * n_frames + t) * N;
    float *odist = list_dist + ((size_t)c * n_frames + t) * N;
    if (N >= p.n_density) {
        // compute_dist_all: every density in index order (n_density <= N <= 8 lanes)
        if (lane < p.n_density) { oid[lane] = lane; odist[lane] = d[0]; }
        return;
    }
    // eligible: dval >= WORST_DIST (NaN never is)
    unsigned long long key[kMsK];
#pragma unroll
    for (int k = 0; k < kMsK; ++k) {
        const int cw = k * 64 + lane;
        const bool ok = (cw < p.n_density) && (d[k] >= (float)kMaxNegInt32);
        key[k] = ok ? (((unsigned long long)fkey(d[k]) << 32) | (uint32_t)cw) : 0ull;
    }
    int filled = 0;
#pragma unroll
    for (int r = 0; r < N; ++r) {
        unsigned long long loc = key[0];
#pragma unroll
        for (int k = 1; k < kMsK; ++k) loc = key[k] > loc ? key[k] : loc;
        const unsigned long long w = wave_max_u64(loc);
        if (w != 0ull) {
            const int cw = (int)(w & 0xffffffffu);
            // the owner publishes the exact float and retires the candidate
#pragma unroll
            for (int k = 0; k < kMsK; ++k)
                if (key[k] == w) { odist[r] = d[k]; oid[r] = cw; key[k] = 0ull; }
            ++filled;
        }
        else if (lane == 0)
            odist[r] = (float)kMaxNegInt32;          // id keeps its previous contents (ms_gauden.c:438-440)
    }
    if (filled < N && batch && lane == 0 && flag)
        atomicOr(flag, 1);                           // batch mode cannot know the stale ids
}

// ---------------------------------------------------------------------------
// kernel 2: senone_eval over the listed senones of one frame + normalisation
// ---------------------------------------------------------------------------
constexpr int kMsSenThreads = 256;

template <int N>
__global__ __launch_bounds__(kMsSenThreads)
void ms_senone_kernel(MsDev p, int32_t compall, int32_t n_list, const uint16_t *__restrict__ list,
                      const int32_t *__restrict__ list_id, const float *__restrict__ list_dist,
                      int32_t n_frames, int16_t *__restrict__ out, int64_t out_stride,
                      uint32_t *__restrict__ done_word, uint32_t seq, int32_t raw)
{
    extern __shared__ __attribute__((aligned(16))) int32_t s_dyn[];      // [la entries] int32 | [n] int16
    __shared__ int32_t s_best;
    const int tid = threadIdx.x;
    const int frame = blockIdx.x;
    const bool la_lds = p.logadd_size <= kMsLaLds;
    int32_t *s_la = s_dyn;
    int16_t *s_scr = reinterpret_cast<int16_t *>(s_dyn + (la_lds ? p.logadd_size : 0));
    if (la_lds)
        for (int i = tid; i < p.logadd_size; i += kMsSenThreads) s_la[i] = p.logadd[i];
    if (tid == 0) s_best = 0x7fffffff;
    const int ntop = min(N, p.n_density);
    // Senones that share codebooks (pdf_t): the frame's (codebook, stream, rank) entries are
    // few, so their density scores fden and ids are computed ONCE into LDS instead of once
    // per senone.
    __shared__ int32_t s_fd[kMsStage];
    __shared__ uint8_t s_idb[kMsStage];
    const int n_ent = p.n_mgau * p.n_feat * N;
    const bool staged = p.pdf_t != nullptr && n_ent <= kMsStage;
    if (staged) {
        for (int e = tid; e < n_ent; e += kMsSenThreads) {
            const int c = e / N, t = e - c * N;
            const size_t li = ((size_t)c * n_frames + frame) * N + t;
            const float dv = list_dist[li];
            s_fd[e] = (dv < (float)kMaxNegInt32)
                ? (kMaxNegInt32 >> kSenscrShift)
                : (((int32_t)dv + ((1 << kSenscrShift) - 1)) >> kSenscrShift);
            s_idb[e] = (uint8_t)list_id[li];
        }
    }
    __syncthreads();
    const int32_t *la = la_lds ? s_la : p.logadd;
    const int n = compall ? p.n_sen : n_list;
    int32_t mybest = 0x7fffffff;
    for (int i = tid; i < n; i += kMsSenThreads) {
        const int sen = compall ? i : list[i];
        const int cb = p.sen2mgau[sen];
        // lists are codebook-major: entry (cb, f) of this frame at ((cb * n_feat + f) * n_frames + frame) * N
        const size_t lbase = ((size_t)cb * p.n_feat * n_frames + frame) * N;
        const size_t lstep = (size_t)n_frames * N;          // next stream
        int32_t scr = 0;
        for (int f = 0; f < p.n_feat; ++f) {
            // senone weights: [sen][f][cw] (own codebook per senone: one 16-byte row) or, when
            // senones share codebooks, the transposed copy [f][cw][sen] (coalesced along senones)
            const uint8_t *pdf = p.pdf_t ? p.pdf_t + (size_t)f * p.n_density * p.n_sen + sen
                                         : p.pdf + ((size_t)sen * p.n_feat + f) * p.n_density;
            const size_t pstep = p.pdf_t ? (size_t)p.n_sen : 1;
            int32_t fscr = 0;
            for (int t = 0; t < ntop; ++t) {
                int32_t fden, id;
                if (staged) {
                    const int e = (cb * p.n_feat + f) * N + t;
                    fden = s_fd[e]; id = s_idb[e];
                }
                else {
                    const float dv = list_dist[lbase + f * lstep + t];
                    fden = (dv < (float)kMaxNegInt32)
                        ? (kMaxNegInt32 >> kSenscrShift)
                        : (((int32_t)dv + ((1 << kSenscrShift) - 1)) >> kSenscrShift);
                    id = list_id[lbase + f * lstep + t];
                }
                const int32_t fw = fden - (int32_t)pdf[(size_t)id * pstep];
                if (t == 0) fscr = fw;
                else {
                    // logmath_add (util/logmath.c:401-446)
                    if (fscr <= p.log_zero) fscr = fw;
                    else if (fw > p.log_zero) {
                        const int32_t r = max(fscr, fw);
                        const int32_t dd = r - min(fscr, fw);
                        fscr = (dd < 0 || dd >= p.logadd_size) ? r : r + la[dd];
                    }
                }
            }
            scr -= fscr;
        }
        scr /= p.aw;                                     // C division, truncates toward zero
        scr = max(-32768, min(32767, scr));              // senone_eval's clamp + int16 store (ms_mgau.c:219)
        s_scr[i] = (int16_t)scr;
        mybest = min(mybest, scr);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mybest = min(mybest, __shfl_xor(mybest, off));
    if ((tid & 63) == 0) atomicMin(&s_best, mybest);
    __syncthreads();
    const int32_t best = raw ? 0 : s_best;               // raw: stop after senone_eval's own clamp (look-ahead rows)
    int16_t *o = out + (size_t)frame * out_stride;
    for (int i = tid; i < n; i += kMsSenThreads) {
        int32_t bs = (int32_t)s_scr[i] - best;
        bs = max(-32768, min(32767, bs));
        o[i] = (int16_t)bs;                              // list order (per-call) / senone order (compall)
    }
    if (done_word) {                                     // per-call entry: completion word for the polling host
        __threadfence_system();
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store(done_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T>
static int upl(T **dst, const T *src, size_t n)
{
    PSGPU_HIP(hipMalloc((void **)dst, n * sizeof(T) ? n * sizeof(T) : 1));
    PSGPU_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PSGPU_OK;
}

static void launch_topn(const MsDev &d, const float *feats, int32_t T, const uint8_t *active,
                        int32_t *ids, float *dist, int32_t batch, int32_t *flag, hipStream_t st)
{
    const long long waves = (long long)T * d.n_mgau * d.n_feat;
    const int blocks = (int)((waves + 3) / 4);
#define PSGPU_MS_TOPN(NN) case NN: hipLaunchKernelGGL((ms_topn_kernel<NN>), dim3(blocks), dim3(256), 0, st, \
        d, feats, T, active, ids, dist, batch, flag); break;
    switch (d.topn) {
        PSGPU_MS_TOPN(1) PSGPU_MS_TOPN(2) PSGPU_MS_TOPN(3) PSGPU_MS_TOPN(4)
        PSGPU_MS_TOPN(5) PSGPU_MS_TOPN(6) PSGPU_MS_TOPN(7) default: PSGPU_MS_TOPN(8)
    }
#undef PSGPU_MS_TOPN
}

static void launch_senone(const MsDev &d, int32_t T, int32_t compall, int32_t n_list, const uint16_t *list,
                          const int32_t *ids, const float *dist, int16_t *out,
                          int64_t out_stride, hipStream_t st, uint32_t *done_word = nullptr, uint32_t seq = 0,
                          int32_t raw = 0)
{
    const int n = compall ? d.n_sen : n_list;
    const size_t smem = ((size_t)(d.logadd_size <= kMsLaLds ? d.logadd_size : 0) * 4 +
                         (size_t)(n > 0 ? n : 1) * 2 + 15) / 16 * 16;
#define PSGPU_MS_SEN(NN) case NN: hipLaunchKernelGGL((ms_senone_kernel<NN>), dim3(T), dim3(kMsSenThreads), smem, st, \
        d, compall, n_list, list, ids, dist, T, out, out_stride, done_word, seq, raw); break;
    switch (d.topn) {
        PSGPU_MS_SEN(1) PSGPU_MS_SEN(2) PSGPU_MS_SEN(3) PSGPU_MS_SEN(4)
        PSGPU_MS_SEN(5) PSGPU_MS_SEN(6) PSGPU_MS_SEN(7) default: PSGPU_MS_SEN(8)
    }
#undef PSGPU_MS_SEN
}

// ---------------------------------------------------------------------------
// kernel 1b (batched entry): frames on lanes.  One wavefront = one
// (codebook, stream) x 64 consecutive frames; the codebook's parameters are
// wave-uniform (scalar cache -> SGPR operands), the lane's feature values sit
// in VGPRs, densities are visited in index order and each lane keeps its N
// best 64-bit keys (order-preserving float bits << 32 | density) sorted with a
// max/min bubble.  No cross-lane traffic, no history: exact by construction.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned long long umin64(unsigned long long a, unsigned long long b) { return a > b ? b : a; }

template <int N, int LEN>
__global__ __launch_bounds__(256)
void ms_lane_kernel(MsDev p, const float *__restrict__ feats, int32_t n_frames,
                    int32_t *__restrict__ list_id, float *__restrict__ list_dist,
                    int32_t *__restrict__ flag)
{
    const int lane = threadIdx.x & 63;
    const long long wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6));
    const int n_tiles = (n_frames + 63) >> 6;
    const int c = (int)(wave / n_tiles);                       // consecutive waves: tiles of ONE codebook
    if (c >= p.n_mgau * p.n_feat)
        return;
    const int tile = (int)(wave - (long long)c * n_tiles);
    const int f = c % p.n_feat;
    const int t = tile * 64 + lane;
    const bool valid = t < n_frames;
    const int tl = valid ? t : n_frames - 1;
    float x[LEN];
    {
        const float *xp = feats + (size_t)tl * p.veclen + p.featoff[f];
#pragma unroll
        for (int j = 0; j < LEN; ++j) x[j] = xp[j];
    }
    kfloat *mean = as_k(p.mean + p.cboff[c]), *var = as_k(p.var + p.cboff[c]);
    kfloat *det = as_k(p.det + (size_t)c * p.n_density);
    int32_t *oid = list_id + ((size_t)c * n_frames + t) * N;
    float *odist = list_dist + ((size_t)c * n_frames + t) * N;

    if (N >= p.n_density) {                                    // compute_dist_all: index order, no selection
        for (int dn = 0; dn < p.n_density; ++dn) {
            kfloat *m = mean + dn * LEN, *v = var + dn * LEN;
            float acc = det[dn];
#pragma unroll
            for (int j = 0; j < LEN; ++j) {
                const float diff = __fsub_rn(x[j], m[j]);
                acc = __fsub_rn(acc, __fmul_rn(__fmul_rn(diff, diff), v[j]));
            }
            if (valid) { oid[dn] = dn; odist[dn] = acc; }
        }
        return;
    }
    unsigned long long key[N];
#pragma unroll
    for (int r = 0; r < N; ++r) key[r] = 0ull;
    for (int dn = 0; dn < p.n_density; ++dn) {
        kfloat *m = mean + dn * LEN, *v = var + dn * LEN;
        float acc = det[dn];
#pragma unroll
        for (int j = 0; j < LEN; ++j) {
            const float diff = __fsub_rn(x[j], m[j]);
            acc = __fsub_rn(acc, __fmul_rn(__fmul_rn(diff, diff), v[j]));
        }
        unsigned long long k = (acc >= (float)kMaxNegInt32)
            ? (((unsigned long long)fkey(acc) << 32) | (uint32_t)dn) : 0ull;
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const unsigned long long hi = umax64(key[r], k);
            k = umin64(key[r], k);
            key[r] = hi;
        }
    }
    if (valid) {
        bool unfilled = false;
#pragma unroll
        for (int r = 0; r < N; ++r) {
            if (key[r] != 0ull) {
                const uint32_t fk = (uint32_t)(key[r] >> 32);
                const uint32_t u = (fk & 0x80000000u) ? (fk & 0x7fffffffu) : ~fk;   // inverse of fkey
                odist[r] = __builtin_bit_cast(float, u);
                oid[r] = (int32_t)(key[r] & 0xffffffffu);
            }
            else { odist[r] = (float)kMaxNegInt32; oid[r] = 0; unfilled = true; }
        }
        if (unfilled) atomicOr(flag, 1);                       // stale ids: only the per-call entry knows them
    }
}

template <int N>
static bool launch_lane_n(const MsDev &d, const float *feats, int32_t T, int32_t *ids, float *dist,
                          int32_t *flag, hipStream_t st)
{
    int len = d.featlen[0];
    for (int f = 1; f < d.n_feat; ++f) if (d.featlen[f] != len) return false;
    const long long waves = (long long)((T + 63) / 64) * d.n_mgau * d.n_feat;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    if (len == 13)
        hipLaunchKernelGGL((ms_lane_kernel<N, 13>), dim3(blocks), dim3(256), 0, st, d, feats, T, ids, dist, flag);
    else if (len == 39)
        hipLaunchKernelGGL((ms_lane_kernel<N, 39>), dim3(blocks), dim3(256), 0, st, d, feats, T, ids, dist, flag);
    else
        return false;
    return true;
}

static bool launch_lane(const MsDev &d, const float *feats, int32_t T, int32_t *ids, float *dist,
                        int32_t *flag, hipStream_t st)
{
    switch (d.topn) {
    case 1: return launch_lane_n<1>(d, feats, T, ids, dist, flag, st);
    case 2: return launch_lane_n<2>(d, feats, T, ids, dist, flag, st);
    case 3: return launch_lane_n<3>(d, feats, T, ids, dist, flag, st);
    case 4: return launch_lane_n<4>(d, feats, T, ids, dist, flag, st);
    case 5: return launch_lane_n<5>(d, feats, T, ids, dist, flag, st);
    case 6: return launch_lane_n<6>(d, feats, T, ids, dist, flag, st);
    case 7: return launch_lane_n<7>(d, feats, T, ids, dist, flag, st);
    default: return launch_lane_n<8>(d, feats, T, ids, dist, flag, st);
    }
}

// ---------------------------------------------------------------------------
// fused kernel for fully continuous models (batched entry): every senone owns
// its codebook (".cont." mapping, ms_senone.c:305-315) and there is one stream.
// Frames on lanes as in ms_lane_kernel, but the wave goes straight on from the
// N best densities of (senone c, frame t) to the senone's score -- the list
// never round-trips through HBM (32 B per senone-frame against 2 B of output;
// the list buffers are still filled when the caller passes them).  A workgroup
// = 64 frames x kContG consecutive senones (4 waves x kContG/4 senones each,
// the lane's feature vector loaded once); scores are transposed through LDS so
// that a frame's kContG scores leave as one contiguous segment; the per-frame
// minimum goes to best[t] with one atomicMin per (workgroup, frame), and
// ms_cont_norm_kernel applies ms_cont_mgau_frame_eval's second clamp
// (ms_mgau.c:229-234).
// ---------------------------------------------------------------------------
constexpr int kContG = 64;

template <int N, int LEN>
__global__ __launch_bounds__(256)
void ms_cont_kernel(MsDev p, const float *__restrict__ feats, int32_t n_frames,
                    int32_t *__restrict__ list_id, float *__restrict__ list_dist,
                    int16_t *__restrict__ out, int64_t out_stride, int32_t *__restrict__ best,
                    int32_t *__restrict__ flag)
{
    extern __shared__ __attribute__((aligned(16))) int32_t s_dyn[];      // [la entries] int32
    __shared__ int16_t s_tile[64][kContG + 2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n_tiles = (n_frames + 63) >> 6;
    const int tile = blockIdx.x % n_tiles, grp = blockIdx.x / n_tiles;   // consecutive workgroups: same senones
    const int g0 = grp * kContG;
    const int t = tile * 64 + lane;
    const bool valid = t < n_frames;
    const int tl = valid ? t : n_frames - 1;
    const bool la_lds = p.logadd_size <= kMsLaLds;
    if (la_lds)
        for (int i = threadIdx.x; i < p.logadd_size; i += 256) s_dyn[i] = p.logadd[i];
    __syncthreads();
    const int32_t *la = la_lds ? s_dyn : p.logadd;
    float x[LEN];
    {
        const float *xp = feats + (size_t)tl * p.veclen;
#pragma unroll
        for (int j = 0; j < LEN; ++j) x[j] = xp[j];
    }
    for (int k = 0; k < kContG / 4; ++k) {
        const int c = __builtin_amdgcn_readfirstlane(g0 + w * (kContG / 4) + k);
        if (c >= p.n_sen) break;
        kfloat *mean = as_k(p.mean + p.cboff[c]), *var = as_k(p.var + p.cboff[c]);
        kfloat *det = as_k(p.det + (size_t)c * p.n_density);
        unsigned long long key[N];
#pragma unroll
        for (int r = 0; r < N; ++r) key[r] = 0ull;
        for (int dn = 0; dn < p.n_density; ++dn) {
            kfloat *m = mean + dn * LEN, *v = var + dn * LEN;
            float acc = det[dn];
#pragma unroll
            for (int j = 0; j < LEN; ++j) {
                const float diff = __fsub_rn(x[j], m[j]);
                acc = __fsub_rn(acc, __fmul_rn(__fmul_rn(diff, diff), v[j]));
            }
            unsigned long long kk = (acc >= (float)kMaxNegInt32)
                ? (((unsigned long long)fkey(acc) << 32) | (uint32_t)dn) : 0ull;
#pragma unroll
            for (int r = 0; r < N; ++r) {
                const unsigned long long hi = umax64(key[r], kk);
                kk = umin64(key[r], kk);
                key[r] = hi;
            }
        }
        // senone_eval (ms_senone.c:357-407) on the list, best first
        const uint8_t *pdf = p.pdf + (size_t)c * p.n_density;
        int32_t fscr = 0;
        bool unfilled = false;
#pragma unroll
        for (int r = 0; r < N; ++r) {
            float dv; int32_t id;
            if (key[r] != 0ull) {
                const uint32_t fk = (uint32_t)(key[r] >> 32);
                const uint32_t u = (fk & 0x80000000u) ? (fk & 0x7fffffffu) : ~fk;   // inverse of fkey
                dv = __builtin_bit_cast(float, u);
                id = (int32_t)(key[r] & 0xffffffffu);
            }
            else { dv = (float)kMaxNegInt32; id = 0; unfilled = true; }
            if (list_id && valid) {
                const size_t li = ((size_t)c * n_frames + t) * N + r;
                list_id[li] = id; list_dist[li] = dv;
            }
            const int32_t fden = (dv < (float)kMaxNegInt32)
                ? (kMaxNegInt32 >> kSenscrShift)
                : (((int32_t)dv + ((1 << kSenscrShift) - 1)) >> kSenscrShift);
            const int32_t fw = fden - (int32_t)pdf[id];
            if (r == 0) fscr = fw;
            else if (fscr <= p.log_zero) fscr = fw;                 // logmath_add (util/logmath.c:401-446)
            else if (fw > p.log_zero) {
                const int32_t hi = max(fscr, fw);
                const int32_t dd = hi - min(fscr, fw);
                fscr = (dd < 0 || dd >= p.logadd_size) ? hi : hi + la[dd];
            }
        }
        if (unfilled && valid) atomicOr(flag, 1);
        int32_t scr = -fscr;
        scr /= p.aw;
        scr = max(-32768, min(32767, scr));
        s_tile[lane][c - g0] = (int16_t)scr;
    }
    __syncthreads();
    const int ng = min(kContG, p.n_sen - g0);
    if (threadIdx.x < 64 && valid) {                               // per-frame minimum of this group
        int32_t mn = 0x7fffffff;
        for (int i = 0; i < ng; ++i) mn = min(mn, (int32_t)s_tile[lane][(i + lane) % ng]);
        atomicMin(&best[t], mn);
    }
    // rows out: 4 threads per frame, 16 senones each (32 contiguous bytes per thread)
    {
        const int fr = threadIdx.x >> 2, part = threadIdx.x & 3;
        const int tt = tile * 64 + fr;
        if (tt < n_frames) {
            int16_t *o = out + (size_t)tt * out_stride + g0;
            for (int i = part * (kContG / 4); i < (part + 1) * (kContG / 4) && i < ng; ++i) o[i] = s_tile[fr][i];
        }
    }
}

__global__ __launch_bounds__(256)
void ms_cont_norm_kernel(int16_t *__restrict__ out, int64_t out_stride, int32_t n_sen, const int32_t *__restrict__ best)
{
    const int t = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_sen) return;
    int16_t *o = out + (size_t)t * out_stride;
    int32_t bs = (int32_t)o[i] - best[t];
    bs = max(-32768, min(32767, bs));
    o[i] = (int16_t)bs;
}

template <int N>
static bool launch_cont_n(const MsDev &d, const float *feats, int32_t T, int32_t *ids, float *dist,
                          int16_t *out, int64_t out_stride, int32_t *best, int32_t *flag, hipStream_t st, bool raw)
{
    const int n_tiles = (T + 63) / 64, n_grp = (d.n_sen + kContG - 1) / kContG;
    const size_t smem = (size_t)(d.logadd_size <= kMsLaLds ? d.logadd_size : 0) * 4;
    const dim3 grid((unsigned)(n_tiles * (long long)n_grp));
    if (d.featlen[0] == 13)
        hipLaunchKernelGGL((ms_cont_kernel<N, 13>), grid, dim3(256), smem, st, d, feats, T, ids, dist, out, out_stride, best, flag);
    else if (d.featlen[0] == 39)
        hipLaunchKernelGGL((ms_cont_kernel<N, 39>), grid, dim3(256), smem, st, d, feats, T, ids, dist, out, out_stride, best, flag);
    else
        return false;
    if (!raw)
        hipLaunchKernelGGL(ms_cont_norm_kernel, dim3((d.n_sen + 255) / 256, T), dim3(256), 0, st, out, out_stride, d.n_sen, best);
    return true;
}

static bool launch_cont(const MsDev &d, const float *feats, int32_t T, int32_t *ids, float *dist,
                        int16_t *out, int64_t out_stride, int32_t *best, int32_t *flag, hipStream_t st, bool raw = false)
{
    switch (d.topn) {
    case 1: return launch_cont_n<1>(d, feats, T, ids, dist, out, out_stride, best, flag, st, raw);
    case 2: return launch_cont_n<2>(d, feats, T, ids, dist, out, out_stride, best, flag, st, raw);
    case 3: return launch_cont_n<3>(d, feats, T, ids, dist, out, out_stride, best, flag, st, raw);
    case 4: return launch_cont_n<4>(d, feats, T, ids, dist, out, out_stride, best, flag, st, raw);
    case 5: return launch_cont_n<5>(d, feats, T, ids, dist, out, out_stride, best, flag, st, raw);
    case 6: return launch_cont_n<6>(d, feats, T, ids, dist, out, out_stride, best, flag, st, raw);
    case 7: return launch_cont_n<7>(d, feats, T, ids, dist, out, out_stride, best, flag, st, raw);
    default: return launch_cont_n<8>(d, feats, T, ids, dist, out, out_stride, best, flag, st, raw);
    }
}

static void ms_la_release(psgpu_ms_model_t *m);

extern "C" {

int psgpu_ms_model_create(psgpu_ms_model_t **out, int32_t n_mgau, int32_t n_feat, int32_t n_density,
                          const int32_t *featlen, int32_t n_sen, int32_t topn, int32_t aw,
                          const float *mean, const float *var, const float *det,
                          const uint8_t *pdf, const uint32_t *sen2mgau,
                          const void *logadd, int32_t logadd_size, int32_t logadd_width, int32_t log_zero)
{
    PSGPU_REQUIRE(out && featlen && mean && var && det && pdf && sen2mgau && logadd,
                  "psgpu_ms_model_create: NULL argument");
    PSGPU_REQUIRE(n_mgau >= 1 && n_feat >= 1 && n_feat <= kMsMaxFeat, "n_mgau %d / n_feat %d unsupported", n_mgau, n_feat);
    PSGPU_REQUIRE(n_density >= 1 && n_density <= 64 * kMsK, "n_density %d outside 1..%d", n_density, 64 * kMsK);
    PSGPU_REQUIRE(topn >= 1 && (topn <= kMsMaxTopn), "topn %d outside 1..%d", topn, kMsMaxTopn);
    PSGPU_REQUIRE(topn <= n_density, "topn %d > n_density %d (ms_mgau_init clamps it, ms_mgau.c:141-147)", topn, n_density);
    PSGPU_REQUIRE(aw >= 1, "aw %d < 1", aw);
    PSGPU_REQUIRE(n_sen > 0 && n_sen <= 32768, "n_sen %d outside 1..32768", n_sen);
    PSGPU_REQUIRE(logadd_size >= 1 && (logadd_width == 1 || logadd_width == 2 || logadd_width == 4),
                  "bad log-add table (size %d, width %d)", logadd_size, logadd_width);
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;
    psgpu_ms_model_t *m = new psgpu_ms_model_t();
    memset(m, 0, sizeof *m);
    MsDev &d = m->d;
    d.n_mgau = n_mgau; d.n_feat = n_feat; d.n_density = n_density; d.n_sen = n_sen;
    d.topn = topn; d.aw = aw; d.logadd_size = logadd_size; d.log_zero = log_zero;
    std::vector<int64_t> cboff((size_t)n_mgau * n_feat);
    int64_t o = 0;
    for (int f = 0; f < n_feat; ++f) { d.featlen[f] = featlen[f]; d.featoff[f] = d.veclen; d.veclen += featlen[f]; }
    for (int g = 0; g < n_mgau; ++g)
        for (int f = 0; f < n_feat; ++f) { cboff[(size_t)g * n_feat + f] = o; o += (int64_t)n_density * featlen[f]; }
    std::vector<int32_t> la((size_t)logadd_size), map((size_t)n_sen);
    for (int i = 0; i < logadd_size; ++i)
        la[i] = logadd_width == 1 ? ((const uint8_t *)logadd)[i]
              : logadd_width == 2 ? ((const uint16_t *)logadd)[i] : (int32_t)((const uint32_t *)logadd)[i];
    for (int i = 0; i < n_sen; ++i) {
        if (sen2mgau[i] >= (uint32_t)n_mgau) {
            psgpu_set_error("senone %d maps to codebook %u >= n_mgau %d", i, sen2mgau[i], n_mgau);
            delete m;
            return PSGPU_EINVAL;
        }
        map[i] = (int32_t)sen2mgau[i];
    }
    const size_t nlist = (size_t)n_mgau * n_feat * topn;
    hipError_t e = hipSuccess;
    if ((rc = upl(&m->mean, mean, (size_t)o)) || (rc = upl(&m->var, var, (size_t)o)) ||
        (rc = upl(&m->det, det, (size_t)n_mgau * n_feat * n_density)) ||
        (rc = upl(&m->pdf, pdf, (size_t)n_sen * n_feat * n_density)) ||
        (rc = upl(&m->sen2mgau, map.data(), map.size())) ||
        (rc = upl(&m->logadd, la.data(), la.size())) ||
        (rc = upl(&m->cboff, cboff.data(), cboff.size()))) {
        psgpu_ms_model_free(m);
        return rc;
    }
    if (n_mgau * 2 <= n_sen) {
        std::vector<uint8_t> pt((size_t)n_sen * n_feat * n_density);
        for (int i = 0; i < n_sen; ++i)
            for (int f = 0; f < n_feat; ++f)
                for (int c = 0; c < n_density; ++c)
                    pt[((size_t)f * n_density + c) * n_sen + i] = pdf[((size_t)i * n_feat + f) * n_density + c];
        if ((rc = upl(&m->pdf_t, pt.data(), pt.size()))) { psgpu_ms_model_free(m); return rc; }
    }
    m->h_sen2mgau = (int32_t *)malloc(sizeof(int32_t) * n_sen);
    memcpy(m->h_sen2mgau, map.data(), sizeof(int32_t) * n_sen);
    m->cont = n_feat == 1 && n_mgau == n_sen && topn < n_density;
    for (int i = 0; i < n_sen && m->cont; ++i) m->cont = map[i] == i;
    e = hipMalloc((void **)&m->list_id, nlist * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(m->list_id, 0, nlist * sizeof(int32_t));     // ckd_calloc_3d (ms_mgau.c:150)
    if (e == hipSuccess) e = hipMalloc((void **)&m->list_dist, nlist * sizeof(float));
    if (e == hipSuccess) e = hipMemset(m->list_dist, 0, nlist * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_flag, sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(m->d_flag, 0, sizeof(int32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&m->h_active, (size_t)n_mgau, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void **)&m->h_feat, (size_t)d.veclen * sizeof(float), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void **)&m->h_list, (size_t)n_sen * sizeof(uint16_t), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void **)&m->h_out, ((size_t)n_sen * sizeof(int16_t) + 15) / 16 * 16 + 16, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&m->d_active, m->h_active, 0);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&m->d_feat, m->h_feat, 0);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&m->d_list, m->h_list, 0);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&m->d_out, m->h_out, 0);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        psgpu_set_error("psgpu_ms_model_create: %s", hipGetErrorString(e));
        psgpu_ms_model_free(m);
        return e == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP;
    }
    d.mean = m->mean; d.var = m->var; d.det = m->det; d.pdf = m->pdf; d.pdf_t = m->pdf_t;
    d.sen2mgau = m->sen2mgau; d.logadd = m->logadd; d.cboff = m->cboff;
    *out = m;
    return PSGPU_OK;
}

void psgpu_ms_model_free(psgpu_ms_model_t *m)
{
    if (!m) return;
    if (m->stream) hipStreamDestroy(m->stream);
    hipFree(m->mean); hipFree(m->var); hipFree(m->det); hipFree(m->pdf); hipFree(m->pdf_t);
    hipFree(m->sen2mgau); hipFree(m->logadd); hipFree(m->cboff);
    hipFree(m->list_id); hipFree(m->list_dist); hipFree(m->d_flag); hipFree(m->d_best);
    ms_la_release(m);
    if (m->h_active) hipHostFree(m->h_active);
    if (m->h_feat) hipHostFree(m->h_feat);
    if (m->h_list) hipHostFree(m->h_list);
    if (m->h_out) hipHostFree(m->h_out);
    free(m->h_sen2mgau);
    delete m;
}

int32_t psgpu_ms_n_sen(const psgpu_ms_model_t *m) { return m->d.n_sen; }
int32_t psgpu_ms_veclen(const psgpu_ms_model_t *m) { return m->d.veclen; }


static int ms_batch(psgpu_ms_model_t *m, const float *feats_dev, int32_t total_frames, int32_t *list_id_dev,
                    float *list_dist_dev, int16_t *senscr_dev, void *stream, bool raw);

// ---- look-ahead -----------------------------------------------------------
// The scorer has no time dependence (ms_mgau.c:207), so the frames a caller announces can
// be scored in one batched pass (compallsen, stopping after senone_eval's own clamp) and
// every later frame_eval call on one of them -- any pass, any active list -- is answered on
// the host from the raw row: best over the listed senones, subtract, clamp
// (ms_mgau.c:219-234).  The only thing a call leaves behind is the list of ids of each
// active codebook (msg->dist, ms_gauden.c:438-440: unfilled slots of a later call keep
// them); served calls record which frame's list that is (la.h_last) and the per-call lists
// are brought up to date from the batch's lists before the per-call kernels run again.
__global__ void ms_la_sync_kernel(MsDev p, const int32_t *__restrict__ last, int32_t frame0, int32_t n,
                                  const int32_t *__restrict__ b_id, const float *__restrict__ b_dist,
                                  int32_t *__restrict__ list_id, float *__restrict__ list_dist)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;        // (codebook, stream, rank)
    const int per = p.n_feat * p.topn;
    if (e >= p.n_mgau * per) return;
    const int g = e / per;
    const int fr = last[g];
    if (fr < 0) return;
    const int chain = e / p.topn, r = e - chain * p.topn;
    const size_t src = ((size_t)chain * n + (fr - frame0)) * p.topn + r;
    list_id[e] = b_id[src];
    list_dist[e] = b_dist[src];
}

static void ms_la_release(psgpu_ms_model_t *m)
{
    free(m->la.h_feats); hipFree(m->la.d_feats); hipFree(m->la.d_rows); hipFree(m->la.d_ids); hipFree(m->la.d_dist);
    if (m->la.h_rows) hipHostFree(m->la.h_rows);
    if (m->la.h_last) hipHostFree(m->la.h_last);
    memset(&m->la, 0, sizeof m->la);
}

// per-call lists := what the served calls would have left
static int ms_la_sync(psgpu_ms_model_t *m)
{
    if (!m->la.dirty) return PSGPU_OK;
    const MsDev &d = m->d;
    const int n_ent = d.n_mgau * d.n_feat * d.topn;
    hipLaunchKernelGGL(ms_la_sync_kernel, dim3((n_ent + 255) / 256), dim3(256), 0, m->stream, d, m->la.d_last,
                       m->la.frame0, m->la.n, m->la.d_ids, m->la.d_dist, m->list_id, m->list_dist);
    PSGPU_HIP(hipGetLastError());
    PSGPU_HIP(hipStreamSynchronize(m->stream));
    for (int g = 0; g < d.n_mgau; ++g) m->la.h_last[g] = -1;
    m->la.dirty = false;
    return PSGPU_OK;
}

int psgpu_ms_lookahead(psgpu_ms_model_t *m, const float *feats, int32_t frame0, int32_t n_frames)
{
    PSGPU_REQUIRE(m && (feats || n_frames == 0) && frame0 >= 0 && n_frames >= 0, "psgpu_ms_lookahead: bad argument");
    const MsDev &d = m->d;
    int rc = ms_la_sync(m);                                      // the old cache's lists are still needed for this
    if (rc != PSGPU_OK) return rc;
    m->la.valid = 0;
    if (n_frames == 0) return PSGPU_OK;
    if (n_frames > m->la.cap) {
        int64_t served = m->la.served, batches = m->la.batches;
        ms_la_release(m);
        m->la.served = served; m->la.batches = batches;
        const size_t nl = (size_t)d.n_mgau * d.n_feat * n_frames * d.topn;
        m->la.h_feats = (float *)malloc(sizeof(float) * (size_t)n_frames * d.veclen);
        if (!m->la.h_feats) { psgpu_set_error("out of host memory"); return PSGPU_ENOMEM; }
        PSGPU_HIP(hipMalloc((void **)&m->la.d_feats, sizeof(float) * (size_t)n_frames * d.veclen));
        PSGPU_HIP(hipMalloc((void **)&m->la.d_rows, sizeof(int16_t) * (size_t)n_frames * d.n_sen));
        PSGPU_HIP(hipMalloc((void **)&m->la.d_ids, sizeof(int32_t) * nl));
        PSGPU_HIP(hipMalloc((void **)&m->la.d_dist, sizeof(float) * nl));
        PSGPU_HIP(hipHostMalloc((void **)&m->la.h_rows, sizeof(int16_t) * (size_t)n_frames * d.n_sen, hipHostMallocDefault));
        PSGPU_HIP(hipHostMalloc((void **)&m->la.h_last, sizeof(int32_t) * (size_t)d.n_mgau, hipHostMallocMapped));
        PSGPU_HIP(hipHostGetDevicePointer((void **)&m->la.d_last, m->la.h_last, 0));
        for (int g = 0; g < d.n_mgau; ++g) m->la.h_last[g] = -1;
        m->la.cap = n_frames;
    }
    memcpy(m->la.h_feats, feats, sizeof(float) * (size_t)n_frames * d.veclen);
    PSGPU_HIP(hipMemcpyAsync(m->la.d_feats, m->la.h_feats, sizeof(float) * (size_t)n_frames * d.veclen,
                             hipMemcpyHostToDevice, m->stream));
    rc = ms_batch(m, m->la.d_feats, n_frames, m->la.d_ids, m->la.d_dist, m->la.d_rows, m->stream, true);
    if (rc != PSGPU_OK) return rc;
    int32_t flag = 0;
    PSGPU_HIP(hipMemcpyAsync(&flag, m->d_flag, sizeof flag, hipMemcpyDeviceToHost, m->stream));
    PSGPU_HIP(hipMemcpyAsync(m->la.h_rows, m->la.d_rows, sizeof(int16_t) * (size_t)n_frames * d.n_sen,
                             hipMemcpyDeviceToHost, m->stream));
    PSGPU_HIP(hipStreamSynchronize(m->stream));
    if (flag) return PSGPU_OK;             // a list with unfilled slots: only the per-call path knows its stale ids
    m->la.frame0 = frame0; m->la.n = n_frames; m->la.valid = 1;
    ++m->la.batches;
    return PSGPU_OK;
}

int psgpu_ms_lookahead_covers(const psgpu_ms_model_t *m, const float *feat, int32_t frame)
{
    if (!m || !feat || !m->la.valid || frame < m->la.frame0 || frame >= m->la.frame0 + m->la.n) return 0;
    return memcmp(feat, m->la.h_feats + (size_t)(frame - m->la.frame0) * m->d.veclen, sizeof(float) * m->d.veclen) == 0;
}

int psgpu_ms_lookahead_stats(const psgpu_ms_model_t *m, int64_t *served, int64_t *batches)
{
    PSGPU_REQUIRE(m != nullptr, "psgpu_ms_lookahead_stats: NULL model");
    if (served) *served = m->la.served;
    if (batches) *batches = m->la.batches;
    return PSGPU_OK;
}

int psgpu_ms_frame_eval_at(psgpu_ms_model_t *m, int16_t *senscr, const uint8_t *senone_active,
                           int32_t n_senone_active, const float *feat, int32_t frame, int32_t compallsen)
{
    PSGPU_REQUIRE(m && senscr && feat, "psgpu_ms_frame_eval_at: NULL argument");
    PSGPU_REQUIRE(compallsen || n_senone_active == 0 || senone_active, "active list missing");
    const MsDev &d = m->d;
    if (!psgpu_ms_lookahead_covers(m, feat, frame)) {
        int rc = ms_la_sync(m);
        if (rc != PSGPU_OK) return rc;
        return psgpu_ms_frame_eval(m, senscr, senone_active, n_senone_active, feat, compallsen);
    }
    const int16_t *row = m->la.h_rows + (size_t)(frame - m->la.frame0) * d.n_sen;
    if (compallsen) {
        int32_t best = 0x7fffffff;
        for (int i = 0; i < d.n_sen; ++i) best = row[i] < best ? row[i] : best;
        for (int i = 0; i < d.n_sen; ++i) {
            int32_t bs = (int32_t)row[i] - best;
            senscr[i] = (int16_t)(bs > 32767 ? 32767 : bs < -32768 ? -32768 : bs);
        }
        for (int g = 0; g < d.n_mgau; ++g) m->la.h_last[g] = frame;
    }
    else {
        int32_t best = 0x7fffffff;
        int sen = 0;
        for (int i = 0; i < n_senone_active; ++i) {
            sen += senone_active[i];
            if (sen >= d.n_sen) {
                psgpu_set_error("active list runs past n_sen (%d >= %d)", sen, d.n_sen);
                return PSGPU_EINVAL;
            }
            best = row[sen] < best ? row[sen] : best;
        }
        sen = 0;
        for (int i = 0; i < n_senone_active; ++i) {
            sen += senone_active[i];
            int32_t bs = (int32_t)row[sen] - best;
            senscr[sen] = (int16_t)(bs > 32767 ? 32767 : bs < -32768 ? -32768 : bs);
            m->la.h_last[m->h_sen2mgau[sen]] = frame;
        }
        if (n_senone_active == 0) return PSGPU_OK;              // nothing listed, nothing evaluated
    }
    m->la.dirty = true;
    ++m->la.served;
    return PSGPU_OK;
}

int psgpu_ms_frame_eval(psgpu_ms_model_t *m, int16_t *senscr,
                        const uint8_t *senone_active, int32_t n_senone_active,
                        const float *feat, int32_t compallsen)
{
    PSGPU_REQUIRE(m && senscr && feat, "psgpu_ms_frame_eval: NULL argument");
    PSGPU_REQUIRE(compallsen || n_senone_active == 0 || senone_active, "active list missing");
    const MsDev &d = m->d;
    if (m->la.dirty) {                                   // served calls first: their lists are this call's stale ids
        int rc = ms_la_sync(m);
        if (rc != PSGPU_OK) return rc;
    }
    int n_list = 0;
    if (compallsen)
        memset(m->h_active, 1, (size_t)d.n_mgau);
    else {
        memset(m->h_active, 0, (size_t)d.n_mgau);        // ms_mgau.c:238-249
        int sen = 0;
        for (int i = 0; i < n_senone_active; ++i) {
            sen += senone_active[i];
            if (sen >= d.n_sen) {
                psgpu_set_error("active list runs past n_sen (%d >= %d)", sen, d.n_sen);
                return PSGPU_EINVAL;
            }
            m->h_list[n_list++] = (uint16_t)sen;
            m->h_active[m->h_sen2mgau[sen]] = 1;
        }
        if (n_list == 0)
            return PSGPU_OK;                             // nothing listed: senscr untouched
    }
    memcpy(m->h_feat, feat, (size_t)d.veclen * sizeof(float));
    launch_topn(d, m->d_feat, 1, m->d_active, m->list_id, m->list_dist, 0, nullptr, m->stream);
    PSGPU_HIP(hipGetLastError());
    const size_t done_off = ((size_t)d.n_sen * sizeof(int16_t) + 15) / 16 * 16;
    volatile uint32_t *h_done = reinterpret_cast<volatile uint32_t *>(reinterpret_cast<char *>(m->h_out) + done_off);
    uint32_t *d_done = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(m->d_out) + done_off);
    const uint32_t seq = ++m->seq ? m->seq : ++m->seq;
    launch_senone(d, 1, compallsen != 0, n_list, m->d_list, m->list_id, m->list_dist, m->d_out, 0, m->stream,
                  d_done, seq);
    PSGPU_HIP(hipGetLastError());
    {
        bool done = false;
        for (long i = 0; i < 200000000L; ++i) {
            if (*h_done == seq) { done = true; break; }
            __builtin_ia32_pause();
        }
        if (!done) PSGPU_HIP(hipStreamSynchronize(m->stream));
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    if (compallsen)
        memcpy(senscr, m->h_out, (size_t)d.n_sen * sizeof(int16_t));
    else
        for (int i = 0; i < n_list; ++i)                 // only listed senones are written (ms_mgau.c:252-277)
            senscr[m->h_list[i]] = m->h_out[i];
    return PSGPU_OK;
}

static int ms_batch(psgpu_ms_model_t *m, const float *feats_dev, int32_t total_frames,
                    int32_t *list_id_dev, float *list_dist_dev, int16_t *senscr_dev,
                    void *stream, bool raw)
{
    PSGPU_REQUIRE(m && feats_dev, "psgpu_ms_score_batch_dev: NULL argument");
    PSGPU_REQUIRE((list_id_dev == nullptr) == (list_dist_dev == nullptr), "list buffers: both or neither");
    PSGPU_REQUIRE(total_frames >= 0, "negative frame count");
    if (total_frames == 0) return PSGPU_OK;
    const MsDev &d = m->d;
    hipStream_t st = (hipStream_t)stream;
    PSGPU_HIP(hipMemsetAsync(m->d_flag, 0, sizeof(int32_t), st));
    static const int no_cont = [] { const char *e = getenv("PSGPU_MS_NO_FUSED"); return e ? atoi(e) : 0; }();
    if (m->cont && senscr_dev && !no_cont && (d.featlen[0] == 13 || d.featlen[0] == 39)) {
        if (total_frames > m->cap_best) {
            PSGPU_HIP(hipFree(m->d_best)); m->d_best = nullptr; m->cap_best = 0;
            PSGPU_HIP(hipMalloc((void **)&m->d_best, sizeof(int32_t) * (size_t)total_frames));
            m->cap_best = total_frames;
        }
        PSGPU_HIP(hipMemsetAsync(m->d_best, 0x7f, sizeof(int32_t) * (size_t)total_frames, st));
        launch_cont(d, feats_dev, total_frames, list_id_dev, list_dist_dev, senscr_dev, d.n_sen, m->d_best, m->d_flag, st, raw);
        PSGPU_HIP(hipGetLastError());
        return PSGPU_OK;
    }
    PSGPU_REQUIRE(list_id_dev != nullptr, "this model shape needs the list buffers (only fully continuous models "
                  "are scored without them)");
    static const int no_lane = [] { const char *e = getenv("PSGPU_NO_LANE_KERNEL"); return e ? atoi(e) : 0; }();
    if (no_lane || !launch_lane(d, feats_dev, total_frames, list_id_dev, list_dist_dev, m->d_flag, st))
        launch_topn(d, feats_dev, total_frames, nullptr, list_id_dev, list_dist_dev, 1, m->d_flag, st);
    PSGPU_HIP(hipGetLastError());
    if (senscr_dev) {
        launch_senone(d, total_frames, 1, 0, nullptr, list_id_dev, list_dist_dev, senscr_dev, d.n_sen, st, nullptr, 0, raw);
        PSGPU_HIP(hipGetLastError());
    }
    return PSGPU_OK;
}

int psgpu_ms_score_batch_dev(psgpu_ms_model_t *m, const float *feats_dev, int32_t total_frames,
                             int32_t *list_id_dev, float *list_dist_dev, int16_t *senscr_dev,
                             void *stream)
{
    return ms_batch(m, feats_dev, total_frames, list_id_dev, list_dist_dev, senscr_dev, stream, false);
}

int psgpu_ms_score_batch_raw_dev(psgpu_ms_model_t *m, const float *feats_dev, int32_t total_frames,
                                 int32_t *list_id_dev, float *list_dist_dev, int16_t *senscr_dev, void *stream)
{
    return ms_batch(m, feats_dev, total_frames, list_id_dev, list_dist_dev, senscr_dev, stream, true);
}

int32_t psgpu_ms_batch_needs_lists(const psgpu_ms_model_t *m)
{
    static const int no_cont = [] { const char *e = getenv("PSGPU_MS_NO_FUSED"); return e ? atoi(e) : 0; }();
    return (m && m->cont && !no_cont && (m->d.featlen[0] == 13 || m->d.featlen[0] == 39)) ? 0 : 1;
}
int32_t psgpu_ms_list_entries_per_frame(const psgpu_ms_model_t *m) { return m ? m->d.n_mgau * m->d.n_feat * m->d.topn : 0; }

int psgpu_ms_batch_check(psgpu_ms_model_t *m, void *stream)
{
    PSGPU_REQUIRE(m != nullptr, "psgpu_ms_batch_check: NULL model");
    int32_t flag = 0;
    PSGPU_HIP(hipMemcpyAsync(&flag, m->d_flag, sizeof flag, hipMemcpyDeviceToHost, (hipStream_t)stream));
    PSGPU_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (flag) {
        psgpu_set_error("a frame had fewer than topn densities above WORST_DIST: the reference then reuses "
                        "list ids of the previous call (ms_gauden.c:438-440), which only the per-call "
                        "entry psgpu_ms_frame_eval reproduces");
        return PSGPU_ESTATE;
    }
    return PSGPU_OK;
}

int psgpu_ms_score_batch(psgpu_ms_model_t *m, const float *feats, int32_t total_frames, int16_t *senscr)
{
    PSGPU_REQUIRE(m && feats && senscr && total_frames >= 0, "psgpu_ms_score_batch: bad argument");
    if (total_frames == 0) return PSGPU_OK;
    const MsDev &d = m->d;
    const size_t nl = (size_t)total_frames * d.n_mgau * d.n_feat * d.topn;
    float *df = nullptr, *dd = nullptr; int32_t *di = nullptr; int16_t *ds = nullptr;
    auto cleanup = [&]() { hipFree(df); hipFree(dd); hipFree(di); hipFree(ds); };
#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) {                 \
        psgpu_set_error("%s -> %s", #call, hipGetErrorString(e_)); cleanup();          \
        return e_ == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP; } } while (0)
    TRY(hipMalloc((void **)&df, (size_t)total_frames * d.veclen * sizeof(float)));
    TRY(hipMalloc((void **)&dd, nl * sizeof(float)));
    TRY(hipMalloc((void **)&di, nl * sizeof(int32_t)));
    TRY(hipMalloc((void **)&ds, (size_t)total_frames * d.n_sen * sizeof(int16_t)));
    TRY(hipMemcpy(df, feats, (size_t)total_frames * d.veclen * sizeof(float), hipMemcpyHostToDevice));
    int rc = psgpu_ms_score_batch_dev(m, df, total_frames, di, dd, ds, m->stream);
    if (rc == PSGPU_OK) rc = psgpu_ms_batch_check(m, m->stream);
    if (rc == PSGPU_OK)
        TRY(hipMemcpy(senscr, ds, (size_t)total_frames * d.n_sen * sizeof(int16_t), hipMemcpyDeviceToHost));
#undef TRY
    cleanup();
    return rc;
}

}  // extern "C"
