// psgpu_ptm_dev.h -- model view + wavefront helpers shared by the PTM kernels
// (batched chain/senone kernels in psgpu_ptm.hip, per-call frame_eval kernels
// in psgpu_ptm_frame.hip).
#pragma once
#include "psgpu_internal.h"
#include <list>
#include <mutex>

// Scratch of one batched scoring call (the hand-over between its kernels).  One per stream the model is used on: calls on
// one stream are ordered on the device and may share it; calls from different host threads on different streams run
// concurrently and must not.
struct PtmWorkspace {
    hipStream_t stream;
    uint8_t *open_flags;          // [n_chain][frames]: entries the lane kernel left to the fix-up
    int32_t *fix_list;            // [frames * n_chain] open entries + 1 counter word
    size_t flags_cap;
    int count_dirty;              // the open-entry counters must be zeroed before the next lane pass
    hipStream_t aux;              // a batch scored in ranges: the senone passes' stream, and the events that order the two
    hipEvent_t ev[17];
};

struct psgpu_ptm_model_s {
    int32_t n_mgau, n_feat, n_density, n_sen, topn, ds_ratio, veclen, n_chain;
    int32_t featlen[16];
    int32_t featoff[16];
    int32_t uniform_len;          // featlen if all streams are equal, else 0
    int32_t fast_shape;           // 128 densities, top-4, 13-dim streams: the batched kernels' shape
    int device;
    float *mean, *var, *det;      // device
    uint8_t *mixw, *sen2cb, *logadd8;
    uint8_t *h_sen2cb;            // host mirror (active list -> codebook set, ptm_mgau.c:297-321)
    uint8_t *mixw_slot;           // [n_feat][n_density][slot_stride], slot order, rows 64-byte aligned
    uint8_t *mixw_sen;            // [n_sen][n_feat][dens_stride]: a senone's weights side by side, for kernels that score single senones
    uint8_t *group_cb;            // [n_groups] codebook of each 4-slot group
    uint16_t *slot_sen;           // [n_slots] senone id of a slot, 0xffff = pad
    int32_t slot_stride, n_groups;
    int32_t logadd8_size;
    int32_t la_max;               // largest entry of the log-add table (bounds of the senone kernel's biased form)
    std::mutex ws_mu;             // guards the list below (not the workspaces: those belong to their stream)
    std::list<PtmWorkspace> ws;
    hipEvent_t ev[4];             // optional per-kernel timing: lane | fix-up | (gap) | senone (one timed caller at a time)
    int timing;
    bool sen_timed = false;       // ev[3] has been recorded (psgpu_ptm_score_batch_dev ran with timing on)
};

// the workspace of `st` (created on first use; list nodes never move)
static inline PtmWorkspace *ptm_workspace(psgpu_ptm_model_t *m, hipStream_t st, bool create)
{
    std::lock_guard<std::mutex> lock(m->ws_mu);
    for (PtmWorkspace &w : m->ws) if (w.stream == st) return &w;
    if (!create) return nullptr;
    m->ws.push_back(PtmWorkspace{st, nullptr, nullptr, 0, 1, nullptr, {}});
    return &m->ws.back();
}

struct PtmDev {
    const float *mean, *var, *det;
    const uint8_t *mixw, *sen2cb, *logadd8, *mixw_slot, *group_cb;
    const uint16_t *slot_sen;
    int32_t slot_stride, n_groups;
    int32_t n_mgau, n_feat, n_density, n_sen, veclen, n_chain, ds_ratio, logadd8_size, topn;
    int32_t featlen[16], featoff[16];
};

static inline PtmDev dev_view(const psgpu_ptm_model_t *m)
{
    PtmDev p;
    p.mean = m->mean; p.var = m->var; p.det = m->det;
    p.mixw = m->mixw; p.sen2cb = m->sen2cb; p.logadd8 = m->logadd8;
    p.mixw_slot = m->mixw_slot; p.group_cb = m->group_cb; p.slot_sen = m->slot_sen;
    p.slot_stride = m->slot_stride; p.n_groups = m->n_groups;
    p.n_mgau = m->n_mgau; p.n_feat = m->n_feat; p.n_density = m->n_density;
    p.n_sen = m->n_sen; p.veclen = m->veclen; p.n_chain = m->n_chain;
    p.ds_ratio = m->ds_ratio; p.logadd8_size = m->logadd8_size; p.topn = m->topn;
    for (int f = 0; f < 16; ++f) { p.featlen[f] = m->featlen[f]; p.featoff[f] = m->featoff[f]; }
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

// max over the 64 lanes of a wavefront, returned wave-uniform.  Runs entirely
// on the DPP path (no LDS traffic): quad butterflies, row rotations, then the
// two cross-row broadcasts; lane 63 ends up with the wave maximum.  The s_nop
// fill the 2 wait states gfx9 requires between a VALU write and a DPP read of
// the same VGPR (the compiler does not see inside the asm statement).
__device__ __forceinline__ int32_t wave_max_i32(int32_t v)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// Wave-uniform top-N list.
template <int N>
struct TopN {
    int32_t cw[N];
    int32_t sc[N];
};

// Exact emulation of one reference frame step on a chain whose 128 distances
// are d0 (codeword = lane) and d1 (codeword = lane + 64):
//   eval_topn  (ptm_mgau.c:71-136)  re-score the carried list, stable
//              descending insertion sort with strict '>'
//   eval_cb    (ptm_mgau.c:140-226) scan codewords in index order against
//              the moving float threshold, skip-if-present, insert ahead of
//              equal scores, worst entry drops
// All list state is wave-uniform.  This is the slow path, taken only when the
// closed form below cannot be used.
template <int N>
__device__ __forceinline__ void exact_frame_step(TopN<N> &L, float d0, float d1, int lane, bool scan)
{
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
    if (!scan)
        return;
    int pos = 0;                            // next codeword index to look at
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
        const int c = b0 ? (__ffsll((long long)b0) - 1) : (64 + __ffsll((long long)b1) - 1);
        const float d = (c < 64) ? lane_value(d0, c) : lane_value(d1, c - 64);
        const int32_t s = dist_to_int(d);
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


constexpr int32_t kKeyLo = -(1 << 24);          // clamp range of the score part
constexpr int32_t kKeyHi = (1 << 24) - 1;       // of a packed selection key

// Closed form of one frame step on a 128-codeword chain (2 codewords per
// lane): if the four largest truncated scores of the codebook are pairwise
// distinct and strictly above the fifth, the reference's seed/scan procedure
// (eval_topn + eval_cb, ptm_mgau.c:87-226) ends with exactly those four in
// descending order whatever the seeds were (DESIGN.md "top-N closed form").
// Selection key = clamp(trunc(d), -2^24, 2^24-1) << 7 | (127 - codeword):
// unique per codeword, so four wave-max rounds extract the winners.  tag0 =
// 127 - lane.  Returns false (list untouched) if a winner touches the clamp
// bounds or on any tie.
__device__ __forceinline__ bool closed_form_top4(TopN<4> &L, float d0, float d1, int32_t tag0)
{
    constexpr int N = 4;
    const float c0 = __builtin_amdgcn_fmed3f(d0, (float)kKeyLo, (float)kKeyHi);
    const float c1 = __builtin_amdgcn_fmed3f(d1, (float)kKeyLo, (float)kKeyHi);
    const int32_t k0 = ((int32_t)c0 << 7) | tag0;
    const int32_t k1 = (((int32_t)c1 << 7) | tag0) - 64;      // tag1 = tag0 - 64
    int32_t hi = max(k0, k1), lo = min(k0, k1);
    int32_t key[N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
        key[r] = wave_max_i32(hi);
        const bool win = (hi == key[r]);
        hi = win ? lo : hi;
        lo = win ? kMaxNegInt32 : lo;
    }
    const int32_t s4 = key[N - 1] >> 7;
    // a remaining codeword with the 4th winner's score, or a tie among
    // the winners, or a clamped winner -> not closed
    bool bad = __ballot(((hi >> 7) == s4) | ((lo >> 7) == s4)) != 0;
    bad |= ((key[0] >> 7) >= kKeyHi) | (s4 <= kKeyLo);
#pragma unroll
    for (int r = 1; r < N; ++r) bad |= ((key[r] >> 7) == (key[r - 1] >> 7));
    if (bad) return false;
#pragma unroll
    for (int r = 0; r < N; ++r) {
        L.sc[r] = key[r] >> 7;
        L.cw[r] = 127 - (key[r] & 127);
    }
    return true;
}

constexpr int kGenK = 4;                        // codewords per lane of the any-shape path

__device__ __forceinline__ float pick4(const float (&d)[kGenK], int c)
{
    const int k = c >> 6, l = c & 63;
    float v = lane_value(d[0], l);
    if (k == 1) v = lane_value(d[1], l);
    if (k == 2) v = lane_value(d[2], l);
    if (k == 3) v = lane_value(d[3], l);
    return v;
}

// Any-shape exact frame step (up to 256 codewords, 4 per lane: cw = k*64 + lane).
// SEMI = true: eval_topn (s2_semi_mgau.c:69-109) + eval_cb (:111-170).
// SEMI = false: the PTM pair (ptm_mgau.c:87-226), whose acceptance test is the
// finished float distance alone (pass dp = d).  Wave-uniform list state.  d[k] / dp[k] = finished distance / partial sum before the last
// dimension of codeword k*64 + lane.  A codeword is accepted iff every float
// guard `d >= worst->score` passed (<=> dp >= (float)worst) AND the truncated
// finished distance is not below worst (`d_int < worst->score`), it is not in
// the list, and it goes ahead of equal scores.
template <int N, bool SEMI>
__device__ __forceinline__ void generic_frame_step(TopN<N> &L, const float (&d)[kGenK], const float (&dp)[kGenK],
                                                int lane, int n_density, bool scan)
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        L.sc[i] = dist_to_int(pick4(d, L.cw[i]));
#pragma unroll
        for (int j = i; j > 0; --j) {
            if (L.sc[j] > L.sc[j - 1]) {
                int32_t ts = L.sc[j]; L.sc[j] = L.sc[j - 1]; L.sc[j - 1] = ts;
                int32_t tc = L.cw[j]; L.cw[j] = L.cw[j - 1]; L.cw[j - 1] = tc;
            }
        }
    }
    if (!scan)
        return;
    int32_t di[kGenK];
#pragma unroll
    for (int k = 0; k < kGenK; ++k) di[k] = dist_to_int(d[k]);
    int pos = 0;
    for (;;) {
        const int32_t W = L.sc[N - 1];
        const float th = (float)W;
        int found = -1;
#pragma unroll
        for (int k = 0; k < kGenK; ++k) {
            const int cw = k * 64 + lane;
            bool inl = false;
#pragma unroll
            for (int i = 0; i < N; ++i) inl |= (L.cw[i] == cw);
            const bool ok = (cw < n_density) && (cw >= pos) && (dp[k] >= th) && (!SEMI || di[k] >= W) && !inl;
            const unsigned long long b = __ballot(ok);
            if (found < 0 && b) found = k * 64 + __ffsll((long long)b) - 1;
        }
        if (found < 0)
            break;
        const int32_t s = dist_to_int(pick4(d, found));
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
        for (int k = 0; k < N; ++k)
            if (q == k) { L.sc[k] = s; L.cw[k] = found; }
        pos = found + 1;
    }
}

