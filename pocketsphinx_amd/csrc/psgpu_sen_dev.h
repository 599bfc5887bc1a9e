// psgpu_sen_dev.h -- device functions for scoring single senones from a frame's top-N lists, shared by the kernels that score
// on demand instead of reading full score rows (psgpu_search.hip: the lexicon-tree search; psgpu_hmm.hip: the phone loop's
// preparation).  Restates ptm_mgau_codebook_norm (reference src/ptm_mgau.c:265-295) and ptm_mgau_senone_eval (:326-403) for
// the shape the batched scorer's lists have: 3 feature streams, top-4 codewords, lists chain-major
// ([chain = codebook * 3 + stream][frame], psgpu_ptm_score_batch_dev).
#pragma once
#include "psgpu_internal.h"

constexpr int kSenStreams = 3, kSenTopn = 4;
constexpr int32_t kSenMaxNegAscr = 96;      // MAX_NEG_ASCR, ptm_mgau.h
constexpr int kSenShift = 10;               // SENSCR_SHIFT

struct SenModel {
    const uint8_t *mixw;                    // [n_sen][stream][dens_stride] mixture weights, senone-major (psgpu_ptm_view_t.mixw_sen): a
                                            // senone's twelve weights lie in three cache lines, not twelve
    const uint8_t *sen2cb;                  // [n_sen]
    int32_t n_sen, n_density;               // n_density: dens_stride = n_density rounded up to 64
};

// (:280-291) one chain's four raw scores against its stream's normaliser: -(score >> 10 - norm), capped, packed as bytes
__device__ __forceinline__ uint32_t sen_pack_scores(int32_t s0, int32_t s1, int32_t s2, int32_t s3, int32_t norm)
{
    const int32_t a = min(kSenMaxNegAscr, -((s0 >> kSenShift) - norm)), b = min(kSenMaxNegAscr, -((s1 >> kSenShift) - norm));
    const int32_t c = min(kSenMaxNegAscr, -((s2 >> kSenShift) - norm)), d = min(kSenMaxNegAscr, -((s3 >> kSenShift) - norm));
    return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
}

// (:326-403) one senone: the sum over the streams of the log-sum over the top-N codewords of (weight + score).  l_cw / l_sc:
// the frame's lists per chain, packed four to a word (LDS); la: the 8-bit log-add table, readable up to index 511 (zero
// beyond the reference's 256 entries -- fast_logmath_add, tied_mgau_common.h:106-125, min(x, y) - T[|x - y|]).
// (_cb: the senone's codebook given by the caller, who has the map closer than device memory)
__device__ __forceinline__ int32_t sen_eval_f3n4_cb(const SenModel &m, const uint32_t *l_cw, const uint32_t *l_sc, const uint8_t *la, int sen, int cb)
{
    uint32_t w[kSenStreams][kSenTopn], nsc[kSenStreams];
#pragma unroll
    for (int f = 0; f < kSenStreams; ++f) {                  // all twelve weights are asked for before the first is used
        const uint32_t c4 = l_cw[cb * kSenStreams + f];
        nsc[f] = l_sc[cb * kSenStreams + f];
        const int ds = (m.n_density + 63) & ~63;
        const uint8_t *row = m.mixw + ((size_t)sen * kSenStreams + f) * ds;
#pragma unroll
        for (int k = 0; k < kSenTopn; ++k) w[f][k] = row[(c4 >> (8 * k)) & 0xff];
    }
    int32_t fden[kSenStreams];
#pragma unroll
    for (int f = 0; f < kSenStreams; ++f) fden[f] = (int32_t)w[f][0] + (int32_t)(nsc[f] & 0xff);
#pragma unroll
    for (int k = 1; k < kSenTopn; ++k) {                     // the three streams' chains advance together
        int32_t lo[kSenStreams], dd[kSenStreams];
#pragma unroll
        for (int f = 0; f < kSenStreams; ++f) {
            const int32_t y = (int32_t)w[f][k] + (int32_t)((nsc[f] >> (8 * k)) & 0xff);
            lo[f] = min(fden[f], y);
            dd[f] = max(fden[f], y) - lo[f];
        }
#pragma unroll
        for (int f = 0; f < kSenStreams; ++f) fden[f] = lo[f] - (int32_t)la[dd[f]];
    }
    return fden[0] + fden[1] + fden[2];
}
__device__ __forceinline__ int32_t sen_eval_f3n4(const SenModel &m, const uint32_t *l_cw, const uint32_t *l_sc, const uint8_t *la, int sen)
{
    return sen_eval_f3n4_cb(m, l_cw, l_sc, la, sen, m.sen2cb[sen]);
}
