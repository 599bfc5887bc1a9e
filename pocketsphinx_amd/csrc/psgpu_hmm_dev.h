// psgpu_hmm_dev.h -- device functions of the Viterbi step, shared by the kernels that
// evaluate HMMs (psgpu_hmm.hip: the batched step and the phone loop; psgpu_search.hip: the
// lexicon-tree search).  Restates hmm_vit_eval and its variants (reference src/hmm.c:222-805).
#pragma once
#include "psgpu_internal.h"
#include <climits>

constexpr int32_t kW = kWorstScore;
constexpr int kTmatWorst = 255;            // tmat.h: 8-bit floor; "tp > -255" gates skip arcs
constexpr uint16_t kBadSsid = 0xffff;      // hmm.h:89

__device__ __forceinline__ int32_t clampw(int32_t v) { return v < kW ? kW : v; }

struct HmmRegs {
    int32_t score[5], history[5], out_score, out_history, bestscore;
    uint16_t senid[5];
};

// The frame's senone scores as the Viterbi step reads them: `ss[senone]`.  A plain `const int16_t *` row, or a row of
// un-normalised scores with the active list's normaliser applied on the fly (ptm_mgau.c:393-400: score - best in int16
// arithmetic).
struct SenRowNorm {
    const int16_t *row;
    int32_t nb;
    // score - normaliser as the scorers store it: ptm_mgau_senone_eval subtracts in int16 (ptm_mgau.c:398-400; its sums are below
    // 2^11, nothing wraps), ms_cont_mgau_frame_eval clamps the difference of two int16 values to int16 (ms_mgau.c:226-234, :269-277)
    // -- one expression serves both: the clamp never acts on a PTM score
    __device__ __forceinline__ int16_t operator[](int s) const
    {
        const int32_t x = (int32_t)((uint32_t)(int32_t)row[s] - (uint32_t)nb);
        return (int16_t)(x > 32767 ? 32767 : (x < -32768 ? -32768 : x));
    }
};

// ---- 3-state, non-multiplex (hmm.c:529-607) --------------------------------
template <typename S>
__device__ __forceinline__ int32_t vit3(HmmRegs &h, const uint8_t *tp, const S &ss)
{
#define TP(i, j) (-(int32_t)tp[(i) * 4 + (j)])
    int32_t s2 = h.score[2] - ss[h.senid[2]];
    int32_t s1 = h.score[1] - ss[h.senid[1]];
    int32_t s0 = h.score[0] - ss[h.senid[0]];
    int32_t best = kW, t0, t1, t2 = INT_MIN;
    if (s1 > kW) {
        t1 = s2 + TP(2, 3);
        if (TP(1, 3) > -kTmatWorst) t2 = s1 + TP(1, 3);
        int32_t s3;
        if (t1 > t2) { s3 = t1; h.out_history = h.history[2]; }
        else         { s3 = t2; h.out_history = h.history[1]; }
        s3 = clampw(s3);
        h.out_score = s3;
        best = s3;
    }
    t0 = s2 + TP(2, 2);
    t1 = s1 + TP(1, 2);
    if (TP(0, 2) > -kTmatWorst) t2 = s0 + TP(0, 2);     // else t2 keeps its value (stale, as the reference)
    if (t0 > t1) {
        if (t2 > t0) { s2 = t2; h.history[2] = h.history[0]; }
        else s2 = t0;
    }
    else {
        if (t2 > t1) { s2 = t2; h.history[2] = h.history[0]; }
        else { s2 = t1; h.history[2] = h.history[1]; }
    }
    s2 = clampw(s2);
    best = max(best, s2);
    h.score[2] = s2;
    t0 = s1 + TP(1, 1);
    t1 = s0 + TP(0, 1);
    if (t0 > t1) s1 = t0;
    else { s1 = t1; h.history[1] = h.history[0]; }
    s1 = clampw(s1);
    best = max(best, s1);
    h.score[1] = s1;
    s0 = clampw(s0 + TP(0, 0));
    best = max(best, s0);
    h.score[0] = s0;
    h.bestscore = best;
    return best;
#undef TP
}

// ---- 3-state, multiplex (hmm.c:609-707) -------------------------------------
template <typename S>
__device__ __forceinline__ int32_t vit3_mpx(HmmRegs &h, const uint8_t *tp, const S &ss, const uint16_t *sseq)
{
#define TP(i, j) (-(int32_t)tp[(i) * 4 + (j)])
    // the states' senone ids first, all three with nothing between them: one trip to the sseq table instead of three (each SEN()
    // below sat behind the test of its state's ssid, so the compiler waited for one before it asked for the next).  A state
    // without an ssid reads entry 0 and drops it.
    const uint16_t sq_[3] = { sseq[(size_t)h.senid[0] * 3 + 0],
                              sseq[(size_t)(h.senid[1] == kBadSsid ? 0 : h.senid[1]) * 3 + 1],
                              sseq[(size_t)(h.senid[2] == kBadSsid ? 0 : h.senid[2]) * 3 + 2] };
#define SEN(st) (-(int32_t)ss[sq_[st]])
    int32_t s3, s2, s1, s0, t0, t1, t2 = INT_MIN, best;
    if (h.senid[2] == kBadSsid) s2 = t1 = kW;
    else { s2 = h.score[2] + SEN(2); t1 = s2 + TP(2, 3); }
    if (h.senid[1] == kBadSsid) s1 = t2 = kW;
    else {
        s1 = h.score[1] + SEN(1);
        if (TP(1, 3) > -kTmatWorst) t2 = s1 + TP(1, 3);
    }
    if (t1 > t2) { s3 = t1; h.out_history = h.history[2]; }
    else         { s3 = t2; h.out_history = h.history[1]; }
    s3 = clampw(s3);
    h.out_score = s3;
    best = s3;

    s0 = h.score[0] + SEN(0);
    t0 = t1 = kW;
    if (s2 != kW) t0 = s2 + TP(2, 2);
    if (s1 != kW) t1 = s1 + TP(1, 2);
    if (TP(0, 2) > -kTmatWorst) t2 = s0 + TP(0, 2);
    if (t0 > t1) {
        if (t2 > t0) { s2 = t2; h.history[2] = h.history[0]; h.senid[2] = h.senid[0]; }
        else s2 = t0;
    }
    else {
        if (t2 > t1) { s2 = t2; h.history[2] = h.history[0]; h.senid[2] = h.senid[0]; }
        else { s2 = t1; h.history[2] = h.history[1]; h.senid[2] = h.senid[1]; }
    }
    s2 = clampw(s2);
    best = max(best, s2);
    h.score[2] = s2;

    t0 = kW;
    if (s1 != kW) t0 = s1 + TP(1, 1);
    t1 = s0 + TP(0, 1);
    if (t0 > t1) s1 = t0;
    else { s1 = t1; h.history[1] = h.history[0]; h.senid[1] = h.senid[0]; }
    s1 = clampw(s1);
    best = max(best, s1);
    h.score[1] = s1;

    s0 = clampw(s0 + TP(0, 0));
    best = max(best, s0);
    h.score[0] = s0;
    h.bestscore = best;
    return best;
#undef TP
#undef SEN
}

// three-way arg-max of the 5-state forms: self loop T0, neighbour T1 (state
// nb), skip T2 (state sk); destination state nb + 1
template <bool MPX>
__device__ __forceinline__ int32_t pick3(HmmRegs &h, int32_t T0, int32_t T1, int32_t T2, int nb, int sk)
{
    int32_t dst;
    if (T0 > T1) {
        if (T2 > T0) { dst = T2; h.history[nb + 1] = h.history[sk]; if (MPX) h.senid[nb + 1] = h.senid[sk]; }
        else dst = T0;
    }
    else {
        if (T2 > T1) { dst = T2; h.history[nb + 1] = h.history[sk]; if (MPX) h.senid[nb + 1] = h.senid[sk]; }
        else { dst = T1; h.history[nb + 1] = h.history[nb]; if (MPX) h.senid[nb + 1] = h.senid[nb]; }
    }
    return dst;
}

// ---- 5-state, non-multiplex (hmm.c:222-350) ---------------------------------
template <typename S>
__device__ __forceinline__ int32_t vit5(HmmRegs &h, const uint8_t *tp, const S &ss)
{
#define TP(i, j) (-(int32_t)tp[(i) * 6 + (j)])
#define SEN(st) (-(int32_t)ss[h.senid[st]])
    int32_t s5, s4, s3, s2, s1, s0, t0, t1, t2, best = kW;
    s4 = h.score[4] + SEN(4);
    s3 = h.score[3] + SEN(3);
    if (s3 > kW) {
        t1 = s4 + TP(4, 5);
        t2 = s3 + TP(3, 5);
        if (t1 > t2) { s5 = t1; h.out_history = h.history[4]; }
        else         { s5 = t2; h.out_history = h.history[3]; }
        s5 = clampw(s5);
        h.out_score = s5;
        best = s5;
    }
    s2 = h.score[2] + SEN(2);
    if (s2 > kW) {
        t0 = s4 + TP(4, 4); t1 = s3 + TP(3, 4); t2 = s2 + TP(2, 4);
        s4 = clampw(pick3<false>(h, t0, t1, t2, 3, 2));
        best = max(best, s4);
        h.score[4] = s4;
    }
    s1 = h.score[1] + SEN(1);
    if (s1 > kW) {
        t0 = s3 + TP(3, 3); t1 = s2 + TP(2, 3); t2 = s1 + TP(1, 3);
        s3 = clampw(pick3<false>(h, t0, t1, t2, 2, 1));
        best = max(best, s3);
        h.score[3] = s3;
    }
    s0 = h.score[0] + SEN(0);
    t0 = s2 + TP(2, 2); t1 = s1 + TP(1, 2); t2 = s0 + TP(0, 2);
    s2 = clampw(pick3<false>(h, t0, t1, t2, 1, 0));
    best = max(best, s2);
    h.score[2] = s2;

    t0 = s1 + TP(1, 1); t1 = s0 + TP(0, 1);
    if (t0 > t1) s1 = t0;
    else { s1 = t1; h.history[1] = h.history[0]; }
    s1 = clampw(s1);
    best = max(best, s1);
    h.score[1] = s1;

    s0 = clampw(s0 + TP(0, 0));
    best = max(best, s0);
    h.score[0] = s0;
    h.bestscore = best;
    return best;
#undef TP
#undef SEN
}

// ---- 5-state, multiplex (hmm.c:355-525) -------------------------------------
template <typename S>
__device__ __forceinline__ int32_t vit5_mpx(HmmRegs &h, const uint8_t *tp, const S &ss, const uint16_t *sseq)
{
#define TP(i, j) (-(int32_t)tp[(i) * 6 + (j)])
    // (as in vit3_mpx: the five senone ids in one trip)
    const uint16_t sq_[5] = { sseq[(size_t)h.senid[0] * 5 + 0],
                              sseq[(size_t)(h.senid[1] == kBadSsid ? 0 : h.senid[1]) * 5 + 1],
                              sseq[(size_t)(h.senid[2] == kBadSsid ? 0 : h.senid[2]) * 5 + 2],
                              sseq[(size_t)(h.senid[3] == kBadSsid ? 0 : h.senid[3]) * 5 + 3],
                              sseq[(size_t)(h.senid[4] == kBadSsid ? 0 : h.senid[4]) * 5 + 4] };
#define SEN(st) (-(int32_t)ss[sq_[st]])
    int32_t s5, s4, s3, s2, s1, s0, t0, t1, t2, best;
    if (h.senid[4] == kBadSsid) s4 = t1 = kW;
    else { s4 = h.score[4] + SEN(4); t1 = s4 + TP(4, 5); }
    if (h.senid[3] == kBadSsid) s3 = t2 = kW;
    else { s3 = h.score[3] + SEN(3); t2 = s3 + TP(3, 5); }
    if (t1 > t2) { s5 = t1; h.out_history = h.history[4]; }
    else         { s5 = t2; h.out_history = h.history[3]; }
    s5 = clampw(s5);
    h.out_score = s5;
    best = s5;

    if (h.senid[2] == kBadSsid) s2 = t2 = kW;
    else { s2 = h.score[2] + SEN(2); t2 = s2 + TP(2, 4); }
    t0 = t1 = kW;
    if (s4 != kW) t0 = s4 + TP(4, 4);
    if (s3 != kW) t1 = s3 + TP(3, 4);
    s4 = clampw(pick3<true>(h, t0, t1, t2, 3, 2));
    best = max(best, s4);
    h.score[4] = s4;

    if (h.senid[1] == kBadSsid) s1 = t2 = kW;
    else { s1 = h.score[1] + SEN(1); t2 = s1 + TP(1, 3); }
    t0 = t1 = kW;
    if (s3 != kW) t0 = s3 + TP(3, 3);
    if (s2 != kW) t1 = s2 + TP(2, 3);
    s3 = clampw(pick3<true>(h, t0, t1, t2, 2, 1));
    best = max(best, s3);
    h.score[3] = s3;

    s0 = h.score[0] + SEN(0);
    t0 = t1 = kW;
    if (s2 != kW) t0 = s2 + TP(2, 2);
    if (s1 != kW) t1 = s1 + TP(1, 2);
    t2 = s0 + TP(0, 2);
    s2 = clampw(pick3<true>(h, t0, t1, t2, 1, 0));
    best = max(best, s2);
    h.score[2] = s2;

    t0 = kW;
    if (s1 != kW) t0 = s1 + TP(1, 1);
    t1 = s0 + TP(0, 1);
    if (t0 > t1) s1 = t0;
    else { s1 = t1; h.history[1] = h.history[0]; h.senid[1] = h.senid[0]; }
    s1 = clampw(s1);
    best = max(best, s1);
    h.score[1] = s1;

    s0 = clampw(s0 + TP(0, 0));
    best = max(best, s0);
    h.score[0] = s0;
    h.bestscore = best;
    return best;
#undef TP
#undef SEN
}

// ---------------------------------------------------------------------------
// kernel: one lane per active HMM
// ---------------------------------------------------------------------------

// ---- any topology (hmm.c:710-784) -------------------------------------------
// hmm_vit_eval_anytopo: what hmm_vit_eval (hmm.c:786-805) runs for anything but 3 or 5 emitting states, i.e. 1, 2 or 4
// (HMM_MAX_NSTATE is 5).  Any upper-triangular transition matrix.  Differences from the hard-wired forms, all kept:
// only the incoming sums of states 1.. are clamped at WORST_SCORE (state 0's is not); new scores are not clamped; a state
// whose self loop wins (or that nothing reaches) keeps its history and, multiplexed, its ssid.
template <int NE, typename S>
__device__ __forceinline__ int32_t vit_any(HmmRegs &h, const uint8_t *tp, const S &ss, const uint16_t *sseq, bool mpx)
{
#define TP(i, j) (-(int32_t)tp[(i) * (NE + 1) + (j)])
    int32_t st[NE];
#pragma unroll
    for (int from = 0; from < NE; ++from) {
        int32_t sen;                                    // hmm_senscr (hmm.h:207-209)
        if (h.senid[from] == kBadSsid) sen = kW;
        else sen = -(int32_t)ss[mpx ? sseq[(size_t)h.senid[from] * NE + from] : h.senid[from]];
        st[from] = h.score[from] + sen;
        if (from > 0 && st[from] < kW) st[from] = kW;
    }
    int32_t scr = kW, bestscr;
    int bestfrom = -1;
#pragma unroll
    for (int from = NE - 1; from >= 0; --from) {        // the final state: no self transition
        const int32_t t = TP(from, NE);
        if (t > -kTmatWorst && st[from] + t > scr) { scr = st[from] + t; bestfrom = from; }
    }
    h.out_score = scr;
#pragma unroll
    for (int from = 0; from < NE; ++from) if (bestfrom == from) h.out_history = h.history[from];
    bestscr = scr;
#pragma unroll
    for (int to = NE - 1; to >= 0; --to) {
        scr = TP(to, to) > -kTmatWorst ? st[to] + TP(to, to) : kW;
        bestfrom = -1;
#pragma unroll
        for (int from = to - 1; from >= 0; --from) {
            const int32_t t = TP(from, to);
            if (t > -kTmatWorst && st[from] + t > scr) { scr = st[from] + t; bestfrom = from; }
        }
        h.score[to] = scr;
        // (states below `to` still hold their old history / ssid: the sweep goes downwards)
#pragma unroll
        for (int from = 0; from < NE; ++from)
            if (from < to && bestfrom == from) { h.history[to] = h.history[from]; if (mpx) h.senid[to] = h.senid[from]; }
        if (bestscr < scr) bestscr = scr;
    }
    h.bestscore = bestscr;
    return bestscr;
#undef TP
}
