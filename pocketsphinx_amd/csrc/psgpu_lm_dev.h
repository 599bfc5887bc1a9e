// psgpu_lm_dev.h -- the trigram look-up as device code, shared by the batch look-up kernel
// (psgpu_lm.hip) and the lexicon-tree search (psgpu_search.hip).
//
// ngram_tg_score as the n-gram search calls it (lm/ngram_model.c:451 -> ngram_model_set_score,
// lm/ngram_model_set.c:685 -> ngram_ng_score, ngram_model.c:388 -> ngram_model_trie_score,
// lm/ngram_model_trie.c:710-742 -> lm_trie_score, lm/lm_trie.c:549-828): one model without
// classes; a bit-packed reverse trie searched by interpolation (uniform_find), 16-bit
// quantised probabilities and back-offs.  All integer arithmetic in uint32 with the
// reference's wrap-around, float sums in its order and unfused.
//
// One look-up is a chain of dependent loads (3-6 per trie level); the tables of a small model sit
// in L2, so a look-up costs a few microseconds of latency and no bandwidth to speak of: callers
// issue many look-ups in parallel (one per lane), never a loop over words in one lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "psgpu_internal.h"

struct LmLevel { uint32_t off, total_bits, word_bits, word_mask, max_vocab, next_bits, next_mask; };
struct LmDev {
    int32_t order, n_unigrams, n_words;
    const uint32_t *ug;       // [n_unigrams + 1][3]: prob bits, back-off bits, next
    const uint8_t *mem;       // ngram_mem (4-byte aligned, 8 bytes of padding)
    LmLevel lev[4];
    const float *quant;       // [2 * (order - 2) + 1][65536]
    float lw;
    int32_t log_wip, log_zero;
    const int32_t *widmap;    // [n_words]
    // word classes (ngram_ng_score's "declassify", lm/ngram_model.c:388-417): a class word scores as its class's tag word plus its
    // in-class weight; as a history word it IS the tag word.  widmap carries the tag word's id for it (-1 when ngram_class_prob does not
    // find it: log_zero), histmap (= widmap without classes) what it is as history, cwt (or NULL) the weight
    const int32_t *histmap, *cwt;
    // a model SET without a current model (ngram_model_set_score, lm/ngram_model_set.c:685-727): the look-up is the log-sum over the
    // members of lweights[i] + member i's look-up, through the set's own log-add table (logmath_add, util/logmath.c:401-446)
    int32_t n_set;
    const LmDev *set;         // [n_set] members (device memory)
    const int32_t *set_lw;    // [n_set]
    const uint32_t *addtab;   // logadd_t.table widened to 32 bits
    int32_t addtab_n, add_zero;
};

struct LmRange { uint32_t begin, end; };

// bitarr_read_int25 (lm/bitarr.c:74-82): 32 bits little-endian from the byte holding bit `offset`
__device__ __forceinline__ uint32_t lm_read25(const uint8_t *level, uint32_t offset, uint32_t mask)
{
    // (the aligned address is formed by pointer arithmetic on `level`, not by a round trip through an integer: that
    //  keeps the address space the caller established, psgpu_as_global)
    const uint32_t byte = offset >> 3, mis = (uint32_t)(((uintptr_t)level + byte) & 3);
    const uint32_t *w = reinterpret_cast<const uint32_t *>(level + byte - mis);
    const uint64_t v = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    return (uint32_t)(v >> (8 * mis + (offset & 7))) & mask;
}
__device__ __forceinline__ float lm_ug_prob(const LmDev &m, uint32_t w) { return __uint_as_float(psgpu_as_global(m.ug)[3 * (size_t)w]); }
__device__ __forceinline__ float lm_ug_bo(const LmDev &m, uint32_t w) { return __uint_as_float(psgpu_as_global(m.ug)[3 * (size_t)w + 1]); }
__device__ __forceinline__ void lm_ug_range(const LmDev &m, uint32_t w, LmRange &r)      // unigram_find, lm_trie.c:549
{
    r.begin = psgpu_as_global(m.ug)[3 * (size_t)w + 2]; r.end = psgpu_as_global(m.ug)[3 * (size_t)w + 5];
}

// uniform_find, lm_trie.c:564-600
__device__ __forceinline__ bool lm_uniform_find(const uint8_t *level, uint32_t total_bits, uint32_t key_mask,
                                                uint32_t before_it, uint32_t before_v, uint32_t after_it, uint32_t after_v,
                                                uint32_t key, uint32_t &out)
{
    if (key > after_v) return false;
    while (after_it - before_it > 1) {
        const uint32_t pivot = before_it + (1u + ((key - before_v) * (after_it - before_it - 1u)) / (after_v - before_v + 1u));
        const uint32_t mid = lm_read25(level, pivot * total_bits, key_mask);
        if (mid < key) { before_it = pivot; before_v = mid; }
        else if (mid > key) { after_it = pivot; after_v = mid; }
        else { out = pivot; return true; }
    }
    return false;
}
// middle_find, lm_trie.c:602-631: bit offset just after the word field; r becomes the entry's child range
__device__ __forceinline__ bool lm_middle_find(const LmDev &m, int l, uint32_t word, LmRange &r, uint32_t &o)
{
    const LmLevel &v = m.lev[l];
    const uint8_t *level = psgpu_as_global(m.mem) + v.off;
    uint32_t at;
    if (!lm_uniform_find(level, v.total_bits, v.word_mask, r.begin - 1u, 0u, r.end, v.max_vocab, word, at)) return false;
    at = at * v.total_bits + v.word_bits;
    r.begin = lm_read25(level, at + 32u, v.next_mask);
    r.end = lm_read25(level, at + 32u + v.total_bits, v.next_mask);
    o = at;
    return true;
}
// longest_find, lm_trie.c:633-651
__device__ __forceinline__ bool lm_longest_find(const LmDev &m, uint32_t word, const LmRange &r, uint32_t &o)
{
    const LmLevel &v = m.lev[m.order - 2];
    uint32_t at;
    if (!lm_uniform_find(psgpu_as_global(m.mem) + v.off, v.total_bits, v.word_mask, r.begin - 1u, 0u, r.end, v.max_vocab, word, at)) return false;
    o = at * v.total_bits + v.word_bits;
    return true;
}
// lm_trie_quant_mboread / _mpread / _lpread, lm_trie_quant.c:330-354
__device__ __forceinline__ float lm_mid_bo(const LmDev &m, int l, uint32_t o)
{ return psgpu_as_global(m.quant)[(size_t)(2 * l + 1) * 65536 + lm_read25(psgpu_as_global(m.mem) + m.lev[l].off, o, 0xffffu)]; }
__device__ __forceinline__ float lm_mid_prob(const LmDev &m, int l, uint32_t o)
{ return psgpu_as_global(m.quant)[(size_t)(2 * l) * 65536 + lm_read25(psgpu_as_global(m.mem) + m.lev[l].off, o + 16u, 0xffffu)]; }
__device__ __forceinline__ float lm_long_prob(const LmDev &m, uint32_t o)
{ return psgpu_as_global(m.quant)[(size_t)(2 * (m.order - 2)) * 65536 + lm_read25(psgpu_as_global(m.mem) + m.lev[m.order - 2].off, o, 0xffffu)]; }

// get_available_prob, lm_trie.c:653-704 (reached with n_hist < order - 1 only)
__device__ inline float lm_available_prob(const LmDev &m, int32_t wid, const int32_t *hist, int n_hist, int &n_used)
{
    LmRange node;
    float prob = lm_ug_prob(m, wid);
    uint32_t o = 0;
    n_used = 1;
    lm_ug_range(m, wid, node);
    if (n_hist == 0) return prob;
    bool indep = node.begin == node.end;
    int k = 0;
    for (;; ++k) {
        if (k == n_hist) return prob;
        if (indep) return prob;
        if (k == m.order - 2) break;
        const bool found = lm_middle_find(m, k, hist[k], node, o);
        indep = !found || node.begin == node.end;
        if (!found) return prob;
        prob = lm_mid_prob(m, k, o);
        n_used = k + 2;
    }
    if (lm_longest_find(m, hist[k], node, o)) { prob = lm_long_prob(m, o); n_used = m.order; }
    return prob;
}
// get_available_backoff, lm_trie.c:706-731
__device__ inline float lm_available_backoff(const LmDev &m, int start, const int32_t *hist, int n_hist)
{
    float backoff = 0.0f;
    LmRange node;
    uint32_t o = 0;
    lm_ug_range(m, hist[0], node);
    if (start <= 1) { backoff = __fadd_rn(backoff, lm_ug_bo(m, hist[0])); start = 2; }
    for (int k = start - 1; k < n_hist; ++k) {
        if (!lm_middle_find(m, k - 1, hist[k], node, o)) break;
        backoff = __fadd_rn(backoff, lm_mid_bo(m, k - 1, o));
    }
    return backoff;
}
// lm_trie_hist_score with the back-off cache of update_backoff computed in place, lm_trie.c:744-811
__device__ inline float lm_hist_score(const LmDev &m, int32_t wid, const int32_t *hist, int n_hist, int &n_used)
{
    float cache[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    LmRange node;
    uint32_t o = 0;
    if (n_hist > 0) {
        cache[0] = lm_ug_bo(m, hist[0]);
        lm_ug_range(m, hist[0], node);
        for (int i = 1; i < n_hist; ++i) {
            if (!lm_middle_find(m, i - 1, hist[i], node, o)) break;
            cache[i] = lm_mid_bo(m, i - 1, o);
        }
    }
    n_used = 1;
    float prob = lm_ug_prob(m, wid);
    lm_ug_range(m, wid, node);
    if (n_hist == 0) return prob;
    for (int i = 0; i < n_hist - 1; ++i) {
        if (!lm_middle_find(m, i, hist[i], node, o)) {
            for (int j = i; j < n_hist; ++j) prob = __fadd_rn(prob, cache[j]);
            return prob;
        }
        ++n_used;
        prob = lm_mid_prob(m, i, o);
    }
    if (!lm_longest_find(m, hist[n_hist - 1], node, o)) return __fadd_rn(prob, cache[n_hist - 1]);
    ++n_used;
    return lm_long_prob(m, o);
}

// ngram_tg_score(lmset, w3, w2, w1, &n_used) with dictionary word ids; w2 / w1 may be -1.  (lm_tg_score_call below is the
// out-of-line form for kernels with several call sites.)
__device__ inline int32_t lm_tg_score_one(const LmDev &m, int32_t w3, int32_t w2, int32_t w1, int &n_used)
{
    int32_t hist[2];
    int n_hist = min(2, m.order - 1);                       // ngram_model_set.c:693
    const int32_t wid = psgpu_as_global(m.widmap)[w3];
    const int32_t *const hm = psgpu_as_global(m.histmap);     // (never NULL on the device: psgpu_lm_create points it at widmap -- a choice
                                                               //  between two pointers here would make every access through it generic)
    hist[0] = w2 < 0 ? -1 : hm[w2];
    hist[1] = w1 < 0 ? -1 : hm[w1];
    if (wid == -1) return m.log_zero;                       // ngram_model.c:394 (n_used stays what it was: a set's next member may leave the one before's)
    for (int i = 0; i < n_hist; ++i) if (hist[i] < 0) { n_hist = i; break; }    // ngram_model_trie.c:724-731
    float s;
    if (n_hist < m.order - 1) {                             // lm_trie.c:813-828, :733-742
        s = lm_available_prob(m, wid, hist, n_hist, n_used);
        if (!(n_hist < n_used)) s = __fadd_rn(s, lm_available_backoff(m, n_used, hist, n_hist));
    }
    else
        s = lm_hist_score(m, wid, hist, n_hist, n_used);
    const int32_t raw = (int32_t)s;
    const int32_t ws = (int32_t)__fadd_rn(__fmul_rn((float)raw, m.lw), (float)m.log_wip);    // weight_score, ngram_model_trie.c:710
    return m.cwt ? ws + psgpu_as_global(m.cwt)[w3] : ws;    // "multiply by unigram in-class weight", ngram_model.c:415-416
}
// A set without a current model (ngram_model_set.c:697-714): out of line and on its own -- the members' descriptors live in device
// memory, the caller's may be a kernel argument: one inlined body serving both would reach every table through a pointer of unknown
// address space (flat loads, which wait on both memory counters; tests/test_static_compile.py counts them)
static __device__ __attribute__((noinline, unused)) int32_t lm_set_score(const LmDev *set_dev, const int32_t *set_lw, int32_t n_set, const uint32_t *addtab,
                                                                         int32_t addtab_n, int32_t add_zero, int32_t log_zero, int32_t w3, int32_t w2,
                                                                         int32_t w1, int &n_used)
{
    int32_t score = log_zero;
    for (int i = 0; i < n_set; ++i) {
        const int32_t y = psgpu_as_global(set_lw)[i] + lm_tg_score_one(psgpu_as_global(set_dev)[i], w3, w2, w1, n_used);
        // logmath_add (util/logmath.c:401-446) with the set's table
        if (score <= add_zero) { score = y; continue; }
        if (y <= add_zero) continue;
        const int32_t r = score > y ? score : y, d = (int32_t)((uint32_t)r - (uint32_t)(score > y ? y : score));
        score = (d < 0 || d >= addtab_n) ? r : r + (int32_t)psgpu_as_global(addtab)[d];
    }
    return score;
}
__device__ inline int32_t lm_tg_score(const LmDev &m, int32_t w3, int32_t w2, int32_t w1, int &n_used)
{
    n_used = 0;
    if (m.n_set > 0) return lm_set_score(m.set, m.set_lw, m.n_set, m.addtab, m.addtab_n, m.add_zero, m.log_zero, w3, w2, w1, n_used);
    return lm_tg_score_one(m, w3, w2, w1, n_used);
}

// Out of line, the descriptor read from device memory: ONE copy of the trie walk (~1,400 instructions) per kernel however
// many call sites it has.  A frame loop that outgrows the instruction cache (64 KB per pair of CUs) refetches itself from
// L2 every frame; the search kernel calls this from three places.
static __device__ __attribute__((noinline, unused)) int32_t lm_tg_score_call(const LmDev *m_dev, int32_t w3, int32_t w2, int32_t w1)
{
    int nu;
    return lm_tg_score(*psgpu_as_global(m_dev), w3, w2, w1, nu);
}
