/* oracle/ps_oracle_lm.c -- TEST INFRASTRUCTURE (not shipped, never on the product path).
 *
 * CPU restatement of the reference's trigram look-up as the n-gram search calls it:
 *   ngram_tg_score            lm/ngram_model.c:451-458
 *   ngram_model_set_score     lm/ngram_model_set.c:685-732   (one model, set->cur == 0)
 *   ngram_ng_score            lm/ngram_model.c:388-417       (no word classes)
 *   ngram_model_trie_score    lm/ngram_model_trie.c:710-742  (history truncation, weight_score)
 *   lm_trie_score             lm/lm_trie.c:813-828 and the functions it uses, :549-811
 *   bitarr_read_int25         lm/bitarr.c:74-82
 *   lm_trie_quant_*read       lm/lm_trie_quant.c:330-354
 * Pinned against the compiled reference by tests/test_oracle_lm.py on the fixtures
 * tests/golden/lm_*.npz (`ref_dump lm`: tables + ngram_tg_score answers of the reference).
 * The back-off cache of the reference (lm_trie.c:775-811) is a pure function of the history; it
 * is recomputed per call here. */
#include <stdlib.h>
#include <string.h>
#include "ps_oracle_lm.h"

#define MAX_ORDER 5

typedef struct { uint32_t off, total_bits, word_bits, word_mask, max_vocab, next_bits, next_mask; } level_t;

struct pso_lm_s {
    int32_t order, n_unigrams, n_words;
    const uint32_t *ug;
    const uint8_t *mem; uint64_t mem_size;
    level_t lev[MAX_ORDER - 1];
    const float *quant;
    float lw; int32_t log_wip, log_zero;
    const int32_t *widmap;
};

typedef struct { uint32_t begin, end; } range_t;

pso_lm_t *
pso_lm_new(int32_t order, int32_t n_unigrams, int32_t n_words, const uint32_t *unigrams, const uint8_t *ngram_mem,
           uint64_t ngram_mem_size, const uint32_t *levels, const float *quant, float lw, int32_t log_wip,
           int32_t log_zero, const int32_t *widmap)
{
    pso_lm_t *lm;
    int l;
    if (order < 1 || order > MAX_ORDER) return NULL;
    lm = calloc(1, sizeof *lm);
    lm->order = order; lm->n_unigrams = n_unigrams; lm->n_words = n_words; lm->ug = unigrams;
    lm->mem = ngram_mem; lm->mem_size = ngram_mem_size; lm->quant = quant; lm->lw = lw; lm->log_wip = log_wip;
    lm->log_zero = log_zero; lm->widmap = widmap;
    for (l = 0; l < order - 1; ++l) {
        const uint32_t *v = levels + 7 * l;
        lm->lev[l].off = v[0]; lm->lev[l].total_bits = v[1]; lm->lev[l].word_bits = v[2]; lm->lev[l].word_mask = v[3];
        lm->lev[l].max_vocab = v[4]; lm->lev[l].next_bits = v[5]; lm->lev[l].next_mask = v[6];
    }
    return lm;
}

void pso_lm_free(pso_lm_t *lm) { free(lm); }

static float as_float(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }

/* bitarr.c:74: four bytes little-endian at the byte holding bit `offset`, shifted, masked */
static uint32_t
read25(const uint8_t *base, uint32_t offset, uint32_t mask)
{
    const uint8_t *p = base + (offset >> 3);
    uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    return (v >> (offset & 7)) & mask;
}

/* lm_trie.c:549 */
static float ug_prob(const pso_lm_t *lm, uint32_t w) { return as_float(lm->ug[3 * (size_t)w]); }
static float ug_bo(const pso_lm_t *lm, uint32_t w) { return as_float(lm->ug[3 * (size_t)w + 1]); }
static void ug_range(const pso_lm_t *lm, uint32_t w, range_t *r) { r->begin = lm->ug[3 * (size_t)w + 2]; r->end = lm->ug[3 * (size_t)w + 5]; }

/* lm_trie.c:558-600: interpolation search; every product and difference in uint32 as there */
static int
uniform_find(const uint8_t *base, uint32_t total_bits, uint32_t key_mask, uint32_t before_it, uint32_t before_v,
             uint32_t after_it, uint32_t after_v, uint32_t key, uint32_t *out)
{
    if (key > after_v)
        return 0;
    while (after_it - before_it > 1) {
        uint32_t off = key - before_v, range = after_v - before_v, width = after_it - before_it - 1;
        uint32_t pivot = before_it + (1 + (uint32_t)((size_t)(uint32_t)(off * width) / (range + 1)));
        uint32_t mid = read25(base, pivot * total_bits, key_mask);
        if (mid < key) { before_it = pivot; before_v = mid; }
        else if (mid > key) { after_it = pivot; after_v = mid; }
        else { *out = pivot; return 1; }
    }
    return 0;
}

/* lm_trie.c:602-631: returns the bit offset just after the word field (= address.offset), 0 with *found = 0 */
static uint32_t
middle_find(const pso_lm_t *lm, int l, uint32_t word, range_t *r, int *found)
{
    const level_t *m = &lm->lev[l];
    const uint8_t *base = lm->mem + m->off;
    uint32_t at, o;
    if (!uniform_find(base, m->total_bits, m->word_mask, r->begin - 1, 0, r->end, m->max_vocab, word, &at)) {
        *found = 0;
        return 0;
    }
    at *= m->total_bits;
    at += m->word_bits;
    o = at + 32;                                 /* quant_bits of a middle entry: 16 + 16 (lm_trie_quant.c:213) */
    r->begin = read25(base, o, m->next_mask);
    o += m->total_bits;
    r->end = read25(base, o, m->next_mask);
    *found = 1;
    return at;
}

/* lm_trie.c:633-651 */
static uint32_t
longest_find(const pso_lm_t *lm, uint32_t word, const range_t *r, int *found)
{
    const level_t *m = &lm->lev[lm->order - 2];
    uint32_t at;
    if (!uniform_find(lm->mem + m->off, m->total_bits, m->word_mask, r->begin - 1, 0, r->end, m->max_vocab, word, &at)) {
        *found = 0;
        return 0;
    }
    *found = 1;
    return at * m->total_bits + m->word_bits;
}

static float mid_bo(const pso_lm_t *lm, int l, uint32_t o)
{ return lm->quant[(size_t)(2 * l + 1) * 65536 + read25(lm->mem + lm->lev[l].off, o, 0xffff)]; }
static float mid_prob(const pso_lm_t *lm, int l, uint32_t o)
{ return lm->quant[(size_t)(2 * l) * 65536 + read25(lm->mem + lm->lev[l].off, o + 16, 0xffff)]; }
static float long_prob(const pso_lm_t *lm, uint32_t o)
{ return lm->quant[(size_t)(2 * (lm->order - 2)) * 65536 + read25(lm->mem + lm->lev[lm->order - 2].off, o, 0xffff)]; }

/* lm_trie.c:653-704 */
static float
available_prob(const pso_lm_t *lm, int32_t wid, const int32_t *hist, int32_t n_hist, int32_t *n_used)
{
    range_t node;
    float prob = ug_prob(lm, wid);
    int k, found, indep;
    uint32_t o;
    *n_used = 1;
    ug_range(lm, wid, &node);
    if (n_hist == 0)
        return prob;
    indep = node.begin == node.end;
    for (k = 0;; ++k) {
        if (k == n_hist) return prob;
        if (indep) return prob;
        if (k == lm->order - 2) break;
        o = middle_find(lm, k, hist[k], &node, &found);
        indep = !found || node.begin == node.end;
        if (!found) return prob;
        prob = mid_prob(lm, k, o);
        *n_used = k + 2;
    }
    o = longest_find(lm, hist[k], &node, &found);
    if (found) { prob = long_prob(lm, o); *n_used = lm->order; }
    return prob;
}

/* lm_trie.c:706-731 */
static float
available_backoff(const pso_lm_t *lm, int32_t start, const int32_t *hist, int32_t n_hist)
{
    float backoff = 0.0f;
    range_t node;
    int k, found;
    ug_range(lm, hist[0], &node);
    if (start <= 1) { backoff += ug_bo(lm, hist[0]); start = 2; }
    for (k = start - 1; k < n_hist; ++k) {
        uint32_t o = middle_find(lm, k - 1, hist[k], &node, &found);
        if (!found) break;
        backoff += mid_bo(lm, k - 1, o);
    }
    return backoff;
}

/* lm_trie.c:744-773 with the cache of :787-811 computed in place */
static float
hist_score(const pso_lm_t *lm, int32_t wid, const int32_t *hist, int32_t n_hist, int32_t *n_used)
{
    float cache[MAX_ORDER] = { 0, 0, 0, 0, 0 };
    float prob;
    range_t node;
    int i, j, found;
    uint32_t o;

    if (n_hist > 0) {
        cache[0] = ug_bo(lm, hist[0]);
        ug_range(lm, hist[0], &node);
        for (i = 1; i < n_hist; ++i) {
            o = middle_find(lm, i - 1, hist[i], &node, &found);
            if (!found) break;
            cache[i] = mid_bo(lm, i - 1, o);
        }
    }
    *n_used = 1;
    prob = ug_prob(lm, wid);
    ug_range(lm, wid, &node);
    if (n_hist == 0)
        return prob;
    for (i = 0; i < n_hist - 1; ++i) {
        o = middle_find(lm, i, hist[i], &node, &found);
        if (!found) {
            for (j = i; j < n_hist; ++j) prob += cache[j];
            return prob;
        }
        ++*n_used;
        prob = mid_prob(lm, i, o);
    }
    o = longest_find(lm, hist[n_hist - 1], &node, &found);
    if (!found)
        return prob + cache[n_hist - 1];
    ++*n_used;
    return long_prob(lm, o);
}

int32_t
pso_lm_tg_score(const pso_lm_t *lm, int32_t w3, int32_t w2, int32_t w1, int32_t *n_used_out)
{
    int32_t hist[2], n_hist = 2, i, n_used = 0, wid, raw;
    float s;
    /* ngram_model_set.c:693-729: truncate, map word and history ids */
    if (n_hist > lm->order - 1) n_hist = lm->order - 1;
    wid = lm->widmap[w3];
    hist[0] = w2 < 0 ? -1 : lm->widmap[w2];
    hist[1] = w1 < 0 ? -1 : lm->widmap[w1];
    /* ngram_model.c:394: closed vocabulary */
    if (wid == -1) { if (n_used_out) *n_used_out = 0; return lm->log_zero; }
    /* ngram_model_trie.c:724-731 */
    for (i = 0; i < n_hist; ++i) if (hist[i] < 0) { n_hist = i; break; }
    /* lm_trie.c:813-828 */
    if (n_hist < lm->order - 1) {
        s = available_prob(lm, wid, hist, n_hist, &n_used);          /* lm_trie.c:733-742 */
        if (!(n_hist < n_used))
            s = s + available_backoff(lm, n_used, hist, n_hist);
    }
    else
        s = hist_score(lm, wid, hist, n_hist, &n_used);
    raw = (int32_t)s;
    if (n_used_out) *n_used_out = n_used;
    /* weight_score, ngram_model_trie.c:710-714: float32 product, float32 sum, truncation */
    {
        volatile float prod = (float)raw * lm->lw;
        volatile float sum = prod + (float)lm->log_wip;
        return (int32_t)sum;
    }
}

void
pso_lm_tg_score_batch(const pso_lm_t *lm, const int32_t *w3, const int32_t *w2, const int32_t *w1, int64_t n,
                      int32_t *score, int32_t *n_used)
{
    int64_t i;
    for (i = 0; i < n; ++i) score[i] = pso_lm_tg_score(lm, w3[i], w2[i], w1[i], n_used ? &n_used[i] : NULL);
}
