/* integration/psgpu_lm_tables.c -- REFERENCE-SIDE code.
 *
 * Reads the tables psgpu_lm_create needs out of the decoder's language model: only structure
 * fields and non-static functions of the unmodified reference are used, nothing here calls into
 * libpsgpu (oracle/ref_dump.c links this file too, to write the same tables into a fixture).
 * The quantisation tables are private to lm_trie_quant.c, so they are recovered through its
 * read functions: value v is placed in a bit array where lm_trie_quant_mboread / _mpread /
 * _lpread look for it, for every v in 0..65535. */
#include <string.h>

#include <pocketsphinx.h>
#include "util/ckd_alloc.h"
#include "lm/ngram_model_internal.h"
#include "lm/ngram_model_set.h"
#include "lm/ngram_model_trie.h"
#include "lm/lm_trie.h"
#include "lm/lm_trie_quant.h"

#include "psgpu_lm_tables.h"

int
psgpu_lm_tables_read(ngram_model_t *lmset, psgpu_lm_tables_t *t)
{
    ngram_model_set_t *set = (ngram_model_set_t *)lmset;
    ngram_model_t *base;
    lm_trie_t *trie;
    int order, l, n_lev, w;
    float *q;
    int32_t *map;

    memset(t, 0, sizeof *t);
    if (lmset == NULL || set->n_models != 1 || set->cur != 0) {
        E_ERROR("psgpu lm: the model set must hold exactly one, selected, model\n");
        return -1;
    }
    base = set->lms[0];
    if (base->funcs == lmset->funcs || base->n_classes != 0 || base->n < 1 || base->n > PSGPU_LM_MAX_LEVELS + 1) {
        E_ERROR("psgpu lm: needs a trie model without word classes, order <= %d\n", PSGPU_LM_MAX_LEVELS + 1);
        return -1;
    }
    trie = ((ngram_model_trie_t *)base)->trie;
    order = base->n; n_lev = order - 1;
    t->order = order; t->n_unigrams = (int32_t)base->n_counts[0]; t->n_words = lmset->n_words;
    t->unigrams = (const uint32_t *)trie->unigrams;          /* unigram_t = {float, float, uint32}: 12 bytes */
    t->ngram_mem = trie->ngram_mem; t->ngram_mem_size = trie->ngram_mem_size;
    for (l = 0; l < n_lev; ++l) {
        base_t *b = l < order - 2 ? &trie->middle_begin[l].base : &trie->longest->base;
        t->level_offset[l] = (uint32_t)(b->base - trie->ngram_mem);
        t->total_bits[l] = b->total_bits; t->word_bits[l] = b->word_bits; t->word_mask[l] = b->word_mask;
        t->max_vocab[l] = b->max_vocab;
        if (l < order - 2) {
            t->next_bits[l] = trie->middle_begin[l].next_mask.bits;
            t->next_mask[l] = trie->middle_begin[l].next_mask.mask;
        }
    }
    if (order > 1) {
        uint32_t v;
        uint8_t buf[16];
        bitarr_address_t a;
        q = ckd_calloc((size_t)(2 * (order - 2) + 1) * 65536, sizeof *q);
        for (v = 0; v < 65536; ++v) {
            memset(buf, 0, sizeof buf);
            a.base = buf; a.offset = 0;
            buf[0] = buf[2] = (uint8_t)v; buf[1] = buf[3] = (uint8_t)(v >> 8);   /* v at bits 0..15 and 16..31 */
            for (l = 0; l < order - 2; ++l) {
                q[(size_t)(2 * l) * 65536 + v] = lm_trie_quant_mpread(trie->quant, a, l);
                q[(size_t)(2 * l + 1) * 65536 + v] = lm_trie_quant_mboread(trie->quant, a, l);
            }
            q[(size_t)(2 * (order - 2)) * 65536 + v] = lm_trie_quant_lpread(trie->quant, a);
        }
        t->quant = q;
    }
    map = ckd_calloc(lmset->n_words > 0 ? lmset->n_words : 1, sizeof *map);
    for (w = 0; w < lmset->n_words; ++w) map[w] = set->widmap[w][0];
    t->widmap = map;
    t->lw = base->lw; t->log_wip = base->log_wip; t->log_zero = base->log_zero;
    return 0;
}

void
psgpu_lm_tables_release(psgpu_lm_tables_t *t)
{
    ckd_free((void *)t->quant); ckd_free((void *)t->widmap);
    memset(t, 0, sizeof *t);
}
