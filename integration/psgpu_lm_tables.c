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
#include <pocketsphinx/logmath.h>

#include "psgpu_lm_tables.h"

int
psgpu_lm_tables_read(ngram_model_t *lmset, psgpu_lm_tables_t *t)
{
    ngram_model_set_t *set = (ngram_model_set_t *)lmset;
    /* the set's CURRENT model (ngram_model_set_select / -lmname; a set of one has it selected): what ngram_model_set_score delegates
     * to (ngram_model_set.c:715-726).  A set without one interpolates its members: psgpu_lm_tables_read_member for each +
     * psgpu_lm_set_read + psgpu_lm_create_interp. */
    memset(t, 0, sizeof *t);
    if (lmset == NULL || set->cur < 0) {
        E_ERROR("psgpu lm: the model set has no current model (an interpolated set: psgpu_lm_tables_read_member / psgpu_lm_set_read)\n");
        return -1;
    }
    return psgpu_lm_tables_read_member(lmset, set->cur, t);
}

int
psgpu_lm_set_read(ngram_model_t *lmset, psgpu_lm_set_info_t *info)
{
    ngram_model_set_t *set = (ngram_model_set_t *)lmset;
    logmath_t *lmath = lmset->lmath;
    memset(info, 0, sizeof *info);
    info->n_models = set->n_models; info->cur = set->cur; info->lweights = set->lweights;
    info->log_zero = lmset->log_zero; info->add_zero = logmath_get_zero(lmath);
    {
        /* logadd_t is private to logmath.c: the table is read through logmath_add itself -- for 0 <= d < size, with 0 and -d both
         * above the zero, it returns 0 + table[d] (logmath.c:417-443) */
        uint32 size = 0, width = 0, shift = 0, d;
        uint32_t *tab;
        if (logmath_get_table_shape(lmath, &size, &width, &shift) < 0 || size == 0 || shift != 0
            || (int64_t)info->add_zero >= -(int64_t)size) {
            E_ERROR("psgpu lm: the set's log-add goes through a table of shift 0 (logmath_add, logmath.c:401-446)\n");
            return -1;
        }
        tab = ckd_calloc(size, sizeof *tab);
        for (d = 0; d < size; ++d) tab[d] = (uint32_t)logmath_add(lmath, 0, -(int)d);
        info->addtab = tab; info->addtab_width = 4; info->addtab_size = (int32_t)size;
    }
    return 0;
}

int
psgpu_lm_tables_read_member(ngram_model_t *lmset, int member, psgpu_lm_tables_t *t)
{
    ngram_model_set_t *set = (ngram_model_set_t *)lmset;
    ngram_model_t *base;
    lm_trie_t *trie;
    int order, l, n_lev, w;
    float *q;
    int32_t *map, *hmap = NULL, *cwt = NULL;

    memset(t, 0, sizeof *t);
    if (lmset == NULL || member < 0 || member >= set->n_models) {
        E_ERROR("psgpu lm: no such member of the model set\n");
        return -1;
    }
    base = set->lms[member];
    if (base->funcs == lmset->funcs || base->n < 1 || base->n > PSGPU_LM_MAX_LEVELS + 1) {
        E_ERROR("psgpu lm: needs a trie model of order <= %d\n", PSGPU_LM_MAX_LEVELS + 1);
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
    if (base->n_classes > 0) {
        hmap = ckd_calloc(lmset->n_words > 0 ? lmset->n_words : 1, sizeof *hmap);
        cwt = ckd_calloc(lmset->n_words > 0 ? lmset->n_words : 1, sizeof *cwt);
    }
    for (w = 0; w < lmset->n_words; ++w) {
        int32 mw = set->widmap[w][member];
        /* ngram_ng_score's "declassify" (ngram_model.c:396-412), done here once per word: a class word is its class's tag word, its
         * in-class weight is added to a look-up FOR it, and a word ngram_class_prob does not find makes the look-up log_zero */
        if (mw != NGRAM_INVALID_WID && NGRAM_IS_CLASSWID(mw)) {
            ngram_class_t *cls = base->classes[NGRAM_CLASSID(mw)];
            int32 cw = ngram_class_prob(cls, mw);
            hmap[w] = cls->tag_wid;
            map[w] = cw == 1 ? -1 : cls->tag_wid;
            cwt[w] = cw == 1 ? 0 : cw;
        }
        else {
            map[w] = mw;
            if (hmap) hmap[w] = mw;
        }
    }
    t->widmap = map; t->histmap = hmap; t->class_weight = cwt;
    t->lw = base->lw; t->log_wip = base->log_wip; t->log_zero = base->log_zero;
    return 0;
}

void
psgpu_lm_set_release(psgpu_lm_set_info_t *info)
{
    ckd_free((void *)info->addtab);
    memset(info, 0, sizeof *info);
}

void
psgpu_lm_tables_release(psgpu_lm_tables_t *t)
{
    ckd_free((void *)t->quant); ckd_free((void *)t->widmap); ckd_free((void *)t->histmap); ckd_free((void *)t->class_weight);
    memset(t, 0, sizeof *t);
}
