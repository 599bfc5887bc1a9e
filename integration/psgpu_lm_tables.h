/* integration/psgpu_lm_tables.h -- read the language model's tables out of a live decoder. */
#ifndef PSGPU_LM_TABLES_H
#define PSGPU_LM_TABLES_H
#include "lm/ngram_model_internal.h"
#include "psgpu.h"
#ifdef __cplusplus
extern "C" {
#endif
/* lmset: the search's model set (ngram_search_t.lmset) after ngram_model_set_map_words, so that its
 * word ids are dictionary word ids.  Fills *t with pointers into the model (ngram_mem, unigrams) and
 * into arrays this call allocates (quant, widmap): release those with psgpu_lm_tables_release.
 * Returns 0, or -1 when the set is not ONE trie model without classes of order <= 5. */
int psgpu_lm_tables_read(ngram_model_t *lmset, psgpu_lm_tables_t *t);
/* member `member` of the set (a trie model, with or without word classes): its tables with ITS column of the set's word-id map,
 * class words resolved to their tag words and in-class weights (psgpu_lm_tables_t.class_weight / .histmap) */
int psgpu_lm_tables_read_member(ngram_model_t *lmset, int member, psgpu_lm_tables_t *t);
/* what a set WITHOUT a current model needs beside its members' tables (psgpu_lm_create_interp): pointers into the set and its logmath */
typedef struct psgpu_lm_set_info_s {
    int32_t n_models, cur;               /* cur = -1: interpolated (ngram_model_set.c:697-714) */
    const int32_t *lweights;             /* [n_models] */
    const void *addtab; int32_t addtab_width, addtab_size, add_zero, log_zero;
} psgpu_lm_set_info_t;
int psgpu_lm_set_read(ngram_model_t *lmset, psgpu_lm_set_info_t *info);       /* (allocates addtab: psgpu_lm_set_release) */
void psgpu_lm_set_release(psgpu_lm_set_info_t *info);
void psgpu_lm_tables_release(psgpu_lm_tables_t *t);
#ifdef __cplusplus
}
#endif
#endif
