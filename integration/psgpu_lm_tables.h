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
void psgpu_lm_tables_release(psgpu_lm_tables_t *t);
#ifdef __cplusplus
}
#endif
#endif
