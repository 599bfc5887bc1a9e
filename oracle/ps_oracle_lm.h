/* oracle/ps_oracle_lm.h -- TEST INFRASTRUCTURE: the trie language-model oracle
 * (see ps_oracle_lm.c).  Only tests/ may use it. */
#ifndef PS_ORACLE_LM_H
#define PS_ORACLE_LM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pso_lm_s pso_lm_t;

/* The arrays `ref_dump lm` writes (same names).  levels [order - 1][7] uint32:
 * {byte offset in ngram_mem, total_bits, word_bits, word_mask, max_vocab, next_bits, next_mask}.
 * The arrays must outlive the object. */
pso_lm_t *pso_lm_new(int32_t order, int32_t n_unigrams, int32_t n_words, const uint32_t *unigrams,
                     const uint8_t *ngram_mem, uint64_t ngram_mem_size, const uint32_t *levels,
                     const float *quant, float lw, int32_t log_wip, int32_t log_zero, const int32_t *widmap);
void pso_lm_free(pso_lm_t *lm);
/* = ngram_tg_score(lmset, w3, w2, w1, n_used) with dictionary word ids; n_used may be NULL */
int32_t pso_lm_tg_score(const pso_lm_t *lm, int32_t w3, int32_t w2, int32_t w1, int32_t *n_used);
void pso_lm_tg_score_batch(const pso_lm_t *lm, const int32_t *w3, const int32_t *w2, const int32_t *w1,
                           int64_t n, int32_t *score, int32_t *n_used);
#ifdef __cplusplus
}
#endif
#endif
