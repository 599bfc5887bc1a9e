/* oracle/ps_oracle_flat.h -- TEST INFRASTRUCTURE: the flat-lexicon second-pass oracle
 * (see ps_oracle_flat.c).  Only tests/ may use it. */
#ifndef PS_ORACLE_FLAT_H
#define PS_ORACLE_FLAT_H
#include <stdint.h>
#include "ps_oracle_search.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the arrays `ref_dump fwdflat` writes (same names) on top of the first pass's static tables */
typedef struct pso_ff_tables_s {
    pso_ft_tables_t ft;                      /* dictionary, dict2pid, topology, single-phone channels, dense LM (or NULL) */
    const int32_t *pron_off;                 /* [n_w + 1] */
    const int32_t *pron_ci, *pron_ssid;      /* [sum pronlen]: CI phone; word-internal ssid (dict2pid_internal) or -1 */
    const int32_t *ci_ssid;                  /* [n_ci] bin_mdef_pid2ssid of the CI phones */
    const int32_t *lm_known;                 /* [n_w] ngram_model_set_known_wid(lmset, dict_basewid(w)) */
    const int32_t *flat_par;                 /* [16] fwdflatbeam, fwdflatwbeam, min_ef_width, max_sf_win */
    float lwf;                               /* fwdflat_fwdtree_lw_ratio */
} pso_ff_tables_t;

typedef struct pso_ff_s pso_ff_t;
struct pso_lm_s;

pso_ff_t *pso_ff_new(const pso_ff_tables_t *t);          /* the tables must outlive the object */
void pso_ff_free(pso_ff_t *s);
void pso_ff_set_lm(pso_ff_t *s, const struct pso_lm_s *lm);
void pso_ff_start(pso_ff_t *s, const int32_t *bp1, int nb1, int n_frame, const int32_t *w1_ssid);
int pso_ff_active_list(pso_ff_t *s, int frame, int32_t *out);
int pso_ff_step(pso_ff_t *s, int frame, const int32_t *ids, const int16_t *scr, int n, int16_t rest);
void pso_ff_finish(pso_ff_t *s, int n_frames);
int32_t pso_ff_best_score(const pso_ff_t *s);
int32_t pso_ff_bpidx(const pso_ff_t *s);
int32_t pso_ff_bss_head(const pso_ff_t *s);
const pso_bp_t *pso_ff_bp(const pso_ff_t *s);
const int32_t *pso_ff_bss(const pso_ff_t *s);
const int32_t *pso_ff_bp_table_idx(const pso_ff_t *s);
int32_t pso_ff_n_words(const pso_ff_t *s);
const int32_t *pso_ff_wordlist(const pso_ff_t *s);
int32_t pso_ff_n_chan(const pso_ff_t *s);

#ifdef __cplusplus
}
#endif
#endif
