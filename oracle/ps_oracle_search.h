/* oracle/ps_oracle_search.h -- TEST INFRASTRUCTURE: the lexicon-tree search oracle
 * (see ps_oracle_search.c).  Only tests/ may use it. */
#ifndef PS_ORACLE_SEARCH_H
#define PS_ORACLE_SEARCH_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bptbl_t (ngram_search.h:112-124), every field widened to int32 */
typedef struct pso_bp_s {
    int32_t frame, valid, wid, bp, score, s_idx, real_wid, prev_real_wid, last_phone, last2_phone;
} pso_bp_t;

/* the arrays `ref_dump fwdtree` writes (same names) */
typedef struct pso_ft_tables_s {
    const int32_t *par;                       /* [32], see ref_dump.c cmd_fwdtree */
    const int32_t *node_ci, *node_ci2, *node_ssid, *node_tmat, *node_child, *node_sib, *node_penult_wid;   /* [R + M] */
    const int32_t *homophone_set;             /* [n_w] */
    const int32_t *w1_wid, *w1_ci, *w1_ci2, *w1_ssid, *w1_tmat, *w1_mpx;                                    /* [n_1ph] */
    const int32_t *dict_pronlen, *dict_first, *dict_last, *dict_last2, *dict_basewid, *dict_filler;        /* [n_w] */
    const int32_t *rssid_n;                   /* [n_ci][n_ci] */
    const int32_t *rssid_ssid, *rssid_cimap;  /* [n_ci][n_ci][n_ci] */
    const int32_t *ldiph_lc;                  /* [n_ci][n_ci][n_ci] */
    const uint8_t *tp;                        /* [n_tmat][n_emit][n_emit + 1] */
    const uint16_t *sseq;                     /* [n_sseq][n_emit] */
    const int32_t *ci_tmat;                   /* [n_ci] */
    const int32_t *lm;                        /* [n_w][n_w + 1][n_w + 1] */
} pso_ft_tables_t;

typedef struct pso_ft_s pso_ft_t;

pso_ft_t *pso_ft_new(const pso_ft_tables_t *t);     /* the tables must outlive the object */
void pso_ft_free(pso_ft_t *s);
void pso_ft_start(pso_ft_t *s);
/* a session's carry-over: the per-state ssids of the permanent multiplexed channels, [R + n_1ph][n_emit] (roots, then
 * single-phone words).  set: into a new object, after pso_ft_start; get: as they stand now */
void pso_ft_set_mpx_ssids(pso_ft_t *s, const int32_t *ssid);
void pso_ft_get_mpx_ssids(const pso_ft_t *s, int32_t *ssid);
/* language scores from a trie model (ps_oracle_lm.h) instead of the dense table, which may then be NULL */
struct pso_lm_s;
void pso_ft_set_lm(pso_ft_t *s, const struct pso_lm_s *lm);
/* 1: run the tree pruning in its data-parallel formulation (per-node decisions on a snapshot +
 * prefix sums for list positions) instead of the reference's sequential walk; same results.
 * 2: the same decisions with work proportional to the active part of the tree (listed nodes and their
 * children) -- the form for large vocabularies */
void pso_ft_set_parallel(pso_ft_t *s, int on);
/* the senone ids compute_sen_active + acmod_flags2list would list for `frame` (bridging entries
 * included); out has room for n_sen entries */
int pso_ft_active_list(pso_ft_t *s, int frame, int32_t *out);
int pso_ft_step(pso_ft_t *s, int frame, const int32_t *ids, const int16_t *scr, int n, int16_t rest,
                const int32_t *penalties);
void pso_ft_finish(pso_ft_t *s, int n_frames);
int32_t pso_ft_best_score(const pso_ft_t *s);
int32_t pso_ft_last_phone_best_score(const pso_ft_t *s);
int32_t pso_ft_bpidx(const pso_ft_t *s);
int32_t pso_ft_bss_head(const pso_ft_t *s);
const pso_bp_t *pso_ft_bp(const pso_ft_t *s);
const int32_t *pso_ft_bss(const pso_ft_t *s);
const int32_t *pso_ft_bp_table_idx(const pso_ft_t *s);

#ifdef __cplusplus
}
#endif
#endif
