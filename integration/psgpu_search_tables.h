/* psgpu_search_tables.h -- the reference's n-gram search structures flattened to the index arrays the device searches take
 * (psgpu_fwdtree_tables_t / psgpu_fwdflat_tables_t of include/psgpu.h).
 *
 * Reference-side code: compiled against the reference's internal headers, like the rest of integration/.  ONE flattener with
 * two kinds of caller:
 *   - psgpu_device_decode.c hands the arrays to psgpu_fwdtree_create / psgpu_fwdflat_create out of a live decoder;
 *   - psgpu_export_tables.c (and the test harness oracle/ref_dump.c) write them to a table file (psgpu_table_file.h) that
 *     pocketsphinx_amd/tablefile.py reads back -- how the Python product and bench.py get a task's tables.
 *
 * What is flattened, and where the reference builds it:
 *   lexicon tree                   create_search_channels / init_search_tree, src/ngram_search_fwdtree.c:67-336
 *   single-phone word channels     ngram_fwdtree_init, :380-, ngs->single_phone_wid
 *   dictionary columns             src/dict.h accessors
 *   dict2pid right-context tables  dict2pid_build, src/dict2pid.c (rssid, ldiph_lc)
 *   HMM topology                   tmat_t.tp, bin_mdef_t.sseq
 *   `par`: sizes, beams, penalties, special word ids (ngram_search_t fields set by ngram_search_calc_beams, src/ngram_search.c)
 *   second pass extras             pronunciations as word-internal ssids (dict2pid_internal), CI ssids, LM membership, beams
 *   phone loop                     phone_loop_search_t (src/phone_loop_search.h:75-94)
 */
#ifndef PSGPU_SEARCH_TABLES_H
#define PSGPU_SEARCH_TABLES_H

#include <stdint.h>
#include <pocketsphinx.h>
#include "psgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* one named array: dt = 'i' int32, 'B' uint8, 'H' uint16, 'f' float32, 'd' float64 (the table file's type letters) */
typedef void (*psgpu_table_emit_fn)(void *ctx, const char *name, char dt, int nd, const int64_t *dims, const void *data);

typedef struct psgpu_search_tables_s {
    int32_t par[32];
    int32_t n_ci, n_emit, n_w, R, M, N, n1, n_tmat, n_sseq;
    int32_t *node_ci, *node_ci2, *node_ssid, *node_tmat, *node_child, *node_sib, *node_penult_wid;      /* [N] */
    const int32_t *homophone_set;             /* [n_w], the decoder's own array (not owned) */
    int32_t *w1_wid, *w1_ci, *w1_ci2, *w1_ssid, *w1_tmat, *w1_mpx;                                      /* [n1] */
    int32_t *dict_pronlen, *dict_first, *dict_last, *dict_last2, *dict_basewid, *dict_filler, *dict_real;   /* [n_w] */
    int32_t *rssid_n, *rssid_ssid, *rssid_cimap, *ldiph_lc;       /* [n_ci][n_ci], [n_ci]^3 x 3 */
    uint8_t *tp;                              /* [n_tmat][n_emit][n_emit + 1] */
    uint16_t *sseq;                           /* [n_sseq][n_emit] */
    int32_t *ci_tmat;                         /* [n_ci] */
    /* the second pass's extras (want_flat) */
    int32_t has_flat;
    int32_t *pron_off, *pron_ci, *pron_ssid, *ci_ssid, *lm_known;
    int64_t pron_total;
    int32_t flat_par[16];                     /* fwdflatbeam, fwdflatwbeam, min_ef_width, max_sf_win */
    float flat_lwf;
    /* the phone loop feeding the look-ahead penalties, when the decoder has one */
    int32_t has_pl, pl_par[8];                /* n_phones, window, beam, pbeam, pip, pl_window */
    double pl_weight;
    int32_t *pl_ssid, *pl_tmat;               /* [n_phones] */
} psgpu_search_tables_t;

/* Flattens the decoder's n-gram search (ps->search must be one, with -fwdtree yes).  want_flat: also what the flat-lexicon
 * second pass needs.  NULL on failure (E_ERROR says why). */
psgpu_search_tables_t *psgpu_search_tables_collect(ps_decoder_t *ps, int want_flat);
void psgpu_search_tables_free(psgpu_search_tables_t *t);

/* the arrays under the names psgpu_fwdtree_tables_t / psgpu_fwdflat_tables_t use (and the table file keeps) */
void psgpu_search_tables_emit(const psgpu_search_tables_t *t, psgpu_table_emit_fn emit, void *ctx);
/* ... as the C-ABI's structs (pointers into t; lm is left NULL) */
void psgpu_search_tables_view(const psgpu_search_tables_t *t, psgpu_fwdtree_tables_t *ft, psgpu_fwdflat_tables_t *ff);

/* Dense language-score table over dictionary word ids, lm[w3][w2 + 1][w1 + 1] = ngram_tg_score(w3, w2, w1) >> SENSCR_SHIFT
 * (small vocabularies; ckd_calloc'd, n_w * (n_w + 1)^2 entries).  fixed_point != 0: the trie's back-off cache is filled
 * first, so that no entry records its all-zero initial state (a search never meets that state). */
int32_t *psgpu_search_tables_dense_lm(ps_decoder_t *ps, int fixed_point);

/* The decoder's language model as the device trie's tables (psgpu_lm_tables_t) under the table file's names: order,
 * n_unigrams, n_words, unigrams, ngram_mem, levels, quant, lw, log_wip, log_zero, widmap, words.  -1: not one trie model
 * without classes. */
int psgpu_lm_tables_emit(ngram_model_t *lmset, psgpu_table_emit_fn emit, void *ctx);

#ifdef __cplusplus
}
#endif
#endif
