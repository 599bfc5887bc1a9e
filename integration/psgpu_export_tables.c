/* integration/psgpu_export_tables.c -- REFERENCE-SIDE tool: a task's search tables and language model as a table file.
 *
 *   psgpu_export_tables OUT.psgb MODELDIR LM DICT [-- key value ...]
 *
 * Initialises a decoder as any application does (ps_config + ps_init: the reference loads the dictionary, builds dict2pid,
 * reads the language model and builds the lexicon tree -- ngram_fwdtree_init, src/ngram_search_fwdtree.c:67-336, :380-),
 * flattens what the device searches need with the SAME code the live binding uses (psgpu_search_tables.c, also behind
 * psgpu_device_search_attach) and writes it out (psgpu_table_file.c).  The file holds, under the field names of
 * psgpu_fwdtree_tables_t / psgpu_fwdflat_tables_t / psgpu_lm_tables_t: the lexicon tree, single-phone word channels,
 * dictionary columns, dict2pid tables, HMM topology, `par` (sizes, beams, penalties, special word ids), with -fwdflat yes
 * the second pass's extras, the phone loop's parameters, the dictionary's word strings ("dict_words", newline-separated)
 * and the language model: the model's trie when it is one trie model without classes, else (<= 400 words) a dense table
 * "lm".  pocketsphinx_amd/tablefile.py reads it; pocketsphinx_amd/largevocab.py and bench.py build pipelines from it.
 * Nothing is decoded here: this is model loading. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "util/ckd_alloc.h"
#include "ngram_search.h"

#include "acmod.h"
#include "ms_mgau.h"
#include "ms_gauden.h"
#include "ms_senone.h"
#include "tied_mgau_common.h"

#include "psgpu_search_tables.h"
#include "psgpu_table_file.h"

/* The multi-stream scorer's tables as the reference holds them after ms_mgau_init (gauden_init's precomputed variances and
 * determinants, senone_init's quantised mixture weights: src/ms_gauden.c:263-308, src/ms_senone.c:134-320), under "ms_" + the field
 * names psgpu_ms_model_create takes -- what psgpu_mgau_shim.c's ms_upload_model hands the device for a live decoder, written out
 * for a caller that builds the pipeline from a table file (pocketsphinx_amd.MsMgau). */
static void
emit_ms_tables(ps_mgau_t *mg, FILE *fp)
{
    ms_mgau_model_t *msg = (ms_mgau_model_t *)mg;
    gauden_t *gd = msg->g;
    senone_t *sn = msg->s;
    logadd_t *la = LOGMATH_TABLE(sn->lmath);
    size_t tot = 0, o = 0, od = 0;
    float *mean, *var, *det;
    uint8 *pdf;
    int m, f, d;
    uint32 i;
    int32 v;
    int64_t dim, dims[3];
    for (f = 0; f < gd->n_feat; ++f) tot += (size_t)gd->featlen[f];
    tot *= (size_t)gd->n_mgau * gd->n_density;
    mean = ckd_calloc(tot, sizeof(float)); var = ckd_calloc(tot, sizeof(float));
    det = ckd_calloc((size_t)gd->n_mgau * gd->n_feat * gd->n_density, sizeof(float));
    for (m = 0; m < gd->n_mgau; ++m)
        for (f = 0; f < gd->n_feat; ++f)
            for (d = 0; d < gd->n_density; ++d) {
                memcpy(mean + o, gd->mean[m][f][d], sizeof(float) * gd->featlen[f]);
                memcpy(var + o, gd->var[m][f][d], sizeof(float) * gd->featlen[f]);
                o += gd->featlen[f];
                det[od++] = gd->det[m][f][d];
            }
    pdf = ckd_malloc((size_t)sn->n_sen * sn->n_feat * sn->n_cw);      /* canonical [sen][feat][cw] whatever the in-memory transposition */
    for (i = 0; i < sn->n_sen; ++i)
        for (f = 0; (uint32)f < sn->n_feat; ++f)
            for (d = 0; (uint32)d < sn->n_cw; ++d)
                pdf[((size_t)i * sn->n_feat + f) * sn->n_cw + d] = (sn->n_gauden > 1) ? sn->pdf[i][f][d] : sn->pdf[f][d][i];
    dim = 1;
#define PUT1(name, val) do { v = (int32)(val); psgpu_table_file_put(fp, name, 'i', 1, &dim, &v); } while (0)
    PUT1("ms_n_mgau", gd->n_mgau); PUT1("ms_n_feat", gd->n_feat); PUT1("ms_n_density", gd->n_density); PUT1("ms_n_sen", sn->n_sen);
    PUT1("ms_max_topn", msg->topn); PUT1("ms_aw", sn->aw); PUT1("ms_logadd_size", la->table_size); PUT1("ms_logadd_width", la->width);
    PUT1("ms_log_zero", logmath_get_zero(sn->lmath));
#undef PUT1
    dim = gd->n_feat; psgpu_table_file_put(fp, "ms_featlen", 'i', 1, &dim, gd->featlen);
    dim = (int64_t)tot; psgpu_table_file_put(fp, "ms_mean", 'f', 1, &dim, mean); psgpu_table_file_put(fp, "ms_var", 'f', 1, &dim, var);
    dims[0] = gd->n_mgau; dims[1] = gd->n_feat; dims[2] = gd->n_density; psgpu_table_file_put(fp, "ms_det", 'f', 3, dims, det);
    dims[0] = sn->n_sen; dims[1] = sn->n_feat; dims[2] = sn->n_cw; psgpu_table_file_put(fp, "ms_pdf", 'B', 3, dims, pdf);
    dim = sn->n_sen; psgpu_table_file_put(fp, "ms_sen2mgau", 'i', 1, &dim, sn->mgau);
    dim = (int64_t)la->table_size * la->width; psgpu_table_file_put(fp, "ms_logadd", 'B', 1, &dim, la->table);
    ckd_free(mean); ckd_free(var); ckd_free(det); ckd_free(pdf);
}

int
main(int argc, char **argv)
{
    ps_config_t *config;
    ps_decoder_t *ps;
    psgpu_search_tables_t *t;
    ngram_search_t *ngs;
    dict_t *dict;
    FILE *fp;
    int i, xa = argc, w, n_w, rc = 0;
    if (argc < 5) {
        fprintf(stderr, "usage: psgpu_export_tables OUT.psgb MODELDIR LM DICT [-- key value ...]\n");
        return 2;
    }
    for (i = 5; i < argc; ++i) if (!strcmp(argv[i], "--")) { xa = i; break; }
    config = ps_config_init(NULL);
    ps_config_set_str(config, "hmm", argv[2]);
    if (strcmp(argv[3], "-")) ps_config_set_str(config, "lm", argv[3]);
    if (strcmp(argv[4], "-")) ps_config_set_str(config, "dict", argv[4]);
    ps_config_set_str(config, "loglevel", "ERROR");
    for (i = xa + 1; i + 1 < argc; i += 2) {
        const char *k = argv[i][0] == '-' ? argv[i] + 1 : argv[i];
        if (ps_config_set_str(config, k, argv[i + 1]) == NULL) { fprintf(stderr, "bad config %s=%s\n", k, argv[i + 1]); return 2; }
    }
    ps = ps_init(config);
    if (!ps) { fprintf(stderr, "ps_init failed\n"); return 2; }
    if (strcmp(ps_search_type(ps->search), PS_SEARCH_TYPE_NGRAM)) { fprintf(stderr, "not an n-gram search\n"); return 2; }
    ngs = (ngram_search_t *)ps->search;
    t = psgpu_search_tables_collect(ps, ngs->fwdflat ? 1 : 0);
    if (!t) return 2;
    fp = psgpu_table_file_open(argv[1]);
    if (!fp) { perror(argv[1]); return 2; }
    psgpu_search_tables_emit(t, psgpu_table_file_put, fp);
    /* the dictionary's word strings by word id */
    dict = ps_search_dict(ngs); n_w = dict_size(dict);
    {
        size_t nb = 0;
        char *words;
        int64_t dim;
        for (w = 0; w < n_w; ++w) { const char *s = dict_wordstr(dict, w); nb += (s ? strlen(s) : 0) + 1; }
        words = ckd_calloc(nb + 1, 1);
        for (w = 0, nb = 0; w < n_w; ++w) {
            const char *s = dict_wordstr(dict, w);
            if (s) { memcpy(words + nb, s, strlen(s)); nb += strlen(s); }
            words[nb++] = '\n';
        }
        dim = (int64_t)nb;
        psgpu_table_file_put(fp, "dict_words", 'B', 1, &dim, words);
        ckd_free(words);
    }
    /* the language model: the trie's tables, or the dense table of a small vocabulary */
    if (psgpu_lm_tables_emit(ngs->lmset, psgpu_table_file_put, fp) < 0) {
        if (n_w > 400) { fprintf(stderr, "%d words and not one trie model: no language-model tables written\n", n_w); rc = 3; }
        else {
            int32_t *lm = psgpu_search_tables_dense_lm(ps, 1);
            int64_t dims[3];
            dims[0] = n_w; dims[1] = dims[2] = (int64_t)n_w + 1;
            psgpu_table_file_put(fp, "lm", 'i', 3, dims, lm);
            ckd_free(lm);
        }
    }
    /* a decoder whose scorer is the multi-stream one (a continuous model, or -mgau ms / -senmgau .ptm.): its tables too */
    if (!strcmp(ps->acmod->mgau->vt->name, "ms")) emit_ms_tables(ps->acmod->mgau, fp);
    if (psgpu_table_file_close(fp) < 0) { fprintf(stderr, "%s: write failed\n", argv[1]); rc = 2; }
    psgpu_search_tables_free(t);
    ps_free(ps);
    return rc;
}
