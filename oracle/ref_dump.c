/* oracle/ref_dump.c -- TEST INFRASTRUCTURE (golden generator), not product.
 *
 * Links against the UNMODIFIED reference built by oracle/Makefile into
 * oracle/_ref/libpocketsphinx.so and uses the reference's internal headers to
 *   tables  : dump the model tables exactly as the reference holds them after
 *             ps_init() (mean, precomputed var, det, mixw, sen2cb, 8-bit
 *             log-add table, tmat tp, sseq ...)
 *   feats   : run the reference front end over a raw PCM file and dump the
 *             39-dim dynamic features acmod would hand to frame_eval()
 *   ptm     : drive the reference's ptm_mgau_frame_eval() over a feature
 *             matrix (compallsen) and dump int16 senone scores + raw int32
 *             top-N densities + normalised top-N per frame
 *   senlog  : full ps_decode_raw() with the reference's frame_eval wrapped
 *             by a recorder: every (frame, active list, scores) call is logged
 *   decode  : hypothesis + segmentation of ps_decode_raw()
 *   hmm     : hmm_vit_eval() over a seeded random HMM population using the
 *             model's real tmat/sseq tables; state before/after every step
 *
 * The file #includes the reference's ptm_mgau.c *from where it lies* so the
 * static stages (eval_topn/eval_cb/codebook_norm/senone_eval) can be called
 * one by one to capture the raw (pre-normalisation) top-N lists.  No
 * reference source is copied into this repository.
 *
 * Output container ("PSGB1"): a flat sequence of named n-d arrays, read by
 * oracle/psgb.py.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "ptm_mgau.c"           /* reference source, included in place */
#include "s2_semi_mgau.h"
#include "ms_mgau.h"
#include "tmat.h"
#include "bin_mdef.h"
#include "hmm.h"
#include "lm/ngram_model_set.h"
#include "psgpu_lm_tables.h"
#include "psgpu_search_tables.h"

/* ------------------------------------------------------------------ */
static FILE *g_out;

static void
psgb_open(const char *path)
{
    g_out = fopen(path, "wb");
    if (!g_out) { perror(path); exit(2); }
    fwrite("PSGB1\n", 1, 6, g_out);
}

/* dtype codes: 'f' float32, 'i' int32, 'h' int16, 'B' uint8, 'H' uint16, 'q' int64, 'd' float64 */
static void
psgb_put(const char *name, char dtype, int ndim, const int64_t *dims, const void *data)
{
    uint32_t nl = (uint32_t)strlen(name), dt = (uint32_t)dtype, nd = (uint32_t)ndim;
    size_t esz = (dtype == 'f' || dtype == 'i') ? 4 : (dtype == 'h' || dtype == 'H') ? 2
               : (dtype == 'q' || dtype == 'd') ? 8 : 1;
    size_t n = 1;
    int i;
    for (i = 0; i < ndim; ++i) n *= (size_t)dims[i];
    fwrite(&nl, 4, 1, g_out); fwrite(name, 1, nl, g_out);
    fwrite(&dt, 4, 1, g_out); fwrite(&nd, 4, 1, g_out);
    fwrite(dims, 8, ndim, g_out);
    fwrite(data, esz, n, g_out);
}
static void put1(const char *name, char dt, int64_t a, const void *d)
{ int64_t dims[1] = {a}; psgb_put(name, dt, 1, dims, d); }
static void put2(const char *name, char dt, int64_t a, int64_t b, const void *d)
{ int64_t dims[2] = {a, b}; psgb_put(name, dt, 2, dims, d); }
static void put3(const char *name, char dt, int64_t a, int64_t b, int64_t c, const void *d)
{ int64_t dims[3] = {a, b, c}; psgb_put(name, dt, 3, dims, d); }
static void put4(const char *name, char dt, int64_t a, int64_t b, int64_t c, int64_t e, const void *d)
{ int64_t dims[4] = {a, b, c, e}; psgb_put(name, dt, 4, dims, d); }
static void puti(const char *name, int32_t v) { put1(name, 'i', 1, &v); }
/* psgpu_table_emit_fn (integration/psgpu_search_tables.h) onto this file's writer */
static void emit_put(void *ctx, const char *name, char dt, int nd, const int64_t *dims, const void *data)
{ (void)ctx; psgb_put(name, dt, nd, dims, data); }

/* ------------------------------------------------------------------ */
static ps_decoder_t *
make_decoder(const char *modeldir, const char *lm, const char *dict, int argc, char **argv)
{
    ps_config_t *config = ps_config_init(NULL);
    ps_decoder_t *ps;
    int i;
    ps_config_set_str(config, "hmm", modeldir);
    if (lm) ps_config_set_str(config, "lm", lm);
    if (dict) ps_config_set_str(config, "dict", dict);
    ps_config_set_str(config, "loglevel", "ERROR");
    /* extra key=value settings */
    for (i = 0; i + 1 < argc; i += 2) {
        const char *k = argv[i], *v = argv[i + 1];
        if (k[0] == '-') ++k;
        if (ps_config_set_str(config, k, v) == NULL) {
            fprintf(stderr, "bad config %s=%s\n", k, v); exit(2);
        }
    }
    ps = ps_init(config);
    if (!ps) { fprintf(stderr, "ps_init failed\n"); exit(2); }
    return ps;
}

static float *
read_f32(const char *path, int64_t *n)
{
    FILE *fp = fopen(path, "rb");
    long sz; float *buf;
    if (!fp) { perror(path); exit(2); }
    fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    buf = malloc(sz);
    if (fread(buf, 1, sz, fp) != (size_t)sz) { perror("read"); exit(2); }
    fclose(fp);
    *n = sz / 4;
    return buf;
}

static int16 *
read_pcm(const char *path, size_t *n)
{
    FILE *fp = fopen(path, "rb");
    long sz; int16 *buf;
    if (!fp) { perror(path); exit(2); }
    fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    buf = malloc(sz);
    if (fread(buf, 1, sz, fp) != (size_t)sz) { perror("read"); exit(2); }
    fclose(fp);
    *n = sz / 2;
    return buf;
}

/* ------------------------------------------------------------------ */
/* one utterance through the public API: raw PCM (ps_process_raw) or, for a
 * path ending in .mfc, Sphinx cepstra (ps_process_cep) as pocketsphinx_batch
 * does for its -cepdir inputs (programs/pocketsphinx_batch.c:195-250,478-488) */
static void
run_utt(ps_decoder_t *ps, const char *path)
{
    size_t len = strlen(path);
    if (len > 4 && strcmp(path + len - 4, ".mfc") == 0) {
        int ceplen = ps_config_int(ps_get_config(ps), "ceplen");
        FILE *fp = fopen(path, "rb");
        long flen; int32 nmfc; int nfr, i, swap = 0;
        float32 **mfcs;
        if (!fp) { perror(path); exit(2); }
        fseek(fp, 0, SEEK_END); flen = ftell(fp); fseek(fp, 0, SEEK_SET);
        if (fread(&nmfc, 4, 1, fp) != 1) { perror("mfc"); exit(2); }
        if (nmfc != flen / 4 - 1) {
            nmfc = (int32)__builtin_bswap32((uint32_t)nmfc); swap = 1;
            if (nmfc != flen / 4 - 1) { fprintf(stderr, "%s: not an MFCC file\n", path); exit(2); }
        }
        nfr = nmfc / ceplen;
        mfcs = (float32 **)ckd_calloc_2d(nfr, ceplen, sizeof(float32));
        if (fread(mfcs[0], 4, (size_t)nfr * ceplen, fp) != (size_t)nfr * ceplen) { perror("mfc"); exit(2); }
        fclose(fp);
        if (swap)
            for (i = 0; i < nfr * ceplen; ++i) {
                uint32_t *u = (uint32_t *)&mfcs[0][i];
                *u = __builtin_bswap32(*u);
            }
        ps_start_utt(ps);
        ps_process_cep(ps, mfcs, nfr, FALSE, TRUE);
        ps_end_utt(ps);
        ckd_free_2d(mfcs);
    }
    else {
        size_t n; int16 *pcm = read_pcm(path, &n);
        ps_start_utt(ps);
        ps_process_raw(ps, pcm, n, FALSE, TRUE);
        ps_end_utt(ps);
        free(pcm);
    }
}

/* ------------------------------------------------------------------ */
static int
cmd_tables(ps_decoder_t *ps)
{
    acmod_t *acmod = ps->acmod;
    ptm_mgau_t *s = (ptm_mgau_t *)acmod->mgau;
    gauden_t *g = s->g;
    bin_mdef_t *mdef = acmod->mdef;
    tmat_t *tmat = acmod->tmat;
    int32_t i, f;
    int64_t tot;

    if (strcmp(acmod->mgau->vt->name, "ptm") != 0) {
        fprintf(stderr, "not a PTM model\n"); return 2;
    }
    puti("n_mgau", g->n_mgau); puti("n_feat", g->n_feat); puti("n_density", g->n_density);
    put1("featlen", 'i', g->n_feat, g->featlen);
    puti("n_sen", s->n_sen); puti("max_topn", s->max_topn); puti("ds_ratio", s->ds_ratio);
    puti("n_fast_hist", s->n_fast_hist);
    puti("mixw_is_4bit", s->mixw_cb != NULL);
    /* mean/var are [mgau][feat][density][featlen[f]]: contiguous per (mgau, feat)
     * (ms_gauden.c gauden_param_read); with equal featlen the whole block is
     * one contiguous buffer starting at mean[0][0][0]. */
    tot = 0;
    for (f = 0; f < g->n_feat; ++f) tot += g->featlen[f];
    {
        float *mean = malloc(sizeof(float) * g->n_mgau * g->n_density * tot);
        float *var = malloc(sizeof(float) * g->n_mgau * g->n_density * tot);
        float *det = malloc(sizeof(float) * g->n_mgau * g->n_feat * g->n_density);
        int64_t o = 0, od = 0;
        int32_t m, d;
        /* packed as [mgau][feat][density][featlen[f]] in that loop order */
        for (m = 0; m < g->n_mgau; ++m)
            for (f = 0; f < g->n_feat; ++f) {
                for (d = 0; d < g->n_density; ++d) {
                    memcpy(mean + o, g->mean[m][f][d], sizeof(float) * g->featlen[f]);
                    memcpy(var + o, g->var[m][f][d], sizeof(float) * g->featlen[f]);
                    o += g->featlen[f];
                    det[od++] = g->det[m][f][d];
                }
            }
        put1("mean", 'f', o, mean);
        put1("var", 'f', o, var);
        put3("det", 'f', g->n_mgau, g->n_feat, g->n_density, det);
        free(mean); free(var); free(det);
    }
    {
        int32_t cw;
        int64_t rowlen = s->mixw_cb ? (s->n_sen + 1) / 2 : s->n_sen;
        uint8 *mixw = malloc((size_t)g->n_feat * g->n_density * rowlen);
        for (f = 0; f < g->n_feat; ++f)
            for (cw = 0; cw < g->n_density; ++cw)
                memcpy(mixw + ((size_t)f * g->n_density + cw) * rowlen, s->mixw[f][cw], rowlen);
        put3("mixw", 'B', g->n_feat, g->n_density, rowlen, mixw);
        if (s->mixw_cb) put1("mixw_cb", 'B', 16, s->mixw_cb);
        free(mixw);
    }
    put1("sen2cb", 'B', s->n_sen, s->sen2cb);
    {
        logadd_t *t = LOGMATH_TABLE(s->lmath_8b);
        puti("logadd8_size", (int32_t)t->table_size);
        puti("logadd8_width", t->width);
        puti("logadd8_shift", t->shift);
        put1("logadd8", 'B', t->table_size, t->table);
        {
            double base = logmath_get_base(s->lmath_8b);
            int64_t one = 1;
            FILE *o = g_out; (void)o;
            psgb_put("logbase_f64bits", 'q', 1, &one, &base);
        }
    }
    /* transition matrices: tp[tmat][from][to] uint8, n_state x (n_state+1)... */
    puti("n_tmat", tmat->n_tmat); puti("tmat_n_state", tmat->n_state);
    {
        int32_t n = tmat->n_tmat, ns = tmat->n_state, t, a;
        uint8 *tp = malloc((size_t)n * ns * (ns + 1));
        for (t = 0; t < n; ++t)
            for (a = 0; a < ns; ++a)
                memcpy(tp + ((size_t)t * ns + a) * (ns + 1), tmat->tp[t][a], ns + 1);
        put3("tp", 'B', n, ns, ns + 1, tp);
        free(tp);
    }
    /* senone sequences */
    {
        int32_t n_sseq = bin_mdef_n_sseq(mdef), ne = bin_mdef_n_emit_state(mdef);
        uint16 *sseq = malloc(sizeof(uint16) * n_sseq * ne);
        for (i = 0; i < n_sseq; ++i)
            memcpy(sseq + (size_t)i * ne, mdef->sseq[i], sizeof(uint16) * ne);
        puti("n_sseq", n_sseq); puti("n_emit_state", ne);
        put2("sseq", 'H', n_sseq, ne, sseq);
        puti("n_ciphone", bin_mdef_n_ciphone(mdef));
        puti("n_ci_sen", mdef->n_ci_sen);
        free(sseq);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* model tables of the semi-continuous scorer as s2_semi_mgau_init left them */
static int
cmd_tables_semi(ps_decoder_t *ps)
{
    acmod_t *acmod = ps->acmod;
    s2_semi_mgau_t *s = (s2_semi_mgau_t *)acmod->mgau;
    gauden_t *g;
    int32_t f, d;
    int64_t tot = 0, o = 0, od = 0;
    float *mean, *var, *det;

    if (strcmp(acmod->mgau->vt->name, "s2_semi") != 0) {
        fprintf(stderr, "not an s2_semi model (%s)\n", acmod->mgau->vt->name); return 2;
    }
    g = s->g;
    puti("n_mgau", g->n_mgau); puti("n_feat", g->n_feat); puti("n_density", g->n_density);
    put1("featlen", 'i', g->n_feat, g->featlen);
    puti("n_sen", s->n_sen); puti("max_topn", s->max_topn); puti("ds_ratio", s->ds_ratio);
    puti("n_fast_hist", s->n_topn_hist);
    put1("topn_beam", 'B', g->n_feat, s->topn_beam);
    for (f = 0; f < g->n_feat; ++f) tot += g->featlen[f];
    mean = malloc(sizeof(float) * g->n_density * tot);
    var = malloc(sizeof(float) * g->n_density * tot);
    det = malloc(sizeof(float) * g->n_feat * g->n_density);
    for (f = 0; f < g->n_feat; ++f)
        for (d = 0; d < g->n_density; ++d) {
            memcpy(mean + o, g->mean[0][f][d], sizeof(float) * g->featlen[f]);
            memcpy(var + o, g->var[0][f][d], sizeof(float) * g->featlen[f]);
            o += g->featlen[f];
            det[od++] = g->det[0][f][d];
        }
    put1("mean", 'f', o, mean); put1("var", 'f', o, var);
    put2("det", 'f', g->n_feat, g->n_density, det);
    free(mean); free(var); free(det);
    {
        int64_t rowlen = s->mixw_cb ? (s->n_sen + 1) / 2 : s->n_sen;
        uint8 *mixw = malloc((size_t)g->n_feat * g->n_density * rowlen);
        for (f = 0; f < g->n_feat; ++f)
            for (d = 0; d < g->n_density; ++d)
                memcpy(mixw + ((size_t)f * g->n_density + d) * rowlen, s->mixw[f][d], rowlen);
        put3("mixw", 'B', g->n_feat, g->n_density, rowlen, mixw);
        if (s->mixw_cb) put1("mixw_cb", 'B', 16, s->mixw_cb);
        free(mixw);
    }
    {
        logadd_t *t = LOGMATH_TABLE(s->lmath_8b);
        put1("logadd8", 'B', t->table_size, t->table);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* model tables of the multi-stream / continuous scorer (ms_mgau_init) */
static int
cmd_tables_ms(ps_decoder_t *ps)
{
    acmod_t *acmod = ps->acmod;
    ms_mgau_model_t *msg = (ms_mgau_model_t *)acmod->mgau;
    gauden_t *g;
    senone_t *sn;
    int32_t m, f, d, i;
    int64_t tot = 0, o = 0, od = 0;
    float *mean, *var, *det;

    if (strcmp(acmod->mgau->vt->name, "ms") != 0) {
        fprintf(stderr, "not an ms model (%s)\n", acmod->mgau->vt->name); return 2;
    }
    g = msg->g; sn = msg->s;
    puti("n_mgau", g->n_mgau); puti("n_feat", g->n_feat); puti("n_density", g->n_density);
    put1("featlen", 'i', g->n_feat, g->featlen);
    puti("n_sen", (int32_t)sn->n_sen); puti("max_topn", msg->topn); puti("aw", sn->aw);
    puti("n_gauden", (int32_t)sn->n_gauden);
    for (f = 0; f < g->n_feat; ++f) tot += g->featlen[f];
    mean = malloc(sizeof(float) * g->n_mgau * g->n_density * tot);
    var = malloc(sizeof(float) * g->n_mgau * g->n_density * tot);
    det = malloc(sizeof(float) * g->n_mgau * g->n_feat * g->n_density);
    for (m = 0; m < g->n_mgau; ++m)
        for (f = 0; f < g->n_feat; ++f)
            for (d = 0; d < g->n_density; ++d) {
                memcpy(mean + o, g->mean[m][f][d], sizeof(float) * g->featlen[f]);
                memcpy(var + o, g->var[m][f][d], sizeof(float) * g->featlen[f]);
                o += g->featlen[f];
                det[od++] = g->det[m][f][d];
            }
    put1("mean", 'f', o, mean); put1("var", 'f', o, var);
    put3("det", 'f', g->n_mgau, g->n_feat, g->n_density, det);
    free(mean); free(var); free(det);
    {
        /* canonical [sen][feat][cw], whatever the in-memory transposition (ms_senone.c:198-209) */
        uint8 *pdf = malloc((size_t)sn->n_sen * sn->n_feat * sn->n_cw);
        uint32_t *map = malloc(sizeof(uint32_t) * sn->n_sen);
        for (i = 0; (uint32)i < sn->n_sen; ++i) {
            for (f = 0; (uint32)f < sn->n_feat; ++f)
                for (d = 0; (uint32)d < sn->n_cw; ++d)
                    pdf[((size_t)i * sn->n_feat + f) * sn->n_cw + d] =
                        (sn->n_gauden > 1) ? sn->pdf[i][f][d] : sn->pdf[f][d][i];
            map[i] = sn->mgau[i];
        }
        put3("pdf", 'B', sn->n_sen, sn->n_feat, sn->n_cw, pdf);
        put1("sen2mgau", 'i', sn->n_sen, map);
        free(pdf); free(map);
    }
    {
        logadd_t *t = LOGMATH_TABLE(sn->lmath);
        puti("logadd_width", t->width); puti("logadd_size", (int32_t)t->table_size);
        put1("logadd", 'B', (int64_t)t->table_size * t->width, t->table);
        puti("log_zero", logmath_get_zero(sn->lmath));
    }
    return 0;
}

/* ------------------------------------------------------------------ */
static int
cmd_feats(ps_decoder_t *ps, const char *rawpath)
{
    acmod_t *acmod = ps->acmod;
    size_t n; int16 *pcm = read_pcm(rawpath, &n);
    int16 const *p = pcm;
    int nfr, dim = feat_dimension(acmod->fcb);
    acmod_start_utt(acmod);
    acmod_process_raw(acmod, &p, &n, TRUE);
    acmod_end_utt(acmod);
    nfr = acmod->n_feat_frame;
    /* full-utterance mode: feat_buf[feat_outidx][0] is contiguous [nfr][dim] */
    {
        float *out = malloc(sizeof(float) * nfr * dim);
        int t;
        for (t = 0; t < nfr; ++t) {
            int idx = (acmod->feat_outidx + t) % acmod->n_feat_alloc;
            memcpy(out + (size_t)t * dim, acmod->feat_buf[idx][0], sizeof(float) * dim);
        }
        put2("feat", 'f', nfr, dim, out);
        free(out);
    }
    free(pcm);
    return 0;
}

/* ------------------------------------------------------------------ */
/* Drive the reference PTM scorer over a feature matrix, compallsen.
 * Two instances: A = the real ptm_mgau_frame_eval (pure reference call);
 * B = the same stages called one by one (reference statics) so the raw
 * top-N can be captured before ptm_mgau_codebook_norm overwrites it.
 * A and B are asserted identical.  `carry` != 0 keeps the top-N history of
 * a previous segment; segments are given by seglen (frames). */
/* Force exact score ties: overwrite codeword d (d >= n_density/2) of every
 * (codebook, stream) with the parameters of codeword d - n_density/2, in the
 * reference's own in-memory tables. */
static void
dup_codewords(ptm_mgau_t *s)
{
    gauden_t *g = s->g;
    int m, f, d, half = g->n_density / 2;
    for (m = 0; m < g->n_mgau; ++m)
        for (f = 0; f < g->n_feat; ++f)
            for (d = half; d < g->n_density; ++d) {
                memcpy(g->mean[m][f][d], g->mean[m][f][d - half], sizeof(mfcc_t) * g->featlen[f]);
                memcpy(g->var[m][f][d], g->var[m][f][d - half], sizeof(mfcc_t) * g->featlen[f]);
                g->det[m][f][d] = g->det[m][f][d - half];
            }
}

static int
cmd_ptm(ps_decoder_t *psA, ps_decoder_t *psB, const char *featpath, int seglen, int carry, int dup)
{
    ptm_mgau_t *a = (ptm_mgau_t *)psA->acmod->mgau;
    ptm_mgau_t *b = (ptm_mgau_t *)psB->acmod->mgau;
    gauden_t *g = a->g;
    int64_t nfl; float *feat = read_f32(featpath, &nfl);
    int dim = 0, f, nfr, t, cb, k, n_sen = a->n_sen, N = a->max_topn;
    int16 *scrA, *scrB, *allscr;
    int32 *raw_sc, *norm_sc; uint8 *cws;
    mfcc_t *fp[16];
    int nlist = g->n_mgau * g->n_feat;

    if (dup) { dup_codewords(a); dup_codewords(b); }
    for (f = 0; f < g->n_feat; ++f) dim += g->featlen[f];
    nfr = (int)(nfl / dim);
    if (seglen <= 0) seglen = nfr;
    scrA = malloc(sizeof(int16) * n_sen); scrB = malloc(sizeof(int16) * n_sen);
    allscr = malloc(sizeof(int16) * (size_t)n_sen * nfr);
    raw_sc = malloc(sizeof(int32) * (size_t)nfr * nlist * N);
    norm_sc = malloc(sizeof(int32) * (size_t)nfr * nlist * N);
    cws = malloc((size_t)nfr * nlist * N);

    for (t = 0; t < nfr; ++t) {
        int frame = t % seglen;     /* frame index inside the utterance */
        int o = 0, slot;
        ptm_fast_eval_t *lastf;
        for (f = 0; f < g->n_feat; ++f) { fp[f] = feat + (size_t)t * dim + o; o += g->featlen[f]; }
        if (frame == 0) {
            /* acmod_start_utt(): mgau->frame_idx = 0 (acmod.c:419) */
            ps_mgau_base(a)->frame_idx = 0; ps_mgau_base(b)->frame_idx = 0;
            if (!carry) { ptm_mgau_reset_fast_hist(ps_mgau_base(a)); ptm_mgau_reset_fast_hist(ps_mgau_base(b)); }
        }
        /* A: the real thing */
        ptm_mgau_frame_eval(ps_mgau_base(a), scrA, NULL, 0, fp, frame, TRUE);
        /* B: staged (same statements as ptm_mgau_frame_eval, ptm_mgau.c:425-451) */
        slot = frame % b->n_fast_hist;
        b->f = b->hist + slot;
        lastf = (slot == 0) ? b->hist + b->n_fast_hist - 1 : b->hist + slot - 1;
        memcpy(b->f->topn[0][0], lastf->topn[0][0], g->n_mgau * g->n_feat * N * sizeof(ptm_topn_t));
        ptm_mgau_calc_cb_active(b, NULL, 0, TRUE);
        ptm_mgau_codebook_eval(b, fp, frame);
        for (cb = 0; cb < g->n_mgau; ++cb) for (f = 0; f < g->n_feat; ++f) for (k = 0; k < N; ++k) {
            size_t ix = (((size_t)t * g->n_mgau + cb) * g->n_feat + f) * N + k;
            raw_sc[ix] = b->f->topn[cb][f][k].score;
            cws[ix] = (uint8)b->f->topn[cb][f][k].cw;
        }
        ptm_mgau_codebook_norm(b, fp, frame);
        for (cb = 0; cb < g->n_mgau; ++cb) for (f = 0; f < g->n_feat; ++f) for (k = 0; k < N; ++k) {
            size_t ix = (((size_t)t * g->n_mgau + cb) * g->n_feat + f) * N + k;
            norm_sc[ix] = b->f->topn[cb][f][k].score;
        }
        ptm_mgau_senone_eval(b, scrB, NULL, 0, TRUE);
        if (memcmp(scrA, scrB, sizeof(int16) * n_sen) != 0) {
            fprintf(stderr, "staged != frame_eval at frame %d\n", t); return 3;
        }
        for (cb = 0; cb < g->n_mgau; ++cb) for (f = 0; f < g->n_feat; ++f) for (k = 0; k < N; ++k)
            if (a->f->topn[cb][f][k].cw != b->f->topn[cb][f][k].cw ||
                a->f->topn[cb][f][k].score != b->f->topn[cb][f][k].score) {
                fprintf(stderr, "staged topn != frame_eval topn at frame %d\n", t); return 3;
            }
        memcpy(allscr + (size_t)t * n_sen, scrA, sizeof(int16) * n_sen);
        /* acmod_advance(): ++mgau->frame_idx (acmod.c:874) */
        ps_mgau_base(a)->frame_idx++; ps_mgau_base(b)->frame_idx++;
    }
    puti("seglen", seglen); puti("carry", carry);
    put2("senscr", 'h', nfr, n_sen, allscr);
    put4("topn_cw", 'B', nfr, g->n_mgau, g->n_feat, N, cws);
    put4("topn_raw", 'i', nfr, g->n_mgau, g->n_feat, N, raw_sc);
    put4("topn_norm", 'i', nfr, g->n_mgau, g->n_feat, N, norm_sc);
    return 0;
}

/* ------------------------------------------------------------------ */
/* senlog: record every frame_eval call made during ps_decode_raw */
static ps_mgaufuncs_t rec_funcs;
static ps_mgaufuncs_t *orig_funcs;
static int rec_n, rec_cap;
static int32 *rec_frame, *rec_nact, *rec_fidx;
static int64_t *rec_off;           /* offsets into rec_act / rec_scr */
static uint8 *rec_act; static size_t rec_act_n, rec_act_cap;
static int16 *rec_scr; static size_t rec_scr_n, rec_scr_cap;
static int rec_nsen, rec_dim;
static float *rec_feat; static size_t rec_feat_n, rec_feat_cap;

static int
rec_frame_eval(ps_mgau_t *mg, int16 *senscr, uint8 *act, int32 nact,
               mfcc_t **feat, int32 frame, int32 compallsen)
{
    int r = orig_funcs->frame_eval(mg, senscr, act, nact, feat, frame, compallsen);
    if (rec_n == rec_cap) {
        rec_cap = rec_cap ? rec_cap * 2 : 1024;
        rec_frame = realloc(rec_frame, sizeof(int32) * rec_cap);
        rec_nact = realloc(rec_nact, sizeof(int32) * rec_cap);
        rec_fidx = realloc(rec_fidx, sizeof(int32) * rec_cap);
        rec_off = realloc(rec_off, sizeof(int64_t) * rec_cap);
    }
    rec_frame[rec_n] = frame; rec_nact[rec_n] = compallsen ? -1 : nact;
    rec_fidx[rec_n] = mg->frame_idx;
    rec_off[rec_n] = (int64_t)rec_act_n;
    if (!compallsen) {
        if (rec_act_n + nact > rec_act_cap) {
            rec_act_cap = (rec_act_n + nact) * 2;
            rec_act = realloc(rec_act, rec_act_cap);
        }
        memcpy(rec_act + rec_act_n, act, nact); rec_act_n += nact;
    }
    if (rec_scr_n + rec_nsen > rec_scr_cap) {
        rec_scr_cap = (rec_scr_n + rec_nsen) * 2;
        rec_scr = realloc(rec_scr, sizeof(int16) * rec_scr_cap);
    }
    memcpy(rec_scr + rec_scr_n, senscr, sizeof(int16) * rec_nsen); rec_scr_n += rec_nsen;
    if (rec_feat_n + rec_dim > rec_feat_cap) {
        rec_feat_cap = (rec_feat_n + rec_dim) * 2;
        rec_feat = realloc(rec_feat, sizeof(float) * rec_feat_cap);
    }
    /* streams are contiguous inside one frame vector (feat.c:356-384) */
    memcpy(rec_feat + rec_feat_n, feat[0], sizeof(float) * rec_dim); rec_feat_n += rec_dim;
    ++rec_n;
    return r;
}

static void
dump_hyp(ps_decoder_t *ps, const char *prefix)
{
    int32 score = 0; char name[128];
    const char *hyp = ps_get_hyp(ps, &score);
    ps_seg_t *seg;
    int n = 0, cap = 256;
    int32 *segs = malloc(sizeof(int32) * 5 * cap);
    char words[8192]; size_t wl = 0;
    if (!hyp) hyp = "";
    snprintf(name, sizeof name, "%shyp", prefix);
    put1(name, 'B', (int64_t)strlen(hyp), hyp);
    snprintf(name, sizeof name, "%shyp_score", prefix);
    puti(name, score);
    words[0] = 0;
    for (seg = ps_seg_iter(ps); seg; seg = ps_seg_next(seg)) {
        int sf, ef; int32 ascr, lscr, lback;
        const char *w = ps_seg_word(seg);
        ps_seg_frames(seg, &sf, &ef);
        ps_seg_prob(seg, &ascr, &lscr, &lback);
        if (n == cap) { cap *= 2; segs = realloc(segs, sizeof(int32) * 5 * cap); }
        segs[n * 5 + 0] = sf; segs[n * 5 + 1] = ef; segs[n * 5 + 2] = ascr;
        segs[n * 5 + 3] = lscr; segs[n * 5 + 4] = lback;
        wl += snprintf(words + wl, sizeof words - wl, "%s\n", w);
        ++n;
    }
    snprintf(name, sizeof name, "%sseg", prefix);
    put2(name, 'i', n, 5, segs);
    snprintf(name, sizeof name, "%sseg_words", prefix);
    put1(name, 'B', (int64_t)wl, words);
    free(segs);
}

static int
cmd_senlog(ps_decoder_t *ps, const char *rawpath, int nrep)
{
    int r;
    rec_nsen = bin_mdef_n_sen(ps->acmod->mdef);
    rec_dim = feat_dimension(ps->acmod->fcb);
    orig_funcs = ps->acmod->mgau->vt;
    rec_funcs = *orig_funcs;
    rec_funcs.frame_eval = rec_frame_eval;
    ps->acmod->mgau->vt = &rec_funcs;
    for (r = 0; r < nrep; ++r) {
        char pfx[32];
        run_utt(ps, rawpath);
        snprintf(pfx, sizeof pfx, "utt%d_", r);
        dump_hyp(ps, pfx);
    }
    ps->acmod->mgau->vt = orig_funcs;
    puti("n_calls", rec_n); puti("n_sen", rec_nsen);
    put1("call_frame", 'i', rec_n, rec_frame);
    put1("call_nact", 'i', rec_n, rec_nact);
    put1("call_frame_idx", 'i', rec_n, rec_fidx);
    put1("call_act_off", 'q', rec_n, rec_off);
    put1("call_act", 'B', (int64_t)rec_act_n, rec_act ? rec_act : (uint8 *)"");
    put2("call_scr", 'h', rec_n, rec_nsen, rec_scr);
    put2("call_feat", 'f', rec_n, rec_dim, rec_feat);
    return 0;
}

static int
cmd_decode(ps_decoder_t *ps, const char *rawpath)
{
    run_utt(ps, rawpath);
    dump_hyp(ps, "");
    puti("n_frames", ps_get_n_frames(ps));
    return 0;
}

/* ------------------------------------------------------------------ */
/* hmm: drive the reference's hmm_vit_eval() (hmm.c:786-805) over a seeded
 * random population of HMMs that uses the decoder's real tmat / sseq tables;
 * dump the complete state before and after every step. */
static uint32_t g_rng;
static uint32_t rnd(void) { g_rng = g_rng * 1664525u + 1013904223u; return g_rng >> 8; }

#define HF 19   /* score[5] history[5] out_score out_history senid[5] bestscore tmatid */
static void
hmm_pack(const hmm_t *h, int32 *o)
{
    int i;
    for (i = 0; i < 5; ++i) { o[i] = h->score[i]; o[5 + i] = h->history[i]; o[12 + i] = h->senid[i]; }
    o[10] = h->out_score; o[11] = h->out_history;
    o[17] = h->bestscore; o[18] = h->tmatid;
}

static int
hmm_run(int n_emit, int n_tmat, uint8 ***tp, int n_sseq, uint16 **sseq, int n_sen, int n_hmm, int n_steps, int seed)
{
    int16 *senscr = calloc(n_sen, sizeof(int16));
    hmm_context_t *ctx = hmm_context_init(n_emit, tp, senscr, sseq);
    hmm_t *h = calloc(n_hmm, sizeof(hmm_t));
    int32 *before = malloc(sizeof(int32) * (size_t)n_steps * n_hmm * HF);
    int32 *after = malloc(sizeof(int32) * (size_t)n_steps * n_hmm * HF);
    int32 *ret = malloc(sizeof(int32) * (size_t)n_steps * n_hmm);
    int16 *allscr = malloc(sizeof(int16) * (size_t)n_steps * n_sen);
    uint8 *mpx = malloc(n_hmm);
    uint8 *tpflat = malloc((size_t)n_tmat * n_emit * (n_emit + 1));
    uint16 *sseqflat = malloc(sizeof(uint16) * (size_t)n_sseq * n_emit);
    int i, t, st, a, b;

    g_rng = (uint32_t)seed;
    for (i = 0; i < n_tmat; ++i) for (a = 0; a < n_emit; ++a) for (b = 0; b <= n_emit; ++b)
        tpflat[((size_t)i * n_emit + a) * (n_emit + 1) + b] = tp[i][a][b];
    for (i = 0; i < n_sseq; ++i) for (a = 0; a < n_emit; ++a)
        sseqflat[(size_t)i * n_emit + a] = sseq[i][a];
    for (i = 0; i < n_hmm; ++i) {
        mpx[i] = (uint8)(rnd() & 1);
        hmm_init(ctx, &h[i], mpx[i], rnd() % n_sseq, rnd() % n_tmat);
    }
    for (t = 0; t < n_steps; ++t) {
        /* senone scores: mostly speech-like, some saturated */
        for (i = 0; i < n_sen; ++i) {
            uint32_t r = rnd();
            senscr[i] = (int16)((r & 15) == 0 ? 32767 - (r >> 8) % 100 : (r >> 4) % 4000);
        }
        memcpy(allscr + (size_t)t * n_sen, senscr, sizeof(int16) * n_sen);
        for (i = 0; i < n_hmm; ++i) {
            uint32_t r = rnd();
            /* (re-)enter a third of them; push a few towards the WORST_SCORE clamp */
            if (t == 0 || r % 3 == 0)
                hmm_enter(&h[i], -(int32)(rnd() % 200000), (int32)(rnd() % 5000), t);
            if (r % 53 == 0)
                for (st = 0; st < n_emit; ++st)
                    h[i].score[st] = WORST_SCORE + (int32)(rnd() % 3000) - 200;
            if (r % 41 == 0)
                hmm_clear(&h[i]);
            if (mpx[i] && r % 7 == 0 && n_emit > 1) {          /* mpx: states carry their own ssids */
                st = 1 + rnd() % (n_emit - 1);
                h[i].senid[st] = (rnd() % 5 == 0) ? 0xffff : (uint16)(rnd() % n_sseq);
            }
            if (t == 0 && r % 2)                   /* start some with populated inner states */
                for (st = 1; st < n_emit; ++st) {
                    h[i].score[st] = -(int32)(rnd() % 300000);
                    h[i].history[st] = (int32)(rnd() % 5000);
                    if (mpx[i]) h[i].senid[st] = (uint16)(rnd() % n_sseq);
                }
            hmm_pack(&h[i], before + ((size_t)t * n_hmm + i) * HF);
            ret[(size_t)t * n_hmm + i] = hmm_vit_eval(&h[i]);
            hmm_pack(&h[i], after + ((size_t)t * n_hmm + i) * HF);
        }
    }
    puti("n_emit", n_emit); puti("n_sseq", n_sseq); puti("n_tmat", n_tmat); puti("n_sen", n_sen);
    put3("tp", 'B', n_tmat, n_emit, n_emit + 1, tpflat);
    put2("sseq", 'H', n_sseq, n_emit, sseqflat);
    put1("mpx", 'B', n_hmm, mpx);
    put2("senscr", 'h', n_steps, n_sen, allscr);
    put3("before", 'i', n_steps, n_hmm, HF, before);
    put3("after", 'i', n_steps, n_hmm, HF, after);
    put2("ret", 'i', n_steps, n_hmm, ret);
    return 0;
}

static int
cmd_hmm(ps_decoder_t *ps, int n_hmm, int n_steps, int seed)
{
    bin_mdef_t *mdef = ps->acmod->mdef;
    tmat_t *tmat = ps->acmod->tmat;
    return hmm_run(bin_mdef_n_emit_state(mdef), tmat->n_tmat, tmat->tp, bin_mdef_n_sseq(mdef), mdef->sseq,
                   bin_mdef_n_sen(mdef), n_hmm, n_steps, seed);
}

/* hmmsyn: the same drive over a SYNTHETIC context with n_emit emitting states (1..5; no bundled model has other than
 * 3): random upper-triangular transition matrices with missing arcs (255 = TMAT_WORST_SCORE's magnitude) and skips,
 * random senone sequences.  n_emit other than 3 and 5 reaches hmm_vit_eval_anytopo (hmm.c:710-784). */
static int
cmd_hmmsyn(int n_emit, int n_hmm, int n_steps, int seed)
{
    const int n_tmat = 11, n_sseq = 61, n_sen = 400;
    uint8 ***tp = (uint8 ***)ckd_calloc_3d(n_tmat, n_emit, n_emit + 1, sizeof(uint8));
    uint16 **sseq = (uint16 **)ckd_calloc_2d(n_sseq, n_emit, sizeof(uint16));
    int i, a, b;
    g_rng = (uint32_t)seed * 2654435761u + 12345u;
    for (i = 0; i < n_tmat; ++i)
        for (a = 0; a < n_emit; ++a)
            for (b = 0; b <= n_emit; ++b) {
                uint32_t r = rnd();
                tp[i][a][b] = (b < a || r % 4 == 0) ? 255 : (uint8)(1 + (r >> 3) % 140);
            }
    for (i = 0; i < n_sseq; ++i)
        for (a = 0; a < n_emit; ++a)
            sseq[i][a] = (uint16)(rnd() % n_sen);
    return hmm_run(n_emit, n_tmat, tp, n_sseq, sseq, n_sen, n_hmm, n_steps, seed);
}

/* ------------------------------------------------------------------ */
/* dynfeat: cepstra file (.mfc) -> the decoder's dynamic feature computation
 * for a whole utterance (feat_s2mfc2feat_live(begin, end), feat/feat.c:1310:
 * batch CMN + window padding + 1s_c_d_dd deltas + subvector split) */
static int
cmd_dynfeat(ps_decoder_t *ps, const char *mfcpath)
{
    feat_t *fcb = ps->acmod->fcb;
    int ceplen = feat_cepsize(fcb), dim = feat_dimension(fcb);
    FILE *fp = fopen(mfcpath, "rb");
    long flen; int32 nmfc; int nfr, i, swap = 0, nout;
    float32 **mfcs, *copy;
    mfcc_t ***feat;

    if (!fp) { perror(mfcpath); return 2; }
    fseek(fp, 0, SEEK_END); flen = ftell(fp); fseek(fp, 0, SEEK_SET);
    if (fread(&nmfc, 4, 1, fp) != 1) return 2;
    if (nmfc != flen / 4 - 1) { nmfc = (int32)__builtin_bswap32((uint32_t)nmfc); swap = 1; }
    nfr = nmfc / ceplen;
    mfcs = (float32 **)ckd_calloc_2d(nfr, ceplen, sizeof(float32));
    if (fread(mfcs[0], 4, (size_t)nfr * ceplen, fp) != (size_t)nfr * ceplen) return 2;
    fclose(fp);
    if (swap)
        for (i = 0; i < nfr * ceplen; ++i) { uint32_t *u = (uint32_t *)&mfcs[0][i]; *u = __builtin_bswap32(*u); }
    copy = malloc(sizeof(float) * (size_t)nfr * ceplen);
    memcpy(copy, mfcs[0], sizeof(float) * (size_t)nfr * ceplen);
    feat = feat_array_alloc(fcb, nfr);
    nout = nfr;
    nout = feat_s2mfc2feat_live(fcb, mfcs, &nout, TRUE, TRUE, feat);
    puti("cepsize", ceplen); puti("n_out", nout);
    put2("cep", 'f', nfr, ceplen, copy);
    put2("feat", 'f', nout, dim, feat[0][0]);
    free(copy);
    return 0;
}


/* ------------------------------------------------------------------ */
/* mfcc: the front end's precomputed tables (fe_t / melfb_t, fe_internal.h:70-161)
 * and the cepstra of one recording computed exactly as acmod_process_full_raw
 * does (acmod.c:552-557: fe_start_utt, fe_process_frames, fe_end_utt), starting
 * from reset noise statistics (what ps_start_stream leaves, pocketsphinx.c:1081).
 * With nrep > 1 the recording is processed again WITHOUT a reset, so the second
 * block shows the noise tracker carried across utterances. */
#include "fe/fe_internal.h"
#include "fe/fe_noise.h"
static int
cmd_mfcc(fe_t *fe, const char *rawpath, int nrep)
{
    melfb_t *mel = fe->mel_fb;
    size_t n; int16 *pcm = read_pcm(rawpath, &n);
    int32 par[16];
    int i, r, ncoef = 0, outdim = fe_get_output_size(fe);
    float *cos_flat;
    if (!pcm) return 2;
    par[0] = fe->frame_size; par[1] = fe->frame_shift; par[2] = fe->fft_size; par[3] = fe->fft_order;
    par[4] = mel->num_filters; par[5] = fe->num_cepstra; par[6] = fe->feature_dimension;
    par[7] = fe->transform; par[8] = fe->log_spec; par[9] = fe->remove_dc; par[10] = fe->noise_stats != NULL;
    par[11] = mel->lifter_val; par[12] = fe->swap; par[13] = fe->dither; par[14] = 0; par[15] = 0;
    put1("par", 'i', 16, par);
    { int32 seed = fe->dither_seed; put1("dither_seed", 'i', 1, &seed); }
    put1("alpha", 'f', 1, &fe->pre_emphasis_alpha);
    put1("sqrt_inv_n", 'f', 1, &mel->sqrt_inv_n);
    put1("sqrt_inv_2n", 'f', 1, &mel->sqrt_inv_2n);
    put1("hamming", 'd', fe->frame_size / 2, fe->hamming_window);
    put1("ccc", 'd', fe->fft_size / 4, fe->ccc);
    put1("sss", 'd', fe->fft_size / 4, fe->sss);
    put1("spec_start", 'h', mel->num_filters, mel->spec_start);
    put1("filt_start", 'h', mel->num_filters, mel->filt_start);
    put1("filt_width", 'h', mel->num_filters, mel->filt_width);
    for (i = 0; i < mel->num_filters; ++i) ncoef += mel->filt_width[i];
    put1("filt_coeffs", 'f', ncoef, mel->filt_coeffs);
    cos_flat = malloc(sizeof(float) * fe->num_cepstra * mel->num_filters);
    for (i = 0; i < fe->num_cepstra; ++i)
        memcpy(cos_flat + i * mel->num_filters, mel->mel_cosine[i], sizeof(float) * mel->num_filters);
    put2("mel_cosine", 'f', fe->num_cepstra, mel->num_filters, cos_flat);
    if (mel->lifter_val) put1("lifter", 'f', fe->num_cepstra, mel->lifter);
    put1("pcm", 'h', (int64_t)n, pcm);
    fe_reset_noisestats(fe->noise_stats);
    for (r = 0; r < nrep; ++r) {
        int16 const *p = pcm; size_t ns = n; int32 nfr, ntail; char name[32];
        mfcc_t **cep;
        fe_process_frames(fe, NULL, &ns, NULL, &nfr);
        cep = (mfcc_t **)ckd_calloc_2d(nfr + 1, outdim, sizeof(mfcc_t));
        fe_start_utt(fe);
        fe_process_frames(fe, &p, &ns, cep, &nfr);
        fe_end_utt(fe, cep[nfr], &ntail);
        nfr += ntail;
        snprintf(name, sizeof name, r ? "cep%d" : "cep", r);
        put2(name, 'f', nfr, outdim, cep[0]);
        ckd_free_2d(cep);
    }
    free(cos_flat); free(pcm);
    return 0;
}

/* ------------------------------------------------------------------ */
/* fwdtree: everything the lexicon-tree search of one utterance consumes and
 * produces (ngram_search_fwdtree.c), for pinning oracle/ps_oracle_search.c:
 *   static  the tree as create_search_channels built it, flattened (roots first,
 *           then the other nodes in depth-first order), single-phone word
 *           channels, dictionary and dict2pid tables, beams and penalties, and
 *           the language model as a dense table over dictionary word ids
 *           (ngram_tg_score(w3, w2, w1) >> SENSCR_SHIFT for every triple, -1 =
 *           no history) -- small vocabularies only;
 *   trace   per fwdtree frame: the senone scores acmod_score handed to the
 *           search (active ids + scores + the value of every other entry) and the
 *           phone-loop penalties it read;
 *   result  the back-pointer table and score stack after ngram_fwdtree_finish,
 *           per-frame marks, best scores, hypothesis.
 * Run with fwdflat/bestpath off so that nothing else touches the tables. */
#include "ngram_search.h"
#include "phone_loop_search.h"
#include "dict2pid.h"
#include "lm/ngram_model.h"

static ps_searchfuncs_t ft_vt, *ft_orig;
static int ft_in_step = -1;                 /* frame of the fwdtree step in progress */
static ps_decoder_t *ft_ps;
static int32 *ft_pen; static size_t ft_pen_n, ft_pen_cap;
static int32 *ft_step_frame, *ft_best, *ft_bpidx, *ft_lpbest; static size_t ft_n, ft_cap;
static int32 *ft_act; static int16 *ft_scr; static size_t ft_act_n, ft_act_cap;
static int64_t *ft_act_off; static int16 *ft_rest;
static ps_mgaufuncs_t ft_mvt, *ft_morig;

static int
ft_frame_eval(ps_mgau_t *mg, int16 *senscr, uint8 *act, int32 nact, mfcc_t **feat, int32 frame, int32 compallsen)
{
    int r = ft_morig->frame_eval(mg, senscr, act, nact, feat, frame, compallsen);
    if (ft_in_step >= 0 && frame == ft_in_step) {
        /* the search's own call: absolute ids of the listed senones and their scores */
        int i, sen = 0, n_sen = bin_mdef_n_sen(ft_ps->acmod->mdef);
        uint8 *listed = calloc(n_sen, 1);
        if (compallsen) nact = n_sen;                  /* every senone is "listed" */
        if (ft_act_n + nact > ft_act_cap) {
            ft_act_cap = (ft_act_n + nact) * 2 + 1024;
            ft_act = realloc(ft_act, sizeof(int32) * ft_act_cap);
            ft_scr = realloc(ft_scr, sizeof(int16) * ft_act_cap);
        }
        ft_act_off[ft_n] = (int64_t)ft_act_n;
        for (i = 0; i < nact; ++i) {
            sen = compallsen ? i : sen + act[i];
            ft_act[ft_act_n] = sen; ft_scr[ft_act_n] = senscr[sen]; ++ft_act_n;
            listed[sen] = 1;
        }
        ft_rest[ft_n] = 0;
        for (i = 0; i < n_sen; ++i) if (!listed[i]) { ft_rest[ft_n] = senscr[i]; break; }
        free(listed);
    }
    return r;
}

static int
ft_step(ps_search_t *search, int frame_idx)
{
    ngram_search_t *ngs = (ngram_search_t *)search;
    phone_loop_search_t *pls = (phone_loop_search_t *)ps_search_lookahead(search);
    int n_ci = bin_mdef_n_ciphone(ps_search_acmod(search)->mdef), rv, i;
    if (ft_n == ft_cap) {
        ft_cap = ft_cap ? ft_cap * 2 : 1024;
        ft_step_frame = realloc(ft_step_frame, sizeof(int32) * ft_cap);
        ft_best = realloc(ft_best, sizeof(int32) * ft_cap);
        ft_lpbest = realloc(ft_lpbest, sizeof(int32) * ft_cap);
        ft_bpidx = realloc(ft_bpidx, sizeof(int32) * ft_cap);
        ft_act_off = realloc(ft_act_off, sizeof(int64_t) * (ft_cap + 1));
        ft_rest = realloc(ft_rest, sizeof(int16) * ft_cap);
    }
    if (ft_pen_n + n_ci > ft_pen_cap) {
        ft_pen_cap = (ft_pen_n + n_ci) * 2;
        ft_pen = realloc(ft_pen, sizeof(int32) * ft_pen_cap);
    }
    for (i = 0; i < n_ci; ++i) ft_pen[ft_pen_n + i] = pls ? pls->penalties[i] : 0;
    ft_pen_n += n_ci;
    ft_act_off[ft_n] = (int64_t)ft_act_n;
    ft_rest[ft_n] = 0;
    ft_in_step = frame_idx;
    rv = ft_orig->step(search, frame_idx);
    ft_in_step = -1;
    ft_step_frame[ft_n] = frame_idx; ft_best[ft_n] = ngs->best_score; ft_lpbest[ft_n] = ngs->last_phone_best_score;
    ft_bpidx[ft_n] = ngs->bpidx;
    ++ft_n;
    ft_act_off[ft_n] = (int64_t)ft_act_n;
    return rv;
}

/* ---- fwdflat (`fwdflat` command): the second pass (ngram_search_fwdflat.c) runs inside the search's finish();
 * the wrapper snapshots what pass 1 left (back-pointer table, score stack, the multiplex ssids of the permanent
 * single-phone channels: hmm_clear does not reset them), the scorer hook records the scores each pass-2 frame was
 * handed and, one call later, the best score / back-pointer count that frame ended with. */
static int ff_on, ff_phase;                 /* tracing a two-pass decode; inside finish() */
static int32 *ff_bp1, *ff_bss1, *ff_idx1, *ff_w1ssid; static int ff_nb1, ff_nbss1, ff_nfr1;
static int32 *ff_best, *ff_bpidx; static size_t ff_n, ff_cap;
static int32 *ff_wordlist; static int ff_nwd;
static int32 *ff_seed; static int ff_seed_dims[3]; static float *ff_feat; static int ff_feat_dim;
static int32 *ff_act; static int16 *ff_scr, *ff_rest; static size_t ff_act_n, ff_act_cap; static int64_t *ff_act_off;

static void
ff_record_frame(ps_mgau_t *mg, int16 *senscr, uint8 *act, int32 nact, int32 frame, int32 compallsen)
{
    ngram_search_t *ngs = (ngram_search_t *)ft_ps->search;
    int i, sen = 0, n_sen = bin_mdef_n_sen(ft_ps->acmod->mdef);
    uint8 *listed = calloc(n_sen, 1);
    (void)mg;
    if ((size_t)frame != ff_n) { fprintf(stderr, "fwdflat trace: frame %d out of order (%zu recorded)\n", frame, ff_n); exit(2); }
    if (ff_n + 2 > ff_cap) {
        ff_cap = ff_cap ? ff_cap * 2 : 1024;
        ff_best = realloc(ff_best, sizeof(int32) * ff_cap); ff_bpidx = realloc(ff_bpidx, sizeof(int32) * ff_cap);
        ff_act_off = realloc(ff_act_off, sizeof(int64_t) * (ff_cap + 1)); ff_rest = realloc(ff_rest, sizeof(int16) * ff_cap);
    }
    if (frame > 0) { ff_best[frame - 1] = ngs->best_score; ff_bpidx[frame - 1] = ngs->bpidx; }   /* what frame - 1 ended with */
    else {                                              /* the utterance's vocabulary as build_fwdflat_wordlist left it */
        for (ff_nwd = 0; ngs->fwdflat_wordlist[ff_nwd] >= 0; ++ff_nwd);
        ff_wordlist = calloc(ff_nwd + 1, 4);
        memcpy(ff_wordlist, ngs->fwdflat_wordlist, sizeof(int32) * ff_nwd);
    }
    if (compallsen) nact = n_sen;
    if (ff_act_n + nact > ff_act_cap) {
        ff_act_cap = (ff_act_n + nact) * 2 + 1024;
        ff_act = realloc(ff_act, sizeof(int32) * ff_act_cap); ff_scr = realloc(ff_scr, sizeof(int16) * ff_act_cap);
    }
    ff_act_off[ff_n] = (int64_t)ff_act_n;
    for (i = 0; i < nact; ++i) {
        sen = compallsen ? i : sen + act[i];
        ff_act[ff_act_n] = sen; ff_scr[ff_act_n] = senscr[sen]; ++ff_act_n;
        listed[sen] = 1;
    }
    ff_rest[ff_n] = 0;
    for (i = 0; i < n_sen; ++i) if (!listed[i]) { ff_rest[ff_n] = senscr[i]; break; }
    free(listed);
    ++ff_n;
    ff_act_off[ff_n] = (int64_t)ff_act_n;
}

static int
ff_frame_eval(ps_mgau_t *mg, int16 *senscr, uint8 *act, int32 nact, mfcc_t **feat, int32 frame, int32 compallsen)
{
    int r;
    if (!ff_phase) return ft_frame_eval(mg, senscr, act, nact, feat, frame, compallsen);
    r = ft_morig->frame_eval(mg, senscr, act, nact, feat, frame, compallsen);
    ff_record_frame(mg, senscr, act, nact, frame, compallsen);
    return r;
}

static int
ff_finish(ps_search_t *search)
{
    ngram_search_t *ngs = (ngram_search_t *)search;
    int n_emit = bin_mdef_n_emit_state(ps_search_acmod(search)->mdef), i, k, rv;
    ff_nb1 = ngs->bpidx; ff_nbss1 = ngs->bss_head; ff_nfr1 = ngs->n_frame;
    ff_bp1 = calloc((size_t)ff_nb1 * 10 + 1, 4); ff_bss1 = calloc(ff_nbss1 + 1, 4); ff_idx1 = calloc(ff_nfr1 + 2, 4);
    for (i = 0; i < ff_nb1; ++i) {
        bptbl_t *e = &ngs->bp_table[i];
        int32 *b = ff_bp1 + (size_t)i * 10;
        b[0] = e->frame; b[1] = e->valid; b[2] = e->wid; b[3] = e->bp; b[4] = e->score; b[5] = e->s_idx;
        b[6] = e->real_wid; b[7] = e->prev_real_wid; b[8] = e->last_phone; b[9] = e->last2_phone;
    }
    memcpy(ff_bss1, ngs->bscore_stack, sizeof(int32) * ff_nbss1);
    memcpy(ff_idx1, ngs->bp_table_idx, sizeof(int32) * ff_nfr1);
    ff_idx1[ff_nfr1] = ngs->bpidx;                       /* the mark ngram_fwdtree_finish adds */
    ff_w1ssid = calloc((size_t)ngs->n_1ph_words * n_emit + 1, 4);
    for (i = 0; i < ngs->n_1ph_words; ++i) {
        root_chan_t *r = (root_chan_t *)ngs->word_chan[ngs->single_phone_wid[i]];
        for (k = 0; k < n_emit; ++k)
            ff_w1ssid[i * n_emit + k] = hmm_is_mpx(&r->hmm) ? hmm_mpx_ssid(&r->hmm, k) : hmm_nonmpx_ssid(&r->hmm);
    }
    /* what a second pass that scores its own senones needs on top: the feature rows, and -- PTM scorer -- the
     * codeword lists the first pass left in the history slot pass-2 frame 0 starts from (ptm_mgau.c:425-441: slot 0
     * is seeded from slot n_fast_hist - 1) */
    {
        acmod_t *acmod = ps_search_acmod(search);
        int dim = feat_dimension(acmod->fcb), t;
        ff_feat_dim = dim;
        ff_feat = malloc(sizeof(float) * (size_t)(ff_nfr1 + 1) * dim);
        for (t = 0; t < ff_nfr1; ++t) memcpy(ff_feat + (size_t)t * dim, acmod->feat_buf[t % acmod->n_feat_alloc][0], sizeof(float) * dim);
        if (!strcmp(ft_morig->name, "ptm")) {
            ptm_mgau_t *pm = (ptm_mgau_t *)acmod->mgau;
            int m, f, k2, nm = pm->g->n_mgau, nf = pm->g->n_feat, tn = pm->max_topn;
            ptm_fast_eval_t *fe = &pm->hist[pm->n_fast_hist - 1];
            ff_seed = calloc((size_t)nm * nf * tn + 1, 4);
            ff_seed_dims[0] = nm; ff_seed_dims[1] = nf; ff_seed_dims[2] = tn;
            for (m = 0; m < nm; ++m) for (f = 0; f < nf; ++f) for (k2 = 0; k2 < tn; ++k2)
                ff_seed[((size_t)m * nf + f) * tn + k2] = fe->topn[m][f][k2].cw;
        }
    }
    ff_phase = 1;
    rv = ft_orig->finish(search);
    ff_phase = 0;
    if (ff_n > 0) { ff_best[ff_n - 1] = ngs->best_score; ff_bpidx[ff_n - 1] = ngs->bpidx; }
    return rv;
}

static int cmd_lm(ngram_model_t *lmset, const char *qfile);
static int
cmd_fwdtree(ps_decoder_t *ps, const char *rawpath, int flat)
{
    ngram_search_t *ngs = (ngram_search_t *)ps->search;
    acmod_t *acmod = ps->acmod;
    bin_mdef_t *mdef = acmod->mdef;
    dict_t *dict = ps_search_dict(ngs);
    int n_ci = bin_mdef_n_ciphone(mdef), n_emit = bin_mdef_n_emit_state(mdef), n_w = dict_size(dict);
    int R = ngs->n_root_chan, i, j, k;

    if (strcmp(ps_search_type(ps->search), PS_SEARCH_TYPE_NGRAM) || !ngs->fwdtree || (!ngs->fwdflat) != !flat || ngs->bestpath) {
        fprintf(stderr, "fwdtree dump needs an n-gram search with -fwdflat no -bestpath no, fwdflat dump one with -fwdflat yes -bestpath no\n");
        return 2;
    }
    ff_on = flat;
    /* ---- the static tables: the product's own flattener (integration/psgpu_search_tables.c -- the code behind
     *      psgpu_device_search_attach and psgpu_export_tables), under the names the goldens have always used */
    {
        psgpu_search_tables_t *st = psgpu_search_tables_collect(ps, flat);
        if (!st) return 2;
        psgpu_search_tables_emit(st, emit_put, NULL);
        psgpu_search_tables_free(st);
    }
    {   /* the language model over dictionary word ids */
        size_t n1 = (size_t)n_w + 1;
        int32 *lm;
        if (n_w > 400) {       /* too many words for the dense table: the model's trie tables instead (`lm` command) */
            if (cmd_lm(ngs->lmset, NULL) != 0) { fprintf(stderr, "vocabulary too large for a dense LM table and no trie model\n"); return 2; }
            goto traced;
        }
        lm = malloc(sizeof(int32) * n_w * n1 * n1);
        for (i = 0; i < n_w; ++i)
            for (j = -1; j < n_w; ++j)
                for (k = -1; k < n_w; ++k) {
                    int32 nu, v = 0;
                    if (!dict_filler_word(dict, i) && dict_basewid(dict, i) == i)
                        v = ngram_tg_score(ngs->lmset, i, j, k, &nu) >> SENSCR_SHIFT;
                    lm[((size_t)i * n1 + (j + 1)) * n1 + (k + 1)] = v;
                }
        put3("lm", 'i', n_w, (int64_t)n1, (int64_t)n1, lm);
        free(lm);
    }
traced:
    /* REFDUMP_WARMUP=<raw>: the decoder first decodes that utterance, so that the traced one is the SECOND of a session.
     * What the permanent channels carry over -- hmm_clear (hmm.c:181-196) resets scores and histories, not the per-state
     * ssids of the multiplexed roots and single-phone words -- is dumped as they stand when the traced utterance starts. */
    if (getenv("REFDUMP_WARMUP")) {
        int n1 = ngs->n_1ph_words;
        int32 *st = calloc((size_t)(R + n1) * n_emit + 1, 4);
        run_utt(ps, getenv("REFDUMP_WARMUP"));
        for (i = 0; i < R; ++i)
            for (k = 0; k < n_emit; ++k) st[(size_t)i * n_emit + k] = hmm_mpx_ssid(&ngs->root_chan[i].hmm, k);
        for (i = 0; i < n1; ++i) {
            root_chan_t *r = (root_chan_t *)ngs->word_chan[ngs->single_phone_wid[i]];
            for (k = 0; k < n_emit; ++k)
                st[(size_t)(R + i) * n_emit + k] = hmm_is_mpx(&r->hmm) ? hmm_mpx_ssid(&r->hmm, k) : hmm_nonmpx_ssid(&r->hmm);
        }
        put2("mpx_init", 'i', R + n1, n_emit, st);
        free(st);
    }
    /* ---- the decode, traced */
    ft_ps = ps;
    ft_orig = ps->search->vt; ft_vt = *ft_orig; ft_vt.step = ft_step; if (flat) ft_vt.finish = ff_finish; ps->search->vt = &ft_vt;
    ft_morig = acmod->mgau->vt; ft_mvt = *ft_morig; ft_mvt.frame_eval = flat ? ff_frame_eval : ft_frame_eval; acmod->mgau->vt = &ft_mvt;
    run_utt(ps, rawpath);
    ps->search->vt = ft_orig; acmod->mgau->vt = ft_morig;
    dump_hyp(ps, "");
    puti("n_steps", (int32)ft_n);
    put1("step_frame", 'i', ft_n, ft_step_frame); put1("step_best", 'i', ft_n, ft_best);
    put1("step_lpbest", 'i', ft_n, ft_lpbest); put1("step_bpidx", 'i', ft_n, ft_bpidx);
    put2("step_pen", 'i', ft_n, n_ci, ft_pen);
    put1("step_act_off", 'q', ft_n + 1, ft_act_off);
    put1("step_act", 'i', (int64_t)ft_act_n, ft_act); put1("step_scr", 'h', (int64_t)ft_act_n, ft_scr);
    put1("step_rest", 'h', ft_n, ft_rest);
    if (flat) {   /* ---- what pass 1 handed over, and the second pass's trace */
        put2("bp1", 'i', ff_nb1, 10, ff_bp1); put1("bscore_stack1", 'i', ff_nbss1, ff_bss1);
        put1("bp_table_idx1", 'i', ff_nfr1 + 1, ff_idx1);
        put2("flat_w1_ssid", 'i', ngs->n_1ph_words, n_emit, ff_w1ssid);
        put1("flat_wordlist", 'i', ff_nwd, ff_wordlist);
        put2("flat_feat", 'f', ff_nfr1, ff_feat_dim, ff_feat);
        if (ff_seed) put3("flat_ptm_seed", 'i', ff_seed_dims[0], ff_seed_dims[1], ff_seed_dims[2], ff_seed);
        puti("flat_n_steps", (int32)ff_n);
        put1("flat_best", 'i', ff_n, ff_best); put1("flat_bpidx", 'i', ff_n, ff_bpidx);
        put1("flat_act_off", 'q', ff_n + 1, ff_act_off);
        put1("flat_act", 'i', (int64_t)ff_act_n, ff_act); put1("flat_scr", 'h', (int64_t)ff_act_n, ff_scr);
        put1("flat_rest", 'h', ff_n, ff_rest);
    }
    {   /* ---- the result (of the last pass run) */
        int nb = ngs->bpidx;
        int32 *b = calloc((size_t)nb * 10 + 1, 4);
        for (i = 0; i < nb; ++i) {
            bptbl_t *e = &ngs->bp_table[i];
            b[i * 10 + 0] = e->frame; b[i * 10 + 1] = e->valid; b[i * 10 + 2] = e->wid; b[i * 10 + 3] = e->bp;
            b[i * 10 + 4] = e->score; b[i * 10 + 5] = e->s_idx; b[i * 10 + 6] = e->real_wid; b[i * 10 + 7] = e->prev_real_wid;
            b[i * 10 + 8] = e->last_phone; b[i * 10 + 9] = e->last2_phone;
        }
        put2("bp", 'i', nb, 10, b);
        put1("bscore_stack", 'i', ngs->bss_head, ngs->bscore_stack);
        put1("bp_table_idx", 'i', ngs->n_frame + 1, ngs->bp_table_idx);
        puti("n_frame", ngs->n_frame);
    }
    return 0;
}

/* ------------------------------------------------------------------ */

/* ---- language model: the trie's tables + the reference's answers to a list of
 * (w3, w2, w1) queries in the word ids of the model SET (= dictionary word ids inside a decoder) */
static int
cmd_lm(ngram_model_t *lmset, const char *qfile)
{
    psgpu_lm_tables_t t;
    int w;
    if (psgpu_lm_tables_emit(lmset, emit_put, NULL) < 0) return 2;
    if (psgpu_lm_tables_read(lmset, &t) < 0) return 2;
    if (qfile && strcmp(qfile, "-")) {
        FILE *fp = fopen(qfile, "rb");
        long n; int32 *q, *sc, *nu, i, a = -1, b = -1;
        if (!fp) { perror(qfile); return 2; }
        fseek(fp, 0, SEEK_END); n = ftell(fp) / 12; fseek(fp, 0, SEEK_SET);
        q = ckd_calloc(3 * n + 1, 4); sc = ckd_calloc(n + 1, 4); nu = ckd_calloc(n + 1, 4);
        if (fread(q, 12, n, fp) != (size_t)n) { perror(qfile); return 2; }
        fclose(fp);
        /* lm_trie_t starts with an all-zero history cache that a first query with model history
         * (0, 0) would match without filling the back-offs (lm_trie.c:775-828): make one query with
         * another history first, as any decoder will have done */
        for (w = 0; w < t.n_words && b < 0; ++w)
            if (t.widmap[w] > 0) { if (a < 0) a = w; else b = w; }
        if (b >= 0) ngram_tg_score(lmset, a, a, b, &i);
        for (i = 0; i < n; ++i)
            sc[i] = ngram_tg_score(lmset, q[3 * i], q[3 * i + 1], q[3 * i + 2], &nu[i]);
        put2("queries", 'i', n, 3, q); put1("scores", 'i', n, sc); put1("n_used", 'i', n, nu);
    }
    psgpu_lm_tables_release(&t);
    return 0;
}

/* ------------------------------------------------------------------ */
/* dynfeat_cfg: feat_s2mfc2feat_live(begin = end = TRUE) of a feat_t built by feat_init from a type name and normalisation settings
 * (feat.c:704-915) -- every feature type, batch CMN with or without unit variance, agc max -- optionally with a linear transform
 * (feat_lda_transform, lda.c:140-159: a matrix of this program's own, pseudo-random, in the place feat_read_lda leaves one) and a
 * subvector specification (feat_set_subvecs / feat_subvec_project).  Cepstra: a Sphinx .mfc file.
 * args: MFC TYPE CMN VARNORM AGC LDADIM(0: none) SVSPEC(-: none) */
static int
cmd_dynfeat_cfg(const char *mfcpath, const char *type, const char *cmn, int varnorm, const char *agc, int ldadim, const char *svspec)
{
    FILE *fp = fopen(mfcpath, "rb");
    long flen; int32 nmfc; int nfr, i, swap = 0, nout, ceplen = 13, dim, k;
    float32 **mfcs, *copy, *flat;
    mfcc_t ***feat;
    feat_t *fcb;
    if (!fp) { perror(mfcpath); return 2; }
    fseek(fp, 0, SEEK_END); flen = ftell(fp); fseek(fp, 0, SEEK_SET);
    if (fread(&nmfc, 4, 1, fp) != 1) return 2;
    if (nmfc != flen / 4 - 1) { nmfc = (int32)__builtin_bswap32((uint32_t)nmfc); swap = 1; }
    nfr = nmfc / ceplen;
    mfcs = (float32 **)ckd_calloc_2d(nfr, ceplen, sizeof(float32));
    if (fread(mfcs[0], 4, (size_t)nfr * ceplen, fp) != (size_t)nfr * ceplen) return 2;
    fclose(fp);
    if (swap)
        for (i = 0; i < nfr * ceplen; ++i) { uint32_t *u = (uint32_t *)&mfcs[0][i]; *u = __builtin_bswap32(*u); }
    copy = malloc(sizeof(float) * (size_t)nfr * ceplen);
    memcpy(copy, mfcs[0], sizeof(float) * (size_t)nfr * ceplen);
    fcb = feat_init(type, cmn_type_from_str(cmn), varnorm, agc_type_from_str(agc), 0, ceplen);
    if (!fcb) return 2;
    if (ldadim > 0) {
        const int in = (int)fcb->stream_len[0];
        if (fcb->n_stream != 1 || ldadim > in) { fprintf(stderr, "a transform needs one stream and at most its dimension\n"); return 2; }
        fcb->lda = (mfcc_t ***)ckd_calloc_3d(1, ldadim, in, sizeof(mfcc_t));
        fcb->n_lda = 1; fcb->out_dim = ldadim;
        g_rng = 12345u;
        for (i = 0; i < ldadim; ++i) for (k = 0; k < in; ++k) fcb->lda[0][i][k] = ((float)(rnd() & 0xffff) / 32768.0f - 1.0f) * 0.5f;
        put2("lda", 'f', ldadim, in, fcb->lda[0][0]);
    }
    if (strcmp(svspec, "-")) {
        int32 **sv = parse_subvecs(svspec), **p_, *d, n = 0, *lst;
        if (!sv || feat_set_subvecs(fcb, sv) < 0) return 2;
        for (p_ = sv; *p_; ++p_) for (d = *p_; *d != -1; ++d) ++n;
        lst = ckd_calloc(n + 1, sizeof(int32));
        for (n = 0, p_ = sv; *p_; ++p_) for (d = *p_; *d != -1; ++d) lst[n++] = *d;
        put1("subvec", 'i', n, lst);
    }
    feat = feat_array_alloc(fcb, nfr);
    nout = nfr;
    nout = feat_s2mfc2feat_live(fcb, mfcs, &nout, TRUE, TRUE, feat);
    /* the frames' vectors as the scorers read them: streams / subvectors side by side */
    /* (contiguous from feat[i][0]: feat_array_alloc lays a frame's streams side by side, the transform and the projection write their
     *  results to its start -- out_dim / sv_dim values, lda.c:156, feat.c:350) */
    dim = 0;
    if (fcb->sv_dim) dim = fcb->sv_dim;
    else if (fcb->lda) dim = fcb->out_dim;
    else for (i = 0; i < fcb->n_stream; ++i) dim += fcb->stream_len[i];
    flat = malloc(sizeof(float) * (size_t)nout * dim);
    for (i = 0; i < nout; ++i) memcpy(flat + (size_t)i * dim, feat[i][0], sizeof(float) * dim);
    puti("cepsize", ceplen); puti("n_out", nout); puti("dim", dim); puti("window", feat_window_size(fcb));
    put2("cep", 'f', nfr, ceplen, copy);
    put2("feat", 'f', nout, dim, flat);
    return 0;
}

/* ------------------------------------------------------------------ */
/* livefeat: what the decoder's acoustic front half hands its searches when ps_process_raw is fed a recording in CHUNKS
 * (full_utt = FALSE: pocketsphinx.c:1210-1246; acmod_process_raw / acmod_process_mfcbuf / acmod_process_cep, acmod.c:565-762;
 * fe_process_frames with its overflow buffer, fe_interface.c:352-512; feat_s2mfc2feat_live with cmn_live, feat.c:1300-1420,
 * cmn_live.c:86-194): the chunk sizes come from a list that repeats; every feature frame is taken out of the acmod's feature
 * buffer as ps_search_forward would consume it (acmod_advance after each).  `nutt` utterances back to back on ONE decoder
 * after ps_start_stream: the noise tracker and the running cepstral mean carry over from one to the next.
 * Output: feat [total][dim], utt_frames [nutt], the cepstral mean after each utterance, the chunk list. */
static int
cmd_livefeat(ps_decoder_t *ps, const char *rawpath, int nutt, const char *chunks)
{
    acmod_t *acmod = ps->acmod;
    size_t n; int16 *pcm = read_pcm(rawpath, &n);
    int dim = feat_dimension(acmod->fcb), ceplen = feat_cepsize(acmod->fcb);
    int32 csz[64]; int nc = 0, u, k = 0;
    float *out = NULL; size_t no = 0, cap = 0;
    int32 *uf = ckd_calloc(nutt, sizeof(int32));
    float *means = ckd_calloc((size_t)nutt * ceplen, sizeof(float));
    char *cp = ckd_salloc(chunks), *tok;
    for (tok = strtok(cp, ","); tok && nc < 64; tok = strtok(NULL, ",")) csz[nc++] = atoi(tok);
    if (nc == 0) return 2;
    ps_start_stream(ps);
    for (u = 0; u < nutt; ++u) {
        int16 const *p = pcm; size_t left = n;
        int fr0 = (int)no;
        FILE *mfh = tmpfile();
        acmod_start_utt(acmod);
        acmod_set_mfcfh(acmod, mfh);                  /* (the cepstra as acmod_process_cep receives them, before cmn_live) */
        for (;;) {
            size_t take = left < (size_t)csz[k % nc] ? left : (size_t)csz[k % nc];
            int16 const *q = p; size_t m = take;
            ++k;
            /* ps_process_raw's loop (pocketsphinx.c:1228-1243) with the searches replaced by the read-out of their input */
            while (m) {
                if (acmod_process_raw(acmod, &q, &m, FALSE) < 0) return 2;
                while (acmod->n_feat_frame > 0) {
                    int inptr = acmod->feat_outidx;
                    if (no + 1 > cap) { cap = cap ? 2 * cap : 4096; out = realloc(out, sizeof(float) * cap * dim); }
                    memcpy(out + no * dim, acmod->feat_buf[inptr][0], sizeof(float) * dim);
                    ++no;
                    acmod_advance(acmod);
                }
            }
            p += take; left -= take;
            if (left == 0) break;
        }
        /* ps_end_utt (pocketsphinx.c:1313-1333): acmod_end_utt, then the searches drain what it released */
        {
            FILE *keep = fdopen(dup(fileno(mfh)), "rb");      /* (acmod_end_utt closes its handle) */
            long sz; float *cb; char nm[32];
            acmod_end_utt(acmod);
            fseek(keep, 0, SEEK_END); sz = ftell(keep); fseek(keep, 4, SEEK_SET);
            cb = malloc(sz > 4 ? sz - 4 : 4);
            if (sz > 4 && fread(cb, 1, sz - 4, keep) != (size_t)(sz - 4)) return 2;
            fclose(keep);
            { long z; for (z = 0; z < (sz - 4) / 4; ++z) { uint32_t *w = (uint32_t *)cb + z; *w = __builtin_bswap32(*w); } }   /* (acmod_log_mfc writes big-endian) */
            snprintf(nm, sizeof nm, "cep%d", u);
            put2(nm, 'f', (sz - 4) / 4 / ceplen, ceplen, cb);
            free(cb);
        }
        while (acmod->n_feat_frame > 0) {
            int inptr = acmod->feat_outidx;
            if (no + 1 > cap) { cap = cap ? 2 * cap : 4096; out = realloc(out, sizeof(float) * cap * dim); }
            memcpy(out + no * dim, acmod->feat_buf[inptr][0], sizeof(float) * dim);
            ++no;
            acmod_advance(acmod);
        }
        uf[u] = (int32)no - fr0;
        memcpy(means + (size_t)u * ceplen, acmod->fcb->cmn_struct->cmn_mean, sizeof(float) * ceplen);
    }
    puti("dim", dim); puti("n_utt", nutt); puti("n_samples", (int32_t)n);
    put1("chunks", 'i', nc, csz);
    put1("utt_frames", 'i', nutt, uf);
    put2("cmn_mean_after", 'f', nutt, ceplen, means);
    put2("feat", 'f', (int64_t)no, dim, out);
    return 0;
}

/* ------------------------------------------------------------------ */
/* lm_set: a model SET read from an -lmctl file (ngram_model_set_read, lm/ngram_model_set.c:185-330: several models over a merged word
 * list, word classes from class-definition files), looked up as the n-gram search looks its model up (ngram_tg_score on the set) --
 * with one member selected (MODE = select:NAME) or interpolated (MODE = interp, or interp:w0,w1,...; ngram_model_set_score,
 * :685-727).  Written: every member's tables (class words resolved, psgpu_lm_tables_read_member), the set's weights and log-add
 * table, N pseudo-random queries over the set's words (a third with one or both history words absent) and their scores. */
static int
cmd_lm_set(const char *lmctl, const char *mode, int nq, float lw, float wip, const char *qfile)
{
    logmath_t *lmath = logmath_init(1.0001, 0, 1);        /* (the decoder's: a table, shift 0 -- acmod.c:245) */
    ngram_model_t *set = ngram_model_set_read(NULL, lmctl, lmath);
    ngram_model_set_t *s_;
    psgpu_lm_set_info_t info;
    int i, m;
    int32 *q, *sc, *nu;
    if (!set) { fprintf(stderr, "cannot read %s\n", lmctl); return 2; }
    s_ = (ngram_model_set_t *)set;
    ngram_model_apply_weights(set, lw, wip);
    if (!strncmp(mode, "select:", 7)) { if (!ngram_model_set_select(set, mode + 7)) return 2; }
    else if (!strcmp(mode, "interp")) { if (!ngram_model_set_interp(set, NULL, NULL)) return 2; }
    else if (!strncmp(mode, "interp:", 7)) {
        float32 w[16]; const char *names[16]; int n = 0; char *cp = ckd_salloc(mode + 7), *tok;
        for (tok = strtok(cp, ","); tok && n < 16; tok = strtok(NULL, ",")) w[n++] = (float32)atof(tok);
        if (n != s_->n_models) return 2;
        for (i = 0; i < n; ++i) names[i] = s_->names[i];
        if (!ngram_model_set_interp(set, names, w)) return 2;
    }
    else return 2;
    if (psgpu_lm_set_read(set, &info) < 0) return 2;
    puti("n_models", info.n_models); puti("cur", info.cur); puti("set_log_zero", info.log_zero); puti("add_zero", info.add_zero);
    put1("lweights", 'i', info.n_models, info.lweights);
    put1("addtab", 'i', info.addtab_size, info.addtab);
    puti("n_words", set->n_words);
    for (m = 0; m < info.n_models; ++m) {
        psgpu_lm_tables_t t;
        uint32_t lev[PSGPU_LM_MAX_LEVELS * 7];
        char nm[48];
        int32_t v; int l;
        if (psgpu_lm_tables_read_member(set, m, &t) < 0) return 2;
#define NM(x) (snprintf(nm, sizeof nm, "m%d_%s", m, x), nm)
        v = t.order; put1(NM("order"), 'i', 1, &v); v = t.n_unigrams; put1(NM("n_unigrams"), 'i', 1, &v); v = t.n_words; put1(NM("n_words"), 'i', 1, &v);
        put2(NM("unigrams"), 'i', t.n_unigrams + 1, 3, t.unigrams);
        put1(NM("ngram_mem"), 'B', (int64_t)t.ngram_mem_size, t.ngram_mem ? (const void *)t.ngram_mem : (const void *)"");
        for (l = 0; l < t.order - 1; ++l) {
            lev[7 * l] = t.level_offset[l]; lev[7 * l + 1] = t.total_bits[l]; lev[7 * l + 2] = t.word_bits[l];
            lev[7 * l + 3] = t.word_mask[l]; lev[7 * l + 4] = t.max_vocab[l]; lev[7 * l + 5] = t.next_bits[l]; lev[7 * l + 6] = t.next_mask[l];
        }
        put2(NM("levels"), 'i', t.order - 1, 7, lev);
        if (t.order > 1) put2(NM("quant"), 'f', 2 * (t.order - 2) + 1, 65536, t.quant);
        put1(NM("lw"), 'f', 1, &t.lw); v = t.log_wip; put1(NM("log_wip"), 'i', 1, &v); v = t.log_zero; put1(NM("log_zero"), 'i', 1, &v);
        put1(NM("widmap"), 'i', t.n_words, t.widmap);
        if (t.class_weight) { put1(NM("class_weight"), 'i', t.n_words, t.class_weight); put1(NM("histmap"), 'i', t.n_words, t.histmap); }
#undef NM
        psgpu_lm_tables_release(&t);
    }
    q = ckd_calloc(3 * (size_t)nq + 1, 4); sc = ckd_calloc(nq + 1, 4); nu = ckd_calloc(nq + 1, 4);
    g_rng = 777u;
    for (i = 0; i < nq; ++i) {
        q[3 * i] = (int32)(rnd() % (uint32_t)set->n_words);
        q[3 * i + 1] = (rnd() % 6 == 0) ? -1 : (int32)(rnd() % (uint32_t)set->n_words);
        q[3 * i + 2] = (q[3 * i + 1] < 0 || rnd() % 5 == 0) ? -1 : (int32)(rnd() % (uint32_t)set->n_words);
    }
    /* QFILE: int32 triples (w3, w2, w1) in the set's word ids replace the pseudo-random ones (oracle/make_golden.py writes n-grams the
     * members hold, from their text files: the trie's middle and longest levels are reached, not only the back-off to unigrams) */
    if (qfile && strcmp(qfile, "-")) {
        FILE *fp = fopen(qfile, "rb");
        long n;
        if (!fp) { perror(qfile); return 2; }
        fseek(fp, 0, SEEK_END); n = ftell(fp) / 12; fseek(fp, 0, SEEK_SET);
        if (n > nq) n = nq;
        if (fread(q, 12, n, fp) != (size_t)n) return 2;
        fclose(fp);
        for (i = 0; i < 3 * (int)n; ++i) if (q[i] >= set->n_words || q[i] < -1 || (i % 3 == 0 && q[i] < 0)) return 2;
    }
    {   /* the set's words, newline-separated */
        size_t nb = 0; char *words; int w;
        for (w = 0; w < set->n_words; ++w) nb += strlen(ngram_word(set, w)) + 1;
        words = ckd_calloc(nb + 1, 1);
        for (w = 0, nb = 0; w < set->n_words; ++w) { const char *x = ngram_word(set, w); memcpy(words + nb, x, strlen(x)); nb += strlen(x); words[nb++] = '\n'; }
        put1("words", 'B', (int64_t)nb, words);
    }
    for (i = 0; i < nq; ++i) sc[i] = ngram_tg_score(set, q[3 * i], q[3 * i + 1], q[3 * i + 2], &nu[i]);
    put2("queries", 'i', nq, 3, q); put1("scores", 'i', nq, sc); put1("n_used", 'i', nq, nu);
    psgpu_lm_set_release(&info);
    return 0;
}

int
main(int argc, char **argv)
{
    /* usage: ref_dump CMD OUT MODELDIR LM DICT [cmd args...] [-- key val ...] */
    const char *cmd, *out, *modeldir, *lm, *dict;
    int rc = 2, i, xa = argc;
    char **extra = NULL; int nextra = 0;
    if (argc < 6) {
        fprintf(stderr, "usage: ref_dump CMD OUT MODELDIR LM|- DICT|- [args] [-- key val ...]\n");
        return 2;
    }
    cmd = argv[1]; out = argv[2]; modeldir = argv[3];
    lm = strcmp(argv[4], "-") ? argv[4] : NULL;
    dict = strcmp(argv[5], "-") ? argv[5] : NULL;
    for (i = 6; i < argc; ++i)
        if (strcmp(argv[i], "--") == 0) { xa = i; extra = argv + i + 1; nextra = argc - i - 1; break; }
    err_set_loglevel(ERR_ERROR);
    psgb_open(out);
    if (!strcmp(cmd, "tables_ms")) {
        rc = cmd_tables_ms(make_decoder(modeldir, lm, dict, nextra, extra));
    } else if (!strcmp(cmd, "tables_semi")) {
        rc = cmd_tables_semi(make_decoder(modeldir, lm, dict, nextra, extra));
    } else if (!strcmp(cmd, "tables")) {
        rc = cmd_tables(make_decoder(modeldir, lm, dict, nextra, extra));
    } else if (!strcmp(cmd, "feats") && xa > 6) {
        rc = cmd_feats(make_decoder(modeldir, lm, dict, nextra, extra), argv[6]);
    } else if (!strcmp(cmd, "ptm") && xa > 8) {
        ps_decoder_t *a = make_decoder(modeldir, lm, dict, nextra, extra);
        ps_decoder_t *b = make_decoder(modeldir, lm, dict, nextra, extra);
        rc = cmd_ptm(a, b, argv[6], atoi(argv[7]), atoi(argv[8]), xa > 9 ? atoi(argv[9]) : 0);
    } else if (!strcmp(cmd, "senlog") && xa > 7) {
        rc = cmd_senlog(make_decoder(modeldir, lm, dict, nextra, extra), argv[6], atoi(argv[7]));
    } else if (!strcmp(cmd, "dynfeat") && xa > 6) {
        rc = cmd_dynfeat(make_decoder(modeldir, lm, dict, nextra, extra), argv[6]);
    } else if (!strcmp(cmd, "dynfeat_cfg") && xa > 12) {
        rc = cmd_dynfeat_cfg(argv[6], argv[7], argv[8], atoi(argv[9]), argv[10], atoi(argv[11]), argv[12]);
    } else if (!strcmp(cmd, "livefeat") && xa > 8) {
        rc = cmd_livefeat(make_decoder(modeldir, lm, dict, nextra, extra), argv[6], atoi(argv[7]), argv[8]);
    } else if (!strcmp(cmd, "mfcc") && xa > 7) {
        rc = cmd_mfcc(make_decoder(modeldir, lm, dict, nextra, extra)->acmod->fe, argv[6], atoi(argv[7]));
    } else if (!strcmp(cmd, "mfcc_cfg") && xa > 7) {
        /* a front end of its own (fe_init_auto_r, fe_interface.c:198), configured only by the
         * key/value pairs: reaches settings no bundled acoustic model accepts (logspec, ...) */
        ps_config_t *config = ps_config_init(NULL);
        fe_t *fe;
        err_set_loglevel(ERR_ERROR);
        for (i = 0; i + 1 < nextra; i += 2)
            if (ps_config_set_str(config, extra[i], extra[i + 1]) == NULL) {
                fprintf(stderr, "bad config %s\n", extra[i]); return 2;
            }
        fe = fe_init_auto_r(config);
        rc = fe ? cmd_mfcc(fe, argv[6], atoi(argv[7])) : 2;
    } else if (!strcmp(cmd, "fwdtree") && xa > 6) {
        rc = cmd_fwdtree(make_decoder(modeldir, lm, dict, nextra, extra), argv[6], 0);
    } else if (!strcmp(cmd, "fwdflat") && xa > 6) {
        rc = cmd_fwdtree(make_decoder(modeldir, lm, dict, nextra, extra), argv[6], 1);
    } else if (!strcmp(cmd, "lm") && xa > 6) {
        /* the decoder's own model set: word ids are dictionary word ids */
        rc = cmd_lm(((ngram_search_t *)make_decoder(modeldir, lm, dict, nextra, extra)->search)->lmset, argv[6]);
    } else if (!strcmp(cmd, "lm_file") && xa > 8 && lm) {
        /* a model file on its own, as test/unit/test_ngram/test_lm_score.c loads it (logmath 1.0001),
         * wrapped in a one-model set; QUERIES LW WIP */
        logmath_t *lmath = logmath_init(1.0001, 0, 0);
        ngram_model_t *m = ngram_model_read(NULL, lm, NGRAM_AUTO, lmath), *set;
        char *name = "default";
        if (!m) { fprintf(stderr, "cannot read %s\n", lm); return 2; }
        set = ngram_model_set_init(NULL, &m, &name, NULL, 1);
        ngram_model_apply_weights(set, (float32)atof(argv[7]), (float32)atof(argv[8]));
        rc = cmd_lm(set, argv[6]);
    } else if (!strcmp(cmd, "lm_set") && xa > 10) {
        /* ref_dump lm_set OUT - - - LMCTL MODE N_QUERIES LW WIP [QFILE] */
        rc = cmd_lm_set(argv[6], argv[7], atoi(argv[8]), (float)atof(argv[9]), (float)atof(argv[10]), xa > 11 ? argv[11] : NULL);
    } else if (!strcmp(cmd, "hmmsyn") && argc > 6) {
        /* ref_dump hmmsyn out.psgb n_emit n_hmm n_steps seed  (no model) */
        rc = cmd_hmmsyn(atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]));
    } else if (!strcmp(cmd, "hmm") && xa > 8) {
        rc = cmd_hmm(make_decoder(modeldir, lm, dict, nextra, extra), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]));
    } else if (!strcmp(cmd, "decode") && xa > 6) {
        rc = cmd_decode(make_decoder(modeldir, lm, dict, nextra, extra), argv[6]);
    } else {
        fprintf(stderr, "unknown/short command %s\n", cmd);
    }
    fclose(g_out);
    return rc;
}
