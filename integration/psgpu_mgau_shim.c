/* integration/psgpu_mgau_shim.c -- REFERENCE-SIDE BINDING.
 *
 * What a PocketSphinx maintainer adds to plug the MI355X scorer in behind the
 * existing GMM plugin vtable: a `ps_mgau_t` implementation (acmod.h:98-116)
 * whose frame_eval / transform / free forward to the psgpu C ABI
 * (include/psgpu.h).  It is compiled against the reference's own internal
 * headers (-I src) and linked with the reference library; nothing here
 * computes scores -- all arithmetic happens in libpsgpu.so's HIP kernels.
 *
 * Installation is the post-init pointer swap of SURVEY.md 8(b): the reference
 * has no plugin registry (acmod.c:100-119 is a hard-coded if-chain), so
 *
 *     ps_decoder_t *ps = ps_init(config);
 *     psgpu_mgau_attach(ps);            // ps->acmod->mgau := GPU scorer
 *     ps_decode_raw(ps, fh, -1);        // unchanged
 *
 * The wrapped CPU `ptm_mgau_t` is kept: it is the source of the model tables
 * (means, precomputed variances, dets, mixture weights, sen2cb, the 8-bit
 * log-add table) and still serves MLLR transforms, after which the device
 * tables are re-uploaded.  Its top-N history is imported so that attaching in
 * the middle of a session continues bit-exactly.
 */
#include <string.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "acmod.h"
#include "ptm_mgau.h"
#include "s2_semi_mgau.h"
#include "ms_mgau.h"
#include "ms_gauden.h"
#include "ms_senone.h"
#include "tied_mgau_common.h"

#include "psgpu.h"
#include "psgpu_mgau_shim.h"

/* vqFeature_t is private to s2_semi_mgau.c (:64-67); same two int32 fields */
struct vqFeature_s {
    int32 score;
    int32 codeword;
};

typedef struct psgpu_mgau_s {
    ps_mgau_t base;               /* vt + frame_idx: MUST be first (acmod.h:113-116) */
    ptm_mgau_t *cpu;              /* the reference scorer this one replaces ("ptm") ... */
    s2_semi_mgau_t *cpu_semi;     /* ... or ("s2_semi") */
    psgpu_ptm_model_t *model;
    psgpu_ptm_state_t *state;
    psgpu_semi_model_t *smodel;
    psgpu_semi_state_t *sstate;
    ms_mgau_model_t *cpu_ms;      /* ... or ("ms") */
    psgpu_ms_model_t *mmodel;
    acmod_t *acmod;               /* set by psgpu_mgau_attach: lets frame_eval look ahead */
    int la_c0, la_cn;             /* frames announced to psgpu_ptm_state_lookahead */
    int la_expect;                /* next fresh frame if the caller keeps marching */
    float *la_buf;
    int la_cap;
    float *vec;                   /* one frame, streams concatenated */
    int n_feat;
    int veclen;
    int32 n_calls;
} psgpu_mgau_t;

static void shim_announce(struct psgpu_mgau_s *g, int32 frame);
static int shim_frame_eval(ps_mgau_t *ps, int16 *senscr, uint8 *senone_active,
                           int32 n_senone_active, mfcc_t **feat, int32 frame,
                           int32 compallsen);
static int shim_transform(ps_mgau_t *ps, ps_mllr_t *mllr);
static void shim_free(ps_mgau_t *ps);

static ps_mgaufuncs_t psgpu_mgau_funcs = {
    "ptm-psgpu",                  /* name */
    shim_frame_eval,
    shim_transform,
    shim_free
};

static int semi_frame_eval(ps_mgau_t *ps, int16 *senscr, uint8 *senone_active,
                           int32 n_senone_active, mfcc_t **feat, int32 frame,
                           int32 compallsen);
static int semi_transform(ps_mgau_t *ps, ps_mllr_t *mllr);
static void semi_free(ps_mgau_t *ps);

static ps_mgaufuncs_t psgpu_semi_funcs = {
    "s2_semi-psgpu",
    semi_frame_eval,
    semi_transform,
    semi_free
};

/* Upload the tables ptm_mgau_init() built (ptm_mgau.c:804-896). */
static int
upload_model(psgpu_mgau_t *g)
{
    ptm_mgau_t *s = g->cpu;
    gauden_t *gd = s->g;
    logadd_t *la = LOGMATH_TABLE(s->lmath_8b);
    size_t rows = (size_t)gd->n_feat * gd->n_density, r;
    uint8 *mixw;
    int f, d, rc;

    if (la->width != 1) {
        E_ERROR("psgpu: 8-bit log-add table expected (width %d)\n", la->width);
        return -1;
    }
    /* mixw[f][d] rows may live in an mmap of sendump: gather them */
    mixw = ckd_malloc(rows * (size_t)s->n_sen);
    for (r = 0, f = 0; f < gd->n_feat; ++f)
        for (d = 0; d < gd->n_density; ++d, ++r) {
            if (s->mixw_cb) {
                /* a clustered sendump (read_sendump, ptm_mgau.c:457-654): a row is (n_sen + 1) / 2 bytes of two
                 * 4-bit cluster indices.  The weight the reference uses for senone `sen` (ptm_mgau.c:375-379) is
                 *     dcw = row[sen / 2];  dcw = (dcw & 1) ? dcw >> 4 : dcw & 0x0f;  mixw_cb[dcw]
                 * -- the nibble is chosen by the low bit of the BYTE, not of the senone, so both senones of a byte get
                 * the same weight.  That is a per-(row, senone) constant: it is expanded here, quirk included, into the
                 * one-byte-per-senone table the device kernels read. */
                int32 sen;
                for (sen = 0; sen < s->n_sen; ++sen) {
                    int dcw = s->mixw[f][d][sen / 2];
                    dcw = (dcw & 1) ? dcw >> 4 : dcw & 0x0f;
                    mixw[r * (size_t)s->n_sen + sen] = s->mixw_cb[dcw];
                }
            }
            else
                memcpy(mixw + r * (size_t)s->n_sen, s->mixw[f][d], (size_t)s->n_sen);
        }
    if (g->model)
        psgpu_ptm_model_free(g->model);
    g->model = NULL;
    /* mean/var are one contiguous block [mgau][feat][density][featlen]
     * (gauden_param_read, ms_gauden.c:211-221); det likewise (ckd_calloc_3d) */
    rc = psgpu_ptm_model_create(&g->model, gd->n_mgau, gd->n_feat, gd->n_density,
                                gd->featlen, s->n_sen, s->max_topn, s->ds_ratio,
                                gd->mean[0][0][0], gd->var[0][0][0], gd->det[0][0],
                                mixw, s->sen2cb, (const uint8_t *)la->table,
                                (int32_t)la->table_size);
    ckd_free(mixw);
    if (rc != PSGPU_OK) {
        E_ERROR("psgpu_ptm_model_create failed (%d): %s\n", rc, psgpu_last_error());
        return -1;
    }
    return 0;
}

/* Copy the CPU scorer's history ring (ptm_mgau.h:68-71) into the device state. */
static int
import_history(psgpu_mgau_t *g)
{
    ptm_mgau_t *s = g->cpu;
    gauden_t *gd = s->g;
    int n_chain = gd->n_mgau * gd->n_feat, N = s->max_topn;
    int32 *cw = ckd_calloc((size_t)n_chain * N, sizeof(int32));
    int32 *sc = ckd_calloc((size_t)n_chain * N, sizeof(int32));
    uint8 *act = ckd_calloc(gd->n_mgau, 1);
    int slot, i, rc = 0;

    for (slot = 0; slot < s->n_fast_hist && rc == 0; ++slot) {
        ptm_topn_t *tl = s->hist[slot].topn[0][0];
        for (i = 0; i < n_chain * N; ++i) {
            cw[i] = tl[i].cw;
            sc[i] = tl[i].score;
        }
        for (i = 0; i < gd->n_mgau; ++i)
            act[i] = bitvec_is_set(s->hist[slot].mgau_active, i) ? 1 : 0;
        if (psgpu_ptm_state_set_topn(g->state, slot, cw, sc, act) != PSGPU_OK) {
            E_ERROR("psgpu_ptm_state_set_topn failed: %s\n", psgpu_last_error());
            rc = -1;
        }
    }
    ckd_free(cw); ckd_free(sc); ckd_free(act);
    return rc;
}

/* ---- "s2_semi": tables of s2_semi_mgau_init() (s2_semi_mgau.c:1235-1332) ---- */
static int
semi_upload_model(psgpu_mgau_t *g)
{
    s2_semi_mgau_t *s = g->cpu_semi;
    gauden_t *gd = s->g;
    logadd_t *la = LOGMATH_TABLE(s->lmath_8b);
    size_t rowlen = s->mixw_cb ? (size_t)(s->n_sen + 1) / 2 : (size_t)s->n_sen;
    size_t rows = (size_t)gd->n_feat * gd->n_density, r, tot = 0, o = 0;
    uint8 *mixw;
    float *mean, *var, *det;
    int f, d, rc;

    for (f = 0; f < gd->n_feat; ++f)
        tot += (size_t)gd->featlen[f] * gd->n_density;
    mean = ckd_calloc(tot, sizeof(float));
    var = ckd_calloc(tot, sizeof(float));
    det = ckd_calloc(rows, sizeof(float));
    mixw = ckd_malloc(rows * rowlen);
    for (r = 0, f = 0; f < gd->n_feat; ++f)
        for (d = 0; d < gd->n_density; ++d, ++r) {
            memcpy(mean + o, gd->mean[0][f][d], sizeof(float) * gd->featlen[f]);
            memcpy(var + o, gd->var[0][f][d], sizeof(float) * gd->featlen[f]);
            o += gd->featlen[f];
            det[r] = gd->det[0][f][d];
            memcpy(mixw + r * rowlen, s->mixw[f][d], rowlen);
        }
    if (g->smodel)
        psgpu_semi_model_free(g->smodel);
    g->smodel = NULL;
    rc = psgpu_semi_model_create(&g->smodel, gd->n_feat, gd->n_density, gd->featlen, s->n_sen,
                                 s->max_topn, s->ds_ratio, s->topn_beam, mean, var, det, mixw,
                                 s->mixw_cb, (const uint8_t *)la->table, (int32_t)la->table_size);
    ckd_free(mean); ckd_free(var); ckd_free(det); ckd_free(mixw);
    if (rc != PSGPU_OK) {
        E_ERROR("psgpu_semi_model_create failed (%d): %s\n", rc, psgpu_last_error());
        return -1;
    }
    return 0;
}

/* topn_hist / topn_hist_n (s2_semi_mgau.h:83-84) <-> device ring */
static int
semi_sync_history(psgpu_mgau_t *g, int to_device)
{
    s2_semi_mgau_t *s = g->cpu_semi;
    int nf = s->g->n_feat, N = s->max_topn, slot, f, k, rc = 0;
    int32 *cw = ckd_calloc((size_t)nf * N, sizeof(int32));
    int32 *sc = ckd_calloc((size_t)nf * N, sizeof(int32));
    int32 *nu = ckd_calloc(nf, sizeof(int32));

    for (slot = 0; slot < s->n_topn_hist && rc == 0; ++slot) {
        if (!to_device && psgpu_semi_state_get_topn(g->sstate, slot, cw, sc, nu) != PSGPU_OK)
            rc = -1;
        for (f = 0; f < nf && rc == 0; ++f) {
            for (k = 0; k < N; ++k) {
                if (to_device) {
                    cw[f * N + k] = s->topn_hist[slot][f][k].codeword;
                    sc[f * N + k] = s->topn_hist[slot][f][k].score;
                }
                else {
                    s->topn_hist[slot][f][k].codeword = cw[f * N + k];
                    s->topn_hist[slot][f][k].score = sc[f * N + k];
                }
            }
            if (to_device) nu[f] = s->topn_hist_n[slot][f];
            else s->topn_hist_n[slot][f] = (uint8)nu[f];
        }
        if (to_device && psgpu_semi_state_set_topn(g->sstate, slot, cw, sc, nu) != PSGPU_OK)
            rc = -1;
    }
    if (rc < 0)
        E_ERROR("psgpu semi history transfer failed: %s\n", psgpu_last_error());
    ckd_free(cw); ckd_free(sc); ckd_free(nu);
    return rc;
}

static ps_mgau_t *
semi_wrap(ps_mgau_t *cpu_mgau)
{
    psgpu_mgau_t *g = ckd_calloc(1, sizeof(*g));
    s2_semi_mgau_t *s = (s2_semi_mgau_t *)cpu_mgau;
    int f;

    g->base.vt = &psgpu_semi_funcs;
    g->base.frame_idx = cpu_mgau->frame_idx;
    g->cpu_semi = s;
    g->n_feat = s->g->n_feat;
    for (f = 0; f < g->n_feat; ++f)
        g->veclen += s->g->featlen[f];
    g->vec = ckd_calloc(g->veclen, sizeof(float));
    if (semi_upload_model(g) < 0
        || psgpu_semi_state_create(&g->sstate, g->smodel, s->n_topn_hist) != PSGPU_OK
        || semi_sync_history(g, 1) < 0) {
        if (g->smodel && !g->sstate)
            E_ERROR("psgpu_semi_state_create failed: %s\n", psgpu_last_error());
        if (g->sstate) psgpu_semi_state_free(g->sstate);
        if (g->smodel) psgpu_semi_model_free(g->smodel);
        ckd_free(g->vec);
        ckd_free(g);
        return NULL;
    }
    return (ps_mgau_t *)g;
}

static int
semi_frame_eval(ps_mgau_t *ps, int16 *senscr, uint8 *senone_active,
                int32 n_senone_active, mfcc_t **feat, int32 frame, int32 compallsen)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    gauden_t *gd = g->cpu_semi->g;
    int f, o = 0, rc;

    for (f = 0; f < g->n_feat; ++f) {
        memcpy(g->vec + o, feat[f], sizeof(float) * gd->featlen[f]);
        o += gd->featlen[f];
    }
    rc = psgpu_semi_frame_eval(g->sstate, senscr, senone_active, n_senone_active, g->vec,
                               frame, ps->frame_idx, compallsen);
    ++g->n_calls;
    if (rc != PSGPU_OK) {
        E_ERROR("psgpu_semi_frame_eval(frame %d) failed (%d): %s\n", frame, rc, psgpu_last_error());
        return -1;
    }
    return 0;
}

static int
semi_transform(ps_mgau_t *ps, ps_mllr_t *mllr)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    int rc = ps_mgau_transform(ps_mgau_base(g->cpu_semi), mllr);
    if (rc < 0)
        return rc;
    if (semi_sync_history(g, 0) < 0)
        return -1;
    psgpu_semi_state_free(g->sstate);
    g->sstate = NULL;
    if (semi_upload_model(g) < 0)
        return -1;
    if (psgpu_semi_state_create(&g->sstate, g->smodel, g->cpu_semi->n_topn_hist) != PSGPU_OK) {
        E_ERROR("psgpu_semi_state_create failed: %s\n", psgpu_last_error());
        return -1;
    }
    return semi_sync_history(g, 1);
}

static void
semi_free(ps_mgau_t *ps)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    if (g->sstate) psgpu_semi_state_free(g->sstate);
    if (g->smodel) psgpu_semi_model_free(g->smodel);
    if (g->cpu_semi) ps_mgau_free(ps_mgau_base(g->cpu_semi));
    ckd_free(g->vec);
    ckd_free(g);
}

/* ---- "ms": tables of ms_mgau_init() (ms_mgau.c:80-160) ---- */
static int ms_frame_eval(ps_mgau_t *ps, int16 *senscr, uint8 *senone_active,
                         int32 n_senone_active, mfcc_t **feat, int32 frame, int32 compallsen);
static int ms_transform(ps_mgau_t *ps, ps_mllr_t *mllr);
static void ms_free(ps_mgau_t *ps);

static ps_mgaufuncs_t psgpu_ms_funcs = {
    "ms-psgpu",
    ms_frame_eval,
    ms_transform,
    ms_free
};

static int
ms_upload_model(psgpu_mgau_t *g)
{
    ms_mgau_model_t *msg = g->cpu_ms;
    gauden_t *gd = msg->g;
    senone_t *sn = msg->s;
    logadd_t *la = LOGMATH_TABLE(sn->lmath);
    size_t tot = 0, o = 0, od = 0;
    float *mean, *var, *det;
    uint8 *pdf;
    int m, f, d, rc;
    uint32 i;

    for (f = 0; f < gd->n_feat; ++f)
        tot += (size_t)gd->featlen[f];
    tot *= (size_t)gd->n_mgau * gd->n_density;
    mean = ckd_calloc(tot, sizeof(float));
    var = ckd_calloc(tot, sizeof(float));
    det = ckd_calloc((size_t)gd->n_mgau * gd->n_feat * gd->n_density, sizeof(float));
    for (m = 0; m < gd->n_mgau; ++m)
        for (f = 0; f < gd->n_feat; ++f)
            for (d = 0; d < gd->n_density; ++d) {
                memcpy(mean + o, gd->mean[m][f][d], sizeof(float) * gd->featlen[f]);
                memcpy(var + o, gd->var[m][f][d], sizeof(float) * gd->featlen[f]);
                o += gd->featlen[f];
                det[od++] = gd->det[m][f][d];
            }
    /* canonical [sen][feat][cw] whatever the in-memory transposition (ms_senone.c:198-209) */
    pdf = ckd_malloc((size_t)sn->n_sen * sn->n_feat * sn->n_cw);
    for (i = 0; i < sn->n_sen; ++i)
        for (f = 0; (uint32)f < sn->n_feat; ++f)
            for (d = 0; (uint32)d < sn->n_cw; ++d)
                pdf[((size_t)i * sn->n_feat + f) * sn->n_cw + d] =
                    (sn->n_gauden > 1) ? sn->pdf[i][f][d] : sn->pdf[f][d][i];
    if (g->mmodel)
        psgpu_ms_model_free(g->mmodel);
    g->mmodel = NULL;
    rc = psgpu_ms_model_create(&g->mmodel, gd->n_mgau, gd->n_feat, gd->n_density, gd->featlen,
                               (int32_t)sn->n_sen, msg->topn, sn->aw, mean, var, det, pdf, sn->mgau,
                               la->table, (int32_t)la->table_size, la->width,
                               logmath_get_zero(sn->lmath));
    ckd_free(mean); ckd_free(var); ckd_free(det); ckd_free(pdf);
    if (rc != PSGPU_OK) {
        E_ERROR("psgpu_ms_model_create failed (%d): %s\n", rc, psgpu_last_error());
        return -1;
    }
    return 0;
}

static ps_mgau_t *
ms_wrap(ps_mgau_t *cpu_mgau)
{
    psgpu_mgau_t *g = ckd_calloc(1, sizeof(*g));
    int f;

    g->base.vt = &psgpu_ms_funcs;
    g->base.frame_idx = cpu_mgau->frame_idx;
    g->cpu_ms = (ms_mgau_model_t *)cpu_mgau;
    g->n_feat = g->cpu_ms->g->n_feat;
    for (f = 0; f < g->n_feat; ++f)
        g->veclen += g->cpu_ms->g->featlen[f];
    g->vec = ckd_calloc(g->veclen, sizeof(float));
    if (ms_upload_model(g) < 0) {
        ckd_free(g->vec);
        ckd_free(g);
        return NULL;
    }
    return (ps_mgau_t *)g;
}

static int
ms_frame_eval(ps_mgau_t *ps, int16 *senscr, uint8 *senone_active,
              int32 n_senone_active, mfcc_t **feat, int32 frame, int32 compallsen)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    gauden_t *gd = g->cpu_ms->g;
    int f, o = 0, rc;

    for (f = 0; f < g->n_feat; ++f) {
        memcpy(g->vec + o, feat[f], sizeof(float) * gd->featlen[f]);
        o += gd->featlen[f];
    }
    /* The scorer itself ignores `frame` (ms_mgau.c:207); here it keys the look-ahead cache.
     * Full-utterance decoding has every frame's features in acmod->feat_buf before the search
     * starts: announce what lies ahead once, every pass is then answered from that one batch. */
    if (g->acmod && g->acmod->feat_buf && !psgpu_ms_lookahead_covers(g->mmodel, g->vec, frame)
        && frame >= g->acmod->output_frame) {
        acmod_t *a = g->acmod;
        int avail = a->output_frame + a->n_feat_frame - frame, i;
        if (avail >= 8) {
            if (avail > g->la_cap) {
                g->la_buf = ckd_realloc(g->la_buf, sizeof(float) * (size_t)avail * g->veclen);
                g->la_cap = avail;
            }
            for (i = 0; i < avail; ++i) {
                int idx = (a->feat_outidx + (frame - a->output_frame) + i) % a->n_feat_alloc;
                memcpy(g->la_buf + (size_t)i * g->veclen, a->feat_buf[idx][0], sizeof(float) * g->veclen);
            }
            psgpu_ms_lookahead(g->mmodel, g->la_buf, frame, avail);
        }
    }
    rc = psgpu_ms_frame_eval_at(g->mmodel, senscr, senone_active, n_senone_active, g->vec, frame, compallsen);
    ++g->n_calls;
    if (rc != PSGPU_OK) {
        E_ERROR("psgpu_ms_frame_eval failed (%d): %s\n", rc, psgpu_last_error());
        return -1;
    }
    return 0;
}

static int
ms_transform(ps_mgau_t *ps, ps_mllr_t *mllr)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    int rc = ps_mgau_transform(ps_mgau_base(g->cpu_ms), mllr);
    if (rc < 0)
        return rc;
    return ms_upload_model(g);       /* list ids restart at 0 like a fresh msg->dist */
}

static void
ms_free(ps_mgau_t *ps)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    if (g->mmodel) psgpu_ms_model_free(g->mmodel);
    if (g->cpu_ms) ps_mgau_free(ps_mgau_base(g->cpu_ms));
    ckd_free(g->vec);
    ckd_free(g->la_buf);
    ckd_free(g);
}

ps_mgau_t *
psgpu_mgau_wrap(ps_mgau_t *cpu_mgau)
{
    psgpu_mgau_t *g;
    ptm_mgau_t *s = (ptm_mgau_t *)cpu_mgau;
    int f;

    if (cpu_mgau != NULL && strcmp(cpu_mgau->vt->name, "s2_semi") == 0)
        return semi_wrap(cpu_mgau);
    if (cpu_mgau != NULL && strcmp(cpu_mgau->vt->name, "ms") == 0)
        return ms_wrap(cpu_mgau);
    if (cpu_mgau == NULL || strcmp(cpu_mgau->vt->name, "ptm") != 0) {
        E_ERROR("psgpu: only the \"ptm\", \"s2_semi\" and \"ms\" scorers can be wrapped (got \"%s\")\n",
                cpu_mgau ? cpu_mgau->vt->name : "(null)");
        return NULL;
    }
    g = ckd_calloc(1, sizeof(*g));
    g->base.vt = &psgpu_mgau_funcs;
    g->base.frame_idx = cpu_mgau->frame_idx;
    g->cpu = s;
    g->n_feat = s->g->n_feat;
    for (f = 0; f < g->n_feat; ++f)
        g->veclen += s->g->featlen[f];
    g->vec = ckd_calloc(g->veclen, sizeof(float));
    if (upload_model(g) < 0
        || psgpu_ptm_state_create(&g->state, g->model, s->n_fast_hist) != PSGPU_OK
        || import_history(g) < 0) {
        if (g->model && !g->state)
            E_ERROR("psgpu_ptm_state_create failed: %s\n", psgpu_last_error());
        if (g->state) psgpu_ptm_state_free(g->state);
        if (g->model) psgpu_ptm_model_free(g->model);
        ckd_free(g->vec);
        ckd_free(g);
        return NULL;
    }
    return (ps_mgau_t *)g;
}

int
psgpu_mgau_attach(ps_decoder_t *ps)
{
    ps_mgau_t *gpu;

    if (ps == NULL || ps->acmod == NULL || ps->acmod->mgau == NULL)
        return -1;
    gpu = psgpu_mgau_wrap(ps->acmod->mgau);
    if (gpu == NULL)
        return -1;
    if (gpu->vt == &psgpu_mgau_funcs || gpu->vt == &psgpu_ms_funcs)
        ((psgpu_mgau_t *)gpu)->acmod = ps->acmod;
    ps->acmod->mgau = gpu;        /* freed through vt->free by acmod_free (acmod.c:315) */
    return 0;
}

/* One slot of the PTM scorer's history ring := the given codeword lists (cw [n_chain][topn]; their scores are re-computed
 * by eval_topn before they are used, ptm_mgau.c:435-441 + :87-136, so only the codewords and their order matter).  For a
 * search component that scored a pass somewhere else (integration/psgpu_device_decode.c) and hands the scorer back in
 * the state the reference's own pass would have left. */
int
psgpu_mgau_seed_history(ps_mgau_t *ps, int slot, const int32 *cw)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    int32 *sc;
    int rc, n;
    if (ps != NULL && ps->vt == &psgpu_semi_funcs && cw != NULL && slot >= 0 && slot < g->cpu_semi->n_topn_hist) {
        /* the semi-continuous scorer's ring (topn_hist, s2_semi_mgau.c:853-860): the codewords a later frame starts from; their
         * scores are re-derived by mgau_dist, the slot's own count (topn_hist_n) is read by nobody after its frame */
        int32 *nu;
        n = g->cpu_semi->g->n_feat * g->cpu_semi->max_topn;
        sc = ckd_calloc(n, sizeof *sc); nu = ckd_calloc(g->cpu_semi->g->n_feat, sizeof *nu);
        for (rc = 0; rc < g->cpu_semi->g->n_feat; ++rc) nu[rc] = g->cpu_semi->max_topn;
        rc = psgpu_semi_state_set_topn(g->sstate, slot, cw, sc, nu);
        ckd_free(sc); ckd_free(nu);
        return rc == PSGPU_OK ? 0 : -1;
    }
    if (ps == NULL || ps->vt != &psgpu_mgau_funcs || cw == NULL || slot < 0 || slot >= g->cpu->n_fast_hist)
        return -1;
    n = g->cpu->g->n_mgau * g->cpu->g->n_feat * g->cpu->max_topn;
    sc = ckd_calloc(n, sizeof *sc);
    g->la_c0 = g->la_cn = 0; g->la_expect = -1;
    psgpu_ptm_state_lookahead(g->state, NULL, 0, 0);           /* drop a stale look-ahead cache */
    rc = psgpu_ptm_state_set_topn(g->state, slot, cw, sc, NULL);
    ckd_free(sc);
    return rc == PSGPU_OK ? 0 : -1;
}

/* the codewords of history slot `slot` as the wrapped PTM scorer holds them now ([n_mgau * n_feat][topn]) */
int
psgpu_mgau_get_history(ps_mgau_t *ps, int slot, int32 *cw)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    int32 *sc;
    int rc, n;
    if (ps != NULL && ps->vt == &psgpu_semi_funcs && cw != NULL && slot >= 0 && slot < g->cpu_semi->n_topn_hist)
        return psgpu_semi_state_get_topn(g->sstate, slot, cw, NULL, NULL) == PSGPU_OK ? 0 : -1;      /* [n_feat][topn] */
    if (ps == NULL || ps->vt != &psgpu_mgau_funcs || cw == NULL || slot < 0 || slot >= g->cpu->n_fast_hist)
        return -1;
    n = g->cpu->g->n_mgau * g->cpu->g->n_feat * g->cpu->max_topn;
    sc = ckd_calloc(n, sizeof *sc);
    rc = psgpu_ptm_state_get_topn(g->state, slot, cw, sc, NULL);
    ckd_free(sc);
    return rc == PSGPU_OK ? 0 : -1;
}

/* = ptm_mgau_reset_fast_hist (ptm_mgau.c:777-802) for whichever scorer is wrapped:
 * the top-N history a freshly initialised scorer has (the multi-stream scorer is stateless) */
int
psgpu_mgau_reset(ps_mgau_t *ps)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    if (ps == NULL)
        return -1;
    if (ps->vt == &psgpu_mgau_funcs) {
        g->la_c0 = g->la_cn = 0; g->la_expect = -1;
        return psgpu_ptm_state_reset(g->state) == PSGPU_OK ? 0 : -1;
    }
    if (ps->vt == &psgpu_semi_funcs)
        return psgpu_semi_state_reset(g->sstate) == PSGPU_OK ? 0 : -1;
    if (ps->vt == &psgpu_ms_funcs)
        return 0;
    return -1;
}

/* the device model of the wrapped PTM scorer (for components that score whole utterances themselves) */
struct psgpu_ptm_model_s *
psgpu_mgau_ptm_model(ps_mgau_t *ps)
{
    if (ps == NULL || ps->vt != &psgpu_mgau_funcs)
        return NULL;
    return ((psgpu_mgau_t *)ps)->model;
}

struct psgpu_semi_model_s *
psgpu_mgau_semi_model(ps_mgau_t *ps)
{
    if (ps == NULL || ps->vt != &psgpu_semi_funcs)
        return NULL;
    return ((psgpu_mgau_t *)ps)->smodel;
}

struct psgpu_ms_model_s *
psgpu_mgau_ms_model(ps_mgau_t *ps)
{
    if (ps == NULL || ps->vt != &psgpu_ms_funcs)
        return NULL;
    return ((psgpu_mgau_t *)ps)->mmodel;
}

/* For a search component that consumes scores on the device (psgpu_phone_loop_shim.c):
 * announce what lies ahead of `frame` now, have it scored, and hand out the device rows. */
int
psgpu_mgau_prefetch(ps_mgau_t *ps, int frame, const int16_t **raw_dev, const int32_t **best_dev,
                    int *frame0, int *n_frames, int *n_sen)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    int32_t f0 = 0, n = 0;
    if (ps == NULL || ps->vt != &psgpu_mgau_funcs || g->acmod == NULL)
        return -1;
    g->la_expect = -1;                                  /* force a fresh announcement */
    shim_announce(g, frame);
    if (g->la_cn == 0 || g->la_c0 != frame)
        return -1;
    if (psgpu_ptm_state_lookahead_rows(g->state, raw_dev, best_dev, &f0, &n) != PSGPU_OK)
        return -1;
    *frame0 = f0; *n_frames = n; *n_sen = psgpu_ptm_n_sen(g->model);
    return 0;
}

/* the bookkeeping of the fresh all-codebook frame_eval call a device-side phone loop no longer makes */
int
psgpu_mgau_mark_fresh(ps_mgau_t *ps, int frame)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    if (ps == NULL || ps->vt != &psgpu_mgau_funcs)
        return -1;
    if (psgpu_ptm_state_mark_fresh(g->state, frame) != PSGPU_OK)
        return -1;
    g->la_expect = frame + 1;
    return 0;
}

long
psgpu_mgau_n_cache_served(ps_mgau_t *ps)
{
    int64_t a = 0, b = 0;
    if (ps != NULL && ps->vt == &psgpu_ms_funcs) {
        psgpu_ms_lookahead_stats(((psgpu_mgau_t *)ps)->mmodel, &a, &b);
        return (long)a;
    }
    if (ps == NULL || ps->vt != &psgpu_mgau_funcs)
        return 0;
    psgpu_ptm_state_lookahead_stats(((psgpu_mgau_t *)ps)->state, &a, &b);
    return (long)a;
}

int32
psgpu_mgau_n_calls(ps_mgau_t *ps)
{
    if (ps == NULL || (ps->vt != &psgpu_mgau_funcs && ps->vt != &psgpu_semi_funcs && ps->vt != &psgpu_ms_funcs))
        return -1;
    return ((psgpu_mgau_t *)ps)->n_calls;
}

/* Full-utterance decoding puts every frame's features into acmod->feat_buf before
 * the search starts (acmod_process_full_cep, acmod.c:496-528); streaming leaves a
 * few frames there.  Announce whatever lies ahead of `frame` so that the library
 * can score it in one batched pass (psgpu_ptm_state_lookahead). */
static void
shim_announce(psgpu_mgau_t *g, int32 frame)
{
    acmod_t *a = g->acmod;
    int avail, i;

    if (a == NULL || a->feat_buf == NULL)
        return;
    if (frame == g->la_expect && frame >= g->la_c0 && frame < g->la_c0 + g->la_cn) {
        ++g->la_expect;                               /* announced, and arriving in order */
        return;
    }
    g->la_expect = frame + 1;                         /* new utterance / new pass / a jump */
    g->la_c0 = g->la_cn = 0;
    if (frame < a->output_frame)
        return;
    avail = a->output_frame + a->n_feat_frame - frame;
    if (avail < 8) {
        psgpu_ptm_state_lookahead(g->state, NULL, frame, 0);   /* drop a stale cache */
        return;
    }
    if (avail > g->la_cap) {
        g->la_buf = ckd_realloc(g->la_buf, sizeof(float) * (size_t)avail * g->veclen);
        g->la_cap = avail;
    }
    for (i = 0; i < avail; ++i) {
        int idx = (a->feat_outidx + (frame - a->output_frame) + i) % a->n_feat_alloc;
        /* streams of one frame are contiguous (feat_array_alloc, feat/feat.c:356-384) */
        memcpy(g->la_buf + (size_t)i * g->veclen, a->feat_buf[idx][0], sizeof(float) * g->veclen);
    }
    if (psgpu_ptm_state_lookahead(g->state, g->la_buf, frame, avail) == PSGPU_OK) {
        g->la_c0 = frame;
        g->la_cn = avail;
    }
}

/* ps_mgaufuncs_t.frame_eval (acmod.h:101-107), called by acmod_score (acmod.c:1108) */
static int
shim_frame_eval(ps_mgau_t *ps, int16 *senscr, uint8 *senone_active,
                int32 n_senone_active, mfcc_t **feat, int32 frame, int32 compallsen)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    gauden_t *gd = g->cpu->g;
    int f, o = 0, rc;

    for (f = 0; f < g->n_feat; ++f) {
        memcpy(g->vec + o, feat[f], sizeof(float) * gd->featlen[f]);
        o += gd->featlen[f];
    }
    if (frame >= ps->frame_idx)
        shim_announce(g, frame);
    rc = psgpu_ptm_frame_eval(g->state, senscr, senone_active, n_senone_active, g->vec,
                              frame, ps->frame_idx, compallsen);
    ++g->n_calls;
    if (rc != PSGPU_OK) {
        E_ERROR("psgpu_ptm_frame_eval(frame %d) failed (%d): %s\n", frame, rc,
                psgpu_last_error());
        return -1;
    }
    return 0;
}

/* ps_mgaufuncs_t.transform (acmod.h:108-109), called by acmod_update_mllr (acmod.c:329):
 * the reference re-reads and transforms means/variances on the host
 * (ptm_mgau_mllr_transform -> gauden_mllr_transform, ms_gauden.c:511-572);
 * the device copy is then refreshed. */
static int
shim_transform(ps_mgau_t *ps, ps_mllr_t *mllr)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    psgpu_ptm_state_t *old = g->state;
    int rc = ps_mgau_transform(ps_mgau_base(g->cpu), mllr);
    if (rc < 0)
        return rc;
    /* keep the history: read it back into the CPU object's ring first */
    {
        gauden_t *gd = g->cpu->g;
        int n_chain = gd->n_mgau * gd->n_feat, N = g->cpu->max_topn, slot, i;
        int32 *cw = ckd_calloc((size_t)n_chain * N, sizeof(int32));
        int32 *sc = ckd_calloc((size_t)n_chain * N, sizeof(int32));
        uint8 *act = ckd_calloc(gd->n_mgau, 1);
        for (slot = 0; slot < g->cpu->n_fast_hist; ++slot) {
            ptm_topn_t *tl = g->cpu->hist[slot].topn[0][0];
            if (psgpu_ptm_state_get_topn(old, slot, cw, sc, act) != PSGPU_OK)
                break;
            for (i = 0; i < n_chain * N; ++i) {
                tl[i].cw = cw[i];
                tl[i].score = sc[i];
            }
            for (i = 0; i < gd->n_mgau; ++i) {
                if (act[i]) bitvec_set(g->cpu->hist[slot].mgau_active, i);
                else bitvec_clear(g->cpu->hist[slot].mgau_active, i);
            }
        }
        ckd_free(cw); ckd_free(sc); ckd_free(act);
    }
    g->state = NULL;
    g->la_c0 = g->la_cn = 0;
    psgpu_ptm_state_free(old);
    if (upload_model(g) < 0)
        return -1;
    if (psgpu_ptm_state_create(&g->state, g->model, g->cpu->n_fast_hist) != PSGPU_OK) {
        E_ERROR("psgpu_ptm_state_create failed: %s\n", psgpu_last_error());
        return -1;
    }
    return import_history(g);
}

static void
shim_free(ps_mgau_t *ps)
{
    psgpu_mgau_t *g = (psgpu_mgau_t *)ps;
    if (g->state) psgpu_ptm_state_free(g->state);
    if (g->model) psgpu_ptm_model_free(g->model);
    if (g->cpu) ps_mgau_free(ps_mgau_base(g->cpu));
    ckd_free(g->la_buf);
    ckd_free(g->vec);
    ckd_free(g);
}
