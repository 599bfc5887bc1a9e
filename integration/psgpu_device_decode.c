/* integration/psgpu_device_decode.c -- REFERENCE-SIDE code.
 *
 * The first pass of the n-gram search entirely on the MI355X, behind the reference's own
 * interfaces.  Three ways in, one device pipeline (psgpu_decode_*, include/psgpu.h):
 *
 *  1. psgpu_device_search_attach(ps): the decoder's n-gram search gets a ps_searchfuncs_t
 *     (pocketsphinx_internal.h:86-97) whose step() only buffers the frame's feature vector and
 *     whose finish() runs scorer -> phone loop -> lexicon-tree search on the device and copies the
 *     back-pointer table, right-context score stack and frame marks into the ngram_search_t in the
 *     reference's own layout (bptbl_t, ngram_search.h:112-124).  The search object IS
 *     the reference's ngram_search_t (type "ngram", ps_get_lm() keeps working), so
 *     unmodified ps_decode_raw() / ps_process_raw() + ps_end_utt() / ps_get_hyp() / ps_seg_iter() /
 *     ps_get_lattice() reach it; with -fwdflat yes the reference's own second pass then runs on the
 *     injected table (ngram_search.c:791-808), with -bestpath yes its lattice pass.
 *  2. psgpu_device_decode_utt(d, pcm, n): one utterance from PCM, front end on the device too.
 *  3. psgpu_device_decode_batch_run(d, pcm[], n[], B) + _select(d, u): B utterances through ONE
 *     launch set (what psgpu_decode_batch uses with PSGPU_BATCH_DEVICE_FIRST_PASS).
 *
 * What is read out of the decoder, once (attach): the search tree create_search_channels built
 * (flattened: roots, then depth-first), single-phone word channels, dictionary and dict2pid
 * tables, beams and penalties, the phone loop's HMMs and beams, and the language model -- the
 * model's own trie when it is one trie model without classes (psgpu_lm_tables.c), else, for small
 * vocabularies, every ngram_tg_score in a dense table.  The scorer is whichever psgpu scorer the decoder carries: PTM or
 * multi-stream ("ms": any -senmgau model, models without a sendump).  Requires the n-gram search with -fwdtree
 * yes, a psgpu scorer (psgpu_mgau_attach first), the 1s_c_d_dd feature type with batch CMN,
 * the phone-loop look-ahead (pl_window > 0); -compallsen yes is served (psgpu_decode_compallsen). */
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "util/ckd_alloc.h"
#include "acmod.h"
#include "ngram_search.h"
#include "ngram_search_fwdtree.h"
#include "ngram_search_fwdflat.h"
#include "phone_loop_search.h"
#include "dict2pid.h"
#include <stddef.h>
#include "lm/ngram_model.h"
#include "fe/fe_internal.h"

#include "psgpu.h"
#include "psgpu_mgau_shim.h"
#include "psgpu_fe_shim.h"
#include "lm/ngram_model_set.h"
#include "psgpu_lm_tables.h"
#include "psgpu_search_tables.h"
#include "psgpu_device_decode.h"

struct psgpu_device_decode_s {
    ps_decoder_t *ps;
    psgpu_fwdtree_t *ft;
    psgpu_lm_t *lm;                    /* the trie on the device (NULL: dense table inside ft) */
    psgpu_lm_t *lm_members[16];        /* an interpolated model set: its members' tries (lm: psgpu_lm_create_interp over them) */
    int n_lm_members;
    psgpu_hmm_ctx_t *ctx;
    psgpu_fe_t *fe;
    psgpu_decode_t *dec;               /* the pipeline (borrows the four above and the attached scorer's model) */
    int n_ci, n_sen, n_chain, topn, cepsize, veclen;
    int n_words_at_attach;
    ngram_model_t *lmset_at_attach;
    /* host side of the results */
    int32_t *h_res, *h_hn;             /* [B][8], [B][4] of the last run */
    int B, cap_B;
    int32_t *h_bp, *h_bss, *h_idx;
    size_t cap_bp, cap_bss, cap_idx;
    /* ps_search_t binding: the frames of the utterance in progress */
    ps_searchfuncs_t vt, pl_vt;
    ps_searchfuncs_t *orig_vt, *orig_pl_vt;
    float *h_feat; int n_feat, cap_feat;
    int pcm_ok;                        /* the feature configuration the device computes from PCM (1s_c_d_dd, batch CMN): the PCM entries work */
    int pl_frames, n_partial;          /* frames the phone loop has been stepped through; n_feat of the latest partial read-out */
    /* the utterance in progress as a LIVE utterance of the device pipeline (psgpu_decode_live_begin / _step): begun by the first
     * read-out in mid-utterance; live_fed frames handed over so far; live_cap the capacity it was begun with; live_off: the pipeline
     * refused (then every read-out decodes the prefix again, as before) */
    int live_on, live_fed, live_cap, live_off;
    int inj_nb, inj_nh, inj_nfr;       /* what the latest read-out of the live utterance put into the decoder's tables */
    /* a member of a group of live decoders (psgpu_live_group_create): stream gidx of the group's pipeline; g_fed frames of the
     * utterance in progress handed over; g_final: its ps_end_utt is in progress (1) / its last frames are on the device (2);
     * g_utts utterances finished */
    struct psgpu_live_group_s *grp;
    int gidx, g_fed, g_final, g_utts, g_dirty, g_done;
    long live_steps, live_restarts;
    /* the second pass on the device as well (PSGPU_DEVICE_SECOND_PASS=1 with -fwdflat yes) */
    psgpu_fwdflat_t *ff;
    psgpu_ptm_view_t view;
    int n_fast_hist, n1, n_emit;
    uint8_t *h_tcw; size_t cap_tcw;
};

/* N decoders, ONE pipeline in streams mode (psgpu_decode_streams_*): see psgpu_live_group_create */
struct psgpu_live_group_s {
    int n, cap, step_cap;
    psgpu_device_decode_t **m;
    psgpu_decode_t *dec;               /* member 0's pipeline object */
    void *st;
    float *feat; int32_t *n_new; uint8_t *fin;
    int32_t *h_res, *h_hn;             /* [n][8], [n][4]: every stream's result record after the latest step */
    long steps;
};
static int group_step(struct psgpu_live_group_s *g);

#define FREE_DEV(p) do { psgpu_free(p); (p) = NULL; } while (0)
#define FREE_HOST(p) do { ckd_free(p); (p) = NULL; } while (0)

/* The vtable functions find their attachment through the vtable itself: the function table a bound search points to
 * (search->vt) is a member of the attachment (vt for the n-gram search, pl_vt for the phone loop), so its address gives the
 * object -- no registry, no limit on the number of decoders, nothing shared between threads that use different decoders (the
 * reference has no user pointer in ps_search_t). */
static int dev_search_step(ps_search_t *search, int frame_idx);
static int dev_phone_loop_step(ps_search_t *search, int frame_idx);
static int live_advance(psgpu_device_decode_t *d, ngram_search_t *ngs, int T, int final);
static psgpu_device_decode_t *
find_attached(ps_search_t *search)
{
    if (search == NULL || search->vt == NULL) return NULL;
    if (search->vt->step == dev_search_step)
        return (psgpu_device_decode_t *)((char *)search->vt - offsetof(psgpu_device_decode_t, vt));
    if (search->vt->step == dev_phone_loop_step)
        return (psgpu_device_decode_t *)((char *)search->vt - offsetof(psgpu_device_decode_t, pl_vt));
    return NULL;
}

psgpu_device_decode_t *
psgpu_device_decode_attach(ps_decoder_t *ps)
{
    psgpu_device_decode_t *d;
    ngram_search_t *ngs;
    acmod_t *acmod;
    bin_mdef_t *mdef;
    phone_loop_search_t *pls;
    psgpu_search_tables_t *st;
    psgpu_fwdtree_tables_t t;
    psgpu_decode_config_t cfg;
    psgpu_ptm_model_t *model;
    struct psgpu_ms_model_s *msmodel;
    struct psgpu_semi_model_s *smodel;
    int pcm_ok;
    int n_ci, n_emit, n_w, i, j, k, lm_ok, want_ff;
    int32 *lm;
    psgpu_fe_shim_t *fes;
    psgpu_lm_tables_t lt;

    if (ps == NULL || ps->search == NULL || ps->acmod == NULL) return NULL;
    if (strcmp(ps_search_type(ps->search), PS_SEARCH_TYPE_NGRAM)) { E_ERROR("psgpu device decode: not an n-gram search\n"); return NULL; }
    ngs = (ngram_search_t *)ps->search;
    if (!ngs->fwdtree) { E_ERROR("psgpu device decode: needs -fwdtree yes (pass 1 is what runs on the device)\n"); return NULL; }
    want_ff = ngs->fwdflat && getenv("PSGPU_DEVICE_SECOND_PASS") && atoi(getenv("PSGPU_DEVICE_SECOND_PASS"));
    acmod = ps->acmod; mdef = acmod->mdef;
    n_ci = bin_mdef_n_ciphone(mdef); n_emit = bin_mdef_n_emit_state(mdef); n_w = dict_size(ps_search_dict(ngs));
    /* from PCM the device computes 1s_c_d_dd vectors with batch CMN (psgpu_fe, psgpu_feat); any other feature configuration is served
     * through the ps_search_t binding alone, which takes the feature vectors the decoder's own acmod computes */
    pcm_ok = !(strcmp(feat_name(acmod->fcb), "1s_c_d_dd") || acmod->fcb->lda || acmod->fcb->cmn != CMN_BATCH
               || acmod->fcb->agc != AGC_NONE || acmod->fcb->varnorm);
    if (acmod->compallsen && want_ff) {
        E_ERROR("psgpu device decode: the device second pass normalises over its own senone lists (-compallsen no)\n");
        return NULL;
    }
    model = psgpu_mgau_ptm_model(acmod->mgau);
    msmodel = model ? NULL : psgpu_mgau_ms_model(acmod->mgau);
    smodel = (model || msmodel) ? NULL : psgpu_mgau_semi_model(acmod->mgau);
    if (model == NULL && msmodel == NULL && smodel == NULL) {
        E_ERROR("psgpu device decode: attach the psgpu scorer first (psgpu_mgau_attach)\n");
        return NULL;
    }
    if (!model && want_ff) { E_ERROR("psgpu device decode: the device second pass scores from the PTM scorer's lists\n"); return NULL; }
    d = ckd_calloc(1, sizeof *d);
    d->ps = ps;
    d->n_ci = n_ci; d->n_sen = bin_mdef_n_sen(mdef);
    d->pcm_ok = pcm_ok;
    /* the lists a session carries: per (codebook, stream) chain for PTM, per stream for s2_semi; ms: none */
    d->n_chain = model ? psgpu_ptm_n_chain(model) : (smodel ? psgpu_semi_n_feat(smodel) : 0);
    d->topn = model ? psgpu_ptm_topn(model) : (smodel ? psgpu_semi_topn(smodel) : 0);
    d->cepsize = feat_cepsize(acmod->fcb);
    for (d->veclen = 0, j = 0; j < feat_dimension1(acmod->fcb); ++j) d->veclen += feat_dimension2(acmod->fcb, j);
    d->n_words_at_attach = n_w; d->lmset_at_attach = ngs->lmset;
    /* ---- the search tables: the one flattener (psgpu_search_tables.c), which psgpu_export_tables.c writes to a file */
    st = psgpu_search_tables_collect(ps, want_ff);
    if (st == NULL) { ckd_free(d); return NULL; }
    /* language scores: the model's own trie on the device when it is one trie model without classes
     * (psgpu_lm_tables.c), else -- small vocabularies only -- every ngram_tg_score in a dense table */
    lm = NULL;
    if (((ngram_model_set_t *)ngs->lmset)->cur >= 0) {
        /* the set's current model (one model, or -lmname / ps_activate_search's choice among an -lmctl file's), word classes included */
        if (psgpu_lm_tables_read(ngs->lmset, &lt) == 0) {
            if (psgpu_lm_create(&d->lm, &lt) != PSGPU_OK) {
                E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
                d->lm = NULL;
            }
            psgpu_lm_tables_release(&lt);
        }
    }
    else {
        /* no current model: the set interpolates its members (ngram_model_set.c:697-714) -- every member's trie on the device, the
         * log-sum over them through the set's log-add table (psgpu_lm_create_interp) */
        psgpu_lm_set_info_t info;
        if (psgpu_lm_set_read(ngs->lmset, &info) == 0 && info.n_models <= 16) {
            int m, ok = 1;
            for (m = 0; m < info.n_models && ok; ++m) {
                ok = psgpu_lm_tables_read_member(ngs->lmset, m, &lt) == 0 && psgpu_lm_create(&d->lm_members[m], &lt) == PSGPU_OK;
                if (ok) ++d->n_lm_members;
                psgpu_lm_tables_release(&lt);
            }
            if (ok && psgpu_lm_create_interp(&d->lm, (const psgpu_lm_t *const *)d->lm_members, info.lweights, info.n_models, info.addtab,
                                             info.addtab_width, info.addtab_size, info.add_zero, info.log_zero) != PSGPU_OK) {
                E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
                d->lm = NULL;
            }
            psgpu_lm_set_release(&info);
        }
    }
    lm_ok = d->lm != NULL || n_w <= 400;
    if (!lm_ok)
        E_ERROR("psgpu device decode: %d words and no trie model -- the dense LM table is for small vocabularies\n", n_w);
    else if (d->lm == NULL)
        lm = psgpu_search_tables_dense_lm(ps, 1);        /* (the back-off cache's fixed point: see there) */
    psgpu_search_tables_view(st, &t, NULL);
    t.lm = lm;
    i = lm_ok ? psgpu_fwdtree_create(&d->ft, &t) : PSGPU_EINVAL;
    if (i == PSGPU_OK && d->lm) i = psgpu_fwdtree_set_lm(d->ft, d->lm);
    if (i == PSGPU_OK && want_ff) {
        /* ---- what the second pass adds: pronunciations as word-internal ssids, the CI phones' ssids, which words the
         *      language model knows, its beams and windows (psgpu_search_tables.c) */
        psgpu_fwdflat_tables_t t2;
        psgpu_fwdtree_tables_t tv;
        psgpu_search_tables_view(st, &tv, &t2);
        t2.ft = &t;
        i = psgpu_fwdflat_create(&d->ff, &t2);
        if (i == PSGPU_OK && d->lm) i = psgpu_fwdflat_set_lm(d->ff, d->lm);
        d->n_fast_hist = ps->pl_window + 2;               /* ptm_mgau.c:884 */
        d->n1 = ngs->n_1ph_words; d->n_emit = n_emit;
    }
    if (i == PSGPU_OK) i = psgpu_hmm_ctx_create(&d->ctx, n_emit, st->n_tmat, st->tp, st->n_sseq, st->sseq, d->n_sen);
    if (i == PSGPU_OK && pcm_ok) {
        fes = psgpu_fe_wrap(acmod->fe);
        if (fes) d->fe = psgpu_fe_shim_release(fes); else i = PSGPU_EINVAL;
    }
    /* ---- the phone loop (cf. psgpu_phone_loop_shim.c) and the pipeline object */
    pls = (phone_loop_search_t *)ps->phone_loop;
    if (i == PSGPU_OK && pls && ps->pl_window > 0 && pls->n_phones <= 64) {
        uint16_t *ps_ssid = ckd_calloc(pls->n_phones, 2), *cil = ckd_calloc(d->n_sen, 2);
        int16_t *ps_tm = ckd_calloc(pls->n_phones, 2);
        uint8 *flags = ckd_calloc(d->n_sen, 1);
        int last = 0, nl = 0;
        for (j = 0; j < pls->n_phones; ++j) {
            hmm_t *h = (hmm_t *)&pls->hmms[j];
            ps_ssid[j] = hmm_nonmpx_ssid(h); ps_tm[j] = (int16_t)h->tmatid;
            for (k = 0; k < n_emit; ++k) flags[hmm_nonmpx_senid(h, k)] = 1;
        }
        for (j = 0; j < d->n_sen; ++j) {
            if (!flags[j]) continue;
            while (j - last > 255) { last += 255; cil[nl++] = (uint16_t)last; }
            cil[nl++] = (uint16_t)j; last = j;
        }
        memset(&cfg, 0, sizeof cfg);
        cfg.fe = d->fe; cfg.model = model; cfg.ctx = d->ctx; cfg.ft = d->ft;
        if (msmodel) { cfg.scorer_kind = PSGPU_SCORER_MS; cfg.scorer = msmodel; }
        else if (smodel) { cfg.scorer_kind = PSGPU_SCORER_SEMI; cfg.scorer = smodel; }
        cfg.pl.n_phones = pls->n_phones; cfg.pl.window = pls->window; cfg.pl.beam = pls->beam; cfg.pl.pbeam = pls->pbeam;
        cfg.pl.pip = pls->pip; cfg.pl.penalty_weight = pls->penalty_weight;
        cfg.pl_ssid = ps_ssid; cfg.pl_tmatid = ps_tm; cfg.ci_list = cil; cfg.n_ci_list = nl; cfg.pl_window = ps->pl_window;
        cfg.max_words = 0;
        i = psgpu_decode_create(&d->dec, &cfg);
        if (i == PSGPU_OK && acmod->compallsen && !smodel) i = psgpu_decode_compallsen(d->dec, 1);      /* -compallsen yes: rows over all senones (s2_semi's are final anyway) */
        ckd_free(ps_ssid); ckd_free(cil); ckd_free(ps_tm); ckd_free(flags);
    }
    else if (i == PSGPU_OK) {
        E_ERROR("psgpu device decode: needs the phone-loop look-ahead (pl_window > 0, <= 64 CI phones)\n");
        i = PSGPU_EINVAL;
    }
    psgpu_search_tables_free(st); ckd_free(lm);
    if (i != PSGPU_OK) {
        E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
        psgpu_device_decode_detach(d);
        return NULL;
    }
    return d;
}

void
psgpu_device_decode_detach(psgpu_device_decode_t *d)
{
    if (!d) return;
    psgpu_device_search_detach(d);
    psgpu_decode_free(d->dec);
    psgpu_fwdflat_free(d->ff); FREE_HOST(d->h_tcw);
    psgpu_fwdtree_free(d->ft); psgpu_lm_free(d->lm); psgpu_hmm_ctx_free(d->ctx); psgpu_fe_free(d->fe);
    { int m_; for (m_ = 0; m_ < d->n_lm_members; ++m_) psgpu_lm_free(d->lm_members[m_]); d->n_lm_members = 0; }
    FREE_HOST(d->h_res); FREE_HOST(d->h_hn); FREE_HOST(d->h_bp); FREE_HOST(d->h_bss); FREE_HOST(d->h_idx); FREE_HOST(d->h_feat);
    ckd_free(d);
}

/* the decoder may have changed since attach: MLLR re-uploads the scorer's tables (psgpu_mgau_shim.c, a new model handle);
 * a dictionary or language-model change invalidates the flattened tables, which is refused */
static int
refresh(psgpu_device_decode_t *d)
{
    ngram_search_t *ngs = (ngram_search_t *)d->ps->search;
    psgpu_ptm_model_t *model = psgpu_mgau_ptm_model(d->ps->acmod->mgau);
    struct psgpu_ms_model_s *msmodel = model ? NULL : psgpu_mgau_ms_model(d->ps->acmod->mgau);
    struct psgpu_semi_model_s *smodel = (model || msmodel) ? NULL : psgpu_mgau_semi_model(d->ps->acmod->mgau);
    if (model == NULL && msmodel == NULL && smodel == NULL) { E_ERROR("psgpu device decode: the psgpu scorer is no longer attached\n"); return -1; }
    if (dict_size(ps_search_dict(ngs)) != d->n_words_at_attach || ngs->lmset != d->lmset_at_attach) {
        E_ERROR("psgpu device decode: the dictionary or the language model changed after attach (ps_add_word / ps_set_lm): "
                "detach and attach again\n");
        return -1;
    }
    if ((model ? psgpu_decode_set_model(d->dec, model) : psgpu_decode_set_scorer(d->dec, msmodel ? (void *)msmodel : (void *)smodel)) != PSGPU_OK
        || (d->ff && psgpu_ptm_model_view(model, &d->view) != PSGPU_OK)) {
        E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
        return -1;
    }
    return 0;
}

static int
fetch_summary(psgpu_device_decode_t *d, int B)
{
    if (B > d->cap_B) {
        FREE_HOST(d->h_res); FREE_HOST(d->h_hn);
        d->h_res = ckd_calloc((size_t)B * 8, 4); d->h_hn = ckd_calloc((size_t)B * 4, 4);
        d->cap_B = B;
    }
    d->B = B;
    if (psgpu_decode_fetch_hyps(d->dec, d->h_hn, NULL, d->h_res, psgpu_hmm_ctx_stream(d->ctx)) != PSGPU_OK) {
        E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
        return -1;
    }
    return 0;
}

/* ngram_search_mark_bptable's growth (ngram_search.c:323-340): the frame marks and, when the second pass exists, its
 * per-frame word lists grow together */
static void
grow_frames(ngram_search_t *ngs, int need)
{
    while (need >= ngs->n_frame_alloc) {
        int old = ngs->n_frame_alloc;
        ngs->n_frame_alloc *= 2;
        ngs->bp_table_idx = (int32 *)ckd_realloc(ngs->bp_table_idx - 1, (ngs->n_frame_alloc + 1) * sizeof(*ngs->bp_table_idx)) + 1;
        if (ngs->frm_wordlist) {
            ngs->frm_wordlist = ckd_realloc(ngs->frm_wordlist, ngs->n_frame_alloc * sizeof(*ngs->frm_wordlist));
            memset(ngs->frm_wordlist + old, 0, (ngs->n_frame_alloc - old) * sizeof(*ngs->frm_wordlist));
        }
    }
}

/* ---- tables in the reference's layout (ngram_search.h:112-124, ngram_search.c:301-339, 445-497).
 *      bp [10][nb] column-major, bss [nh], idx [nfr + 1] on the host. */
/* (b0 / h0 / f0: the decoder already holds entries [0, b0) of the table, [0, h0) of the stack and the marks of frames [0, f0) from an
 *  earlier read-out of the same utterance -- the tables only grow while it is in progress; bp / bss / idx hold the rest) */
static void
inject(ngram_search_t *ngs, int n_ci, const int32_t *bp, int b0, int nb, const int32_t *bss, int h0, int nh, const int32_t *idx, int f0, int nfr,
       int32 best_score)
{
    int i, nn = nb - b0;
    if (nb > ngs->bp_table_size) {
        ngs->bp_table_size = nb + nb / 2;
        ngs->bp_table = ckd_realloc(ngs->bp_table, ngs->bp_table_size * sizeof(*ngs->bp_table));
    }
    if (nh + n_ci >= ngs->bscore_stack_size) {
        ngs->bscore_stack_size = nh + n_ci + nh / 2 + 1;
        ngs->bscore_stack = ckd_realloc(ngs->bscore_stack, ngs->bscore_stack_size * sizeof(*ngs->bscore_stack));
    }
    grow_frames(ngs, nfr + 1);
    for (i = 0; i < nn; ++i) {
        bptbl_t *e = &ngs->bp_table[b0 + i];
#define COL(c) bp[(size_t)(c) * nn + i]
        e->frame = COL(0); e->valid = (uint8)COL(1); e->refcnt = 0; e->wid = COL(2); e->bp = COL(3); e->score = COL(4);
        e->s_idx = COL(5); e->real_wid = COL(6); e->prev_real_wid = COL(7); e->last_phone = (int16)COL(8); e->last2_phone = (int16)COL(9);
#undef COL
    }
    if (nh > h0) memcpy(ngs->bscore_stack + h0, bss, sizeof(int32) * (nh - h0));          /* (nothing new: bss may be NULL) */
    if (nfr + 1 > f0) memcpy(ngs->bp_table_idx + f0, idx, sizeof(int32) * (nfr + 1 - f0));
    ngs->bpidx = nb; ngs->bss_head = nh; ngs->n_frame = nfr;
    ngs->best_score = best_score;    /* ngram_search_lattice (ngram_search.c:1226) refuses an utterance whose best score is WORST_SCORE */
}

/* result word 3 of a search (include/psgpu.h: psgpu_fwdtree_search_dev, psgpu_fwdtree_grow) in words */
static const char *
status_text(int status)
{
    switch (status) {
    case 1: return "status 1: back-pointer table or score stack full (psgpu_decode_table_capacity)";
    case 2: return "status 2: the LDS layout's evaluation list was too short for a frame (psgpu_fwdtree_use_slab_layout)";
    case 3: return "status 3: two last-phone candidates of one frame name the same word (a lexicon tree with two paths to one dictionary entry)";
    case 4: return "status 4: a frame listed more tree nodes than the compact channels hold (psgpu_fwdtree_grow)";
    case 5: return "status 5: the right-context channels' pool ran out of blocks (psgpu_fwdtree_grow)";
    case 6: return "status 6: the word level's counts outgrew its LDS arrays (psgpu_fwdtree_grow)";
    default: return "the search ended with an unknown status";
    }
}

static int
fetch_and_inject_from(psgpu_device_decode_t *d, int u, int b0, int h0, int f0)
{
    ngram_search_t *ngs = (ngram_search_t *)d->ps->search;
    psgpu_decode_t *dec = d->grp ? d->grp->dec : d->dec;         /* (a group's member: stream gidx of the group's pipeline) */
    const int32_t *res = d->grp ? d->grp->h_res + (size_t)d->gidx * 8 : d->h_res + (size_t)u * 8;
    int nb = res[0], nh = res[1], nfr = res[2];
    if (d->grp) u = d->gidx;
    if (res[3]) { E_ERROR("psgpu device decode: utterance %d: %s\n", u, status_text(res[3])); return -1; }
    if (b0 > nb || h0 > nh || f0 > nfr) b0 = h0 = f0 = 0;
    if ((size_t)nb * 10 > d->cap_bp) { FREE_HOST(d->h_bp); d->cap_bp = (size_t)nb * 15 + 640; d->h_bp = ckd_calloc(d->cap_bp, 4); }
    if ((size_t)nh > d->cap_bss) { FREE_HOST(d->h_bss); d->cap_bss = (size_t)nh + nh / 2 + 64; d->h_bss = ckd_calloc(d->cap_bss, 4); }
    if ((size_t)nfr + 1 > d->cap_idx) { FREE_HOST(d->h_idx); d->cap_idx = (size_t)nfr + nfr / 2 + 64; d->h_idx = ckd_calloc(d->cap_idx, 4); }
    if (psgpu_decode_fetch_tables_range(dec, u, b0, nb - b0, h0, nh - h0, f0, nfr + 1 - f0, d->h_bp, d->h_bss, d->h_idx,
                                        d->grp ? d->grp->st : psgpu_hmm_ctx_stream(d->ctx)) != PSGPU_OK) {
        E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
        return -1;
    }
    inject(ngs, d->n_ci, d->h_bp, b0, nb, d->h_bss, h0, nh, d->h_idx, f0, nfr, res[4]);
    return nfr;
}

static int
fetch_and_inject(psgpu_device_decode_t *d, int u)
{
    return fetch_and_inject_from(d, u, 0, 0, 0);
}

/* ---- the second pass on the device: psgpu_decode_second_pass -- the flat-lexicon search over the first pass's device-resident
 *      tables for every utterance of the latest call, scoring its own senones from the call's feature rows, seeded from the lists
 *      pass 1 left in the scorer's history slot n_fast_hist - 1 (ptm_mgau.c:425-441); the pipeline's fetch entry points then return
 *      the second pass's records and tables */
static int
second_pass_batch(psgpu_device_decode_t *d)
{
    if (psgpu_decode_second_pass(d->dec, d->ff, psgpu_hmm_ctx_stream(d->ctx)) != PSGPU_OK) {
        E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
        return -1;
    }
    return 0;
}

int
psgpu_device_decode_utt(psgpu_device_decode_t *d, int16 const *pcm, size_t n_samples)
{
    ps_decoder_t *ps = d->ps;
    ngram_search_t *ngs = (ngram_search_t *)ps->search;
    const int16 *one[1]; size_t n1[1];
    int nfr;

    if (ngs->fwdflat && !d->ff) {
        E_ERROR("psgpu device decode: -fwdflat yes needs PSGPU_DEVICE_SECOND_PASS=1 at attach for this entry (or use "
                "psgpu_device_search_attach + ps_decode_raw: the reference's second pass then runs on the host)\n");
        return -1;
    }
    if (!d->pcm_ok) { E_ERROR("psgpu device decode: from PCM the device computes 1s_c_d_dd vectors with -cmn batch, no AGC / variance normalisation / LDA (psgpu_device_search_attach serves the others)\n"); return -1; }
    if (refresh(d) < 0) return -1;
    /* the reference's own start / end-of-utterance housekeeping, without any frame going through its search */
    if (ps_start_utt(ps) < 0) return -1;
    if (ps_end_utt(ps) < 0) return -1;
    one[0] = pcm; n1[0] = n_samples;
    if (psgpu_decode_first_pass(d->dec, one, n1, 1, psgpu_hmm_ctx_stream(d->ctx)) != PSGPU_OK) {
        E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
        return -1;
    }
    if (fetch_summary(d, 1) < 0) return -1;
    if (d->h_res[3]) { E_ERROR("psgpu device decode: %s\n", status_text(d->h_res[3])); return -1; }
    if (d->h_res[2] == 0) return 0;
    if (d->ff && (second_pass_batch(d) < 0 || fetch_summary(d, 1) < 0)) return -1;
    nfr = fetch_and_inject(d, 0);
    return nfr;
}

int
psgpu_device_decode_batch_run(psgpu_device_decode_t *d, const int16 *const pcm[], const size_t n[], int B)
{
    if (d == NULL || B < 0 || (B > 0 && (!pcm || !n))) return -1;
    if (!d->pcm_ok) { E_ERROR("psgpu device decode: from PCM the device computes 1s_c_d_dd vectors with -cmn batch, no AGC / variance normalisation / LDA (psgpu_device_search_attach serves the others)\n"); return -1; }
    if (((ngram_search_t *)d->ps->search)->fwdflat && !d->ff) {
        E_ERROR("psgpu device decode: -fwdflat yes needs the device second pass (PSGPU_DEVICE_SECOND_PASS=1 at attach: both passes of "
                "the batch then run on the device), or -fwdflat no\n");
        return -1;
    }
    {
    struct timespec t0, t1, t2, t3;
    int rc;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (refresh(d) < 0) return -1;
    d->B = 0;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (psgpu_decode_first_pass(d->dec, pcm, n, B, psgpu_hmm_ctx_stream(d->ctx)) != PSGPU_OK) {
        E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
        return -1;
    }
    if (getenv("PSGPU_BATCH_TIMING")) psgpu_stream_sync(psgpu_hmm_ctx_stream(d->ctx));
    clock_gettime(CLOCK_MONOTONIC, &t2);
    if (d->ff && B > 0 && second_pass_batch(d) < 0) return -1;       /* (-fwdflat yes: the tables injected below are the second pass's) */
    rc = fetch_summary(d, B);
    clock_gettime(CLOCK_MONOTONIC, &t3);
    if (getenv("PSGPU_BATCH_TIMING"))
        fprintf(stderr, "psgpu_device_decode_batch_run: refresh %.1f ms, first pass %.1f ms, second pass + summary %.1f ms\n",
                1e3 * (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_nsec - t0.tv_nsec), 1e3 * (t2.tv_sec - t1.tv_sec) + 1e-6 * (t2.tv_nsec - t1.tv_nsec),
                1e3 * (t3.tv_sec - t2.tv_sec) + 1e-6 * (t3.tv_nsec - t2.tv_nsec));
    return rc;
    }
}

int
psgpu_device_decode_batch_select(psgpu_device_decode_t *d, int u)
{
    ps_decoder_t *ps;
    if (d == NULL || u < 0 || u >= d->B) return -1;
    ps = d->ps;
    /* the reference's own start / end-of-utterance housekeeping (hypothesis, lattice, timers), no frame through its search */
    if (ps_start_utt(ps) < 0) return -1;
    if (ps_end_utt(ps) < 0) return -1;
    if (d->h_res[(size_t)u * 8 + 2] == 0) return 0;
    return fetch_and_inject(d, u);
}

/* the same read-out into ANOTHER decoder of the same configuration -- a worker's, on the worker's thread: utterance u's tables of
 * the latest psgpu_device_decode_batch_run fetched into the caller's staging buffers (grown as needed) over the caller's stream and
 * injected into `ps`'s search; the attached decoder d->ps is not touched (the batch's read-outs run side by side: the lattice and
 * best path of -bestpath yes are the host's share of an utterance, ngram_search.c:782, ps_lattice.c:1216) */
int
psgpu_device_decode_batch_select_into(psgpu_device_decode_t *d, int u, ps_decoder_t *ps, psgpu_dd_stage_t *st, void *stream)
{
    ngram_search_t *ngs;
    const int32_t *res;
    int nb, nh, nfr;
    if (d == NULL || ps == NULL || st == NULL || u < 0 || u >= d->B) return -1;
    if (ps_start_utt(ps) < 0) return -1;
    if (ps_end_utt(ps) < 0) return -1;
    res = d->h_res + (size_t)u * 8;
    nb = res[0]; nh = res[1]; nfr = res[2];
    if (nfr == 0) return 0;
    if (res[3]) { E_ERROR("psgpu device decode: utterance %d: %s\n", u, status_text(res[3])); return -1; }
    ngs = (ngram_search_t *)ps->search;
    if ((size_t)nb * 10 > st->cap_bp) { ckd_free(st->bp); st->cap_bp = (size_t)nb * 15 + 640; st->bp = ckd_calloc(st->cap_bp, 4); }
    if ((size_t)nh > st->cap_bss) { ckd_free(st->bss); st->cap_bss = (size_t)nh + nh / 2 + 64; st->bss = ckd_calloc(st->cap_bss, 4); }
    if ((size_t)nfr + 1 > st->cap_idx) { ckd_free(st->idx); st->cap_idx = (size_t)nfr + nfr / 2 + 64; st->idx = ckd_calloc(st->cap_idx, 4); }
    if (psgpu_decode_fetch_tables_range(d->dec, u, 0, nb, 0, nh, 0, nfr + 1, st->bp, st->bss, st->idx, stream) != PSGPU_OK) {
        E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
        return -1;
    }
    inject(ngs, d->n_ci, st->bp, 0, nb, st->bss, 0, nh, st->idx, 0, nfr, res[4]);
    return nfr;
}

void
psgpu_dd_stage_release(psgpu_dd_stage_t *st)
{
    if (!st) return;
    ckd_free(st->bp); ckd_free(st->bss); ckd_free(st->idx);
    memset(st, 0, sizeof *st);
}

int
psgpu_device_decode_batch_n_frames(psgpu_device_decode_t *d, int u)
{
    psgpu_decode_view_t v;
    if (d == NULL || u < 0 || u >= d->B || psgpu_decode_view(d->dec, &v) != PSGPU_OK) return -1;
    return v.frame_off[u + 1] - v.frame_off[u];
}

/* ---------------------------------------------------------------------------------------------------------------
 * The ps_search_t binding (SURVEY 8b "Search side"): ps_decode_raw() and friends reach the device search.
 * --------------------------------------------------------------------------------------------------------------- */

/* acmod.c:1035-1056 (calc_feat_idx is static there): ring position of a frame's feature vector */
static int
feat_ring_index(acmod_t *acmod, int frame_idx)
{
    int n_backfr = acmod->n_feat_alloc - acmod->n_feat_frame, feat_idx;
    if (frame_idx < 0 || acmod->output_frame - frame_idx > n_backfr) return -1;
    feat_idx = (acmod->feat_outidx + frame_idx - acmod->output_frame) % acmod->n_feat_alloc;
    if (feat_idx < 0) feat_idx += acmod->n_feat_alloc;
    return feat_idx;
}

/* ---- a decoder session through the ps_search_t binding.  The reference decoder's own structures stay the single holder
 * of what an utterance inherits from the one before -- the per-state ssids of the multiplexed permanent channels (hmm_clear
 * keeps them, hmm.c:181-196) and the scorer's history slot n_fast_hist - 1 (it seeds frame 0, ptm_mgau.c:425-441): they are
 * handed to the device pipeline before its pass and written back after it, so that whatever else runs in between on the
 * host (the reference's second pass evaluates the single-phone channels and re-scores the utterance) is carried too. */
static int
session_push(psgpu_device_decode_t *d, ngram_search_t *ngs)
{
    int ne = hmm_n_emit_state(&ngs->root_chan[0].hmm), R = ngs->n_root_chan, n1 = ngs->n_1ph_words, i, k, rc = 0;
    int H = d->ps->pl_window + 2, n = d->n_chain * d->topn;
    int32 *mpx = ckd_calloc((size_t)(R + n1) * ne + 1, 4), *cw = ckd_calloc(n + 1, 4);
    uint8 *seed = ckd_calloc(n + 1, 1);
    for (i = 0; i < R; ++i)
        for (k = 0; k < ne; ++k) mpx[(size_t)i * ne + k] = hmm_mpx_ssid(&ngs->root_chan[i].hmm, k);
    for (i = 0; i < n1; ++i) {
        hmm_t *h = &((root_chan_t *)ngs->word_chan[ngs->single_phone_wid[i]])->hmm;
        for (k = 0; k < ne; ++k) mpx[(size_t)(R + i) * ne + k] = hmm_is_mpx(h) ? hmm_mpx_ssid(h, k) : hmm_nonmpx_ssid(h);
    }
    if (psgpu_mgau_get_history(ps_search_acmod(ngs)->mgau, H - 1, cw) == 0) {
        for (i = 0; i < n; ++i) seed[i] = (uint8)cw[i];
        if (psgpu_decode_session_set(d->dec, seed, mpx, psgpu_hmm_ctx_stream(d->ctx)) != PSGPU_OK) rc = -1;
    }
    else if (psgpu_decode_session_set(d->dec, NULL, mpx, psgpu_hmm_ctx_stream(d->ctx)) != PSGPU_OK) rc = -1;   /* (not the PTM shim) */
    if (rc < 0) E_ERROR("psgpu device search: %s\n", psgpu_last_error());
    ckd_free(mpx); ckd_free(cw); ckd_free(seed);
    return rc;
}

static int
session_pull(psgpu_device_decode_t *d, ngram_search_t *ngs)
{
    int ne = hmm_n_emit_state(&ngs->root_chan[0].hmm), R = ngs->n_root_chan, n1 = ngs->n_1ph_words, i, k, rc = 0;
    int H = d->ps->pl_window + 2, n = d->n_chain * d->topn;
    int32 *mpx = ckd_calloc((size_t)(R + n1) * ne + 1, 4), *cw = ckd_calloc(n + 1, 4), valid = 0;
    uint8 *seed = ckd_calloc(n + 1, 1);
    if (psgpu_decode_session_get(d->dec, seed, &valid, mpx, psgpu_hmm_ctx_stream(d->ctx)) != PSGPU_OK) {
        E_ERROR("psgpu device search: %s\n", psgpu_last_error());
        rc = -1;
    }
    else {
        for (i = 0; i < R; ++i)
            for (k = 0; k < ne; ++k) ngs->root_chan[i].hmm.senid[k] = (uint16)mpx[(size_t)i * ne + k];
        for (i = 0; i < n1; ++i) {
            hmm_t *h = &((root_chan_t *)ngs->word_chan[ngs->single_phone_wid[i]])->hmm;
            if (hmm_is_mpx(h))
                for (k = 0; k < ne; ++k) h->senid[k] = (uint16)mpx[(size_t)(R + i) * ne + k];
        }
        if (valid) {
            for (i = 0; i < n; ++i) cw[i] = seed[i];
            if (psgpu_mgau_seed_history(ps_search_acmod(ngs)->mgau, H - 1, cw) < 0) {
                /* (a scorer other than the PTM shim keeps its own history) */
            }
        }
    }
    ckd_free(mpx); ckd_free(cw); ckd_free(seed);
    return rc;
}

static int
dev_search_start(ps_search_t *search)
{
    psgpu_device_decode_t *d = find_attached(search);
    if (d == NULL) return -1;
    d->n_feat = 0; d->pl_frames = 0; d->n_partial = -1;
    if (d->grp) {                                        /* the decoder's next utterance on its stream of the group's pipeline */
        d->g_fed = 0; d->g_final = 0;
        /* (g_dirty: the stream has been fed frames since it was begun / last moved on -- also by another member's group_step
         *  taking this member's look-ahead frames along; g_done: that utterance was searched to its end, the next one inherits
         *  from it -- otherwise the stream starts its utterance over from what it began with) */
        if (d->g_dirty
            && (d->g_done ? psgpu_decode_streams_next_utt(d->grp->dec, d->gidx, d->grp->st)
                          : psgpu_decode_streams_restart(d->grp->dec, d->gidx, d->grp->st)) != PSGPU_OK) {
            E_ERROR("psgpu live group: %s\n", psgpu_last_error());
            return -1;
        }
        d->g_dirty = 0; d->g_done = 0;
    }
    d->live_on = 0; d->live_fed = 0; d->live_steps = 0; d->live_restarts = 0;
    d->inj_nb = d->inj_nh = d->inj_nfr = 0;
    return d->orig_vt->start(search);          /* ngram_search_start: tables, timers, <s> entered (ngram_search_fwdtree.c:469-520) */
}

/* called by ps_search_forward (pocketsphinx.c:1173-1197) for frame output_frame - pl_window and by ps_end_utt's drain
 * loop (:1329-1333): the frame's feature vector is kept, nothing is searched yet */
static int
dev_search_step(ps_search_t *search, int frame_idx)
{
    psgpu_device_decode_t *d = find_attached(search);
    acmod_t *acmod = ps_search_acmod(search);
    int fi, s;
    float *dst;
    if (d == NULL) return -1;
    if (frame_idx != d->n_feat) { E_ERROR("psgpu device search: frame %d after %d frames\n", frame_idx, d->n_feat); return -1; }
    fi = feat_ring_index(acmod, frame_idx);
    if (fi < 0) { E_ERROR("psgpu device search: frame %d fell out of the feature window\n", frame_idx); return -1; }
    if (d->n_feat == d->cap_feat) {
        d->cap_feat = d->cap_feat ? 2 * d->cap_feat : 1024;
        d->h_feat = ckd_realloc(d->h_feat, (size_t)d->cap_feat * d->veclen * sizeof(float));
    }
    dst = d->h_feat + (size_t)d->n_feat * d->veclen;
    for (s = 0; s < feat_dimension1(acmod->fcb); ++s) {
        memcpy(dst, acmod->feat_buf[fi][s], feat_dimension2(acmod->fcb, s) * sizeof(float));
        dst += feat_dimension2(acmod->fcb, s);
    }
    ++d->n_feat;
    grow_frames((ngram_search_t *)search, d->n_feat + 1);       /* what the reference's per-frame ngram_search_mark_bptable keeps true */
    return 1;
}

static int
dev_search_finish(ps_search_t *search)
{
    psgpu_device_decode_t *d = find_attached(search);
    ngram_search_t *ngs = (ngram_search_t *)search;
    int32_t off[2];
    int nfr = 0, live = 0;
    if (d == NULL) return -1;
    /* the reference's end-of-pass housekeeping on its own (idle) channels and timers; its mark of "one past the last
     * frame" is overwritten by the injected marks below.  ngram_fwdtree_finish marks frame output_frame
     * (ngram_search_fwdtree.c:1505-1507) and ngram_search_mark_bptable doubles the frame marks only ONCE
     * (ngram_search.c:326-338) because the reference's own search calls it every frame; this binding searches nothing on
     * the host, so the marks are grown to the utterance's length here first (dev_search_step grows them as frames arrive
     * too: the reference's per-frame guarantee) */
    grow_frames(ngs, ps_search_acmod(ngs)->output_frame + 1);
    if (d->n_feat + 1 > ps_search_acmod(ngs)->output_frame + 1) grow_frames(ngs, d->n_feat + 1);
    ngram_fwdtree_finish(ngs);
    if (d->n_feat > 0 && d->grp) {
        /* a group's member: its remaining frames go with the other members' pending ones, its search to the utterance's end */
        d->g_final = 1;
        if (group_step(d->grp) < 0) return -1;
        if (d->grp->h_res[(size_t)d->gidx * 8 + 2] > 0 && (nfr = fetch_and_inject_from(d, 0, d->inj_nb, d->inj_nh, d->inj_nfr)) < 0) return -1;
        ++d->g_utts;
        ngs->n_tot_frame += nfr;
        ngs->done = TRUE;
        return 0;
    }
    if (d->n_feat > 0) {
        if (d->live_on) {                                /* the utterance's remaining frames, the search to its end (lag 0) */
            if (live_advance(d, ngs, d->n_feat, 1) <= 0) return -1;
            live = 1;
        }
        else {
        if (refresh(d) < 0) return -1;
        off[0] = 0; off[1] = d->n_feat;
        if (session_push(d, ngs) < 0) return -1;
        if (psgpu_decode_first_pass_feat(d->dec, d->h_feat, off, 1, psgpu_hmm_ctx_stream(d->ctx)) != PSGPU_OK) {
            E_ERROR("psgpu device search: %s\n", psgpu_last_error());
            return -1;
        }
        }
        if (session_pull(d, ngs) < 0) return -1;
        if (fetch_summary(d, 1) < 0) return -1;
        if (d->h_res[2] > 0 && (nfr = live ? fetch_and_inject_from(d, 0, d->inj_nb, d->inj_nh, d->inj_nfr) : fetch_and_inject(d, 0)) < 0) return -1;
    }
    ngs->n_tot_frame += nfr;
    if (ngs->fwdflat && nfr > 0) {
        /* part of what the second pass inherits (ngram_fwdflat_start's hmm_clear keeps them, ngram_search_fwdflat.c:385-392):
         * the per-state ssids the permanent single-phone word channels ended the first pass with */
        psgpu_decode_view_t v;
        int n1 = ngs->n_1ph_words, ne = hmm_n_emit_state(&ngs->root_chan[0].hmm), i, k;
        int32_t *w1 = ckd_calloc((size_t)n1 * ne + 1, 4);
        if (psgpu_decode_view(d->dec, &v) != PSGPU_OK
            || psgpu_memcpy_d2h(w1, v.w1_ssid_dev, 4 * (size_t)n1 * ne, psgpu_hmm_ctx_stream(d->ctx))
            || psgpu_stream_sync(psgpu_hmm_ctx_stream(d->ctx))) {
            E_ERROR("psgpu device search: %s\n", psgpu_last_error());
            ckd_free(w1);
            return -1;
        }
        for (i = 0; i < n1; ++i) {
            hmm_t *h = &((root_chan_t *)ngs->word_chan[ngs->single_phone_wid[i]])->hmm;
            if (hmm_is_mpx(h))
                for (k = 0; k < ne; ++k) h->senid[k] = (uint16)w1[(size_t)i * ne + k];
        }
        ckd_free(w1);
        /* ... and the scorer's history: pass 2's first frame re-scores the lists pass 1 left in slot n_fast_hist - 1
         * (ptm_mgau.c:425-441), i.e. those of the last frame ts with ts % H == H - 1; the batch scorer has them
         * (chain-major [n_chain][T][topn]) */
        if (d->n_chain > 0 && !live) {                  /* (a live utterance: session_pull above has handed exactly these lists over) */
            int H = d->ps->pl_window + 2, T = v.total_frames, ts = T - 1, c;
            size_t ne_all = (size_t)d->n_chain * T * d->topn;
            while (ts >= 0 && ts % H != H - 1) --ts;
            if (ts >= 0) {
                int32 *cw = ckd_calloc((size_t)d->n_chain * d->topn + 1, 4);
                if (ne_all > d->cap_tcw) { FREE_HOST(d->h_tcw); d->cap_tcw = ne_all + ne_all / 2 + 64; d->h_tcw = ckd_calloc(d->cap_tcw, 1); }
                if (psgpu_memcpy_d2h(d->h_tcw, v.topn_cw_dev, ne_all, psgpu_hmm_ctx_stream(d->ctx))
                    || psgpu_stream_sync(psgpu_hmm_ctx_stream(d->ctx))) {
                    E_ERROR("psgpu device search: %s\n", psgpu_last_error());
                    ckd_free(cw);
                    return -1;
                }
                for (c = 0; c < d->n_chain; ++c)
                    for (i = 0; i < d->topn; ++i) cw[c * d->topn + i] = d->h_tcw[((size_t)c * T + ts) * d->topn + i];
                if (psgpu_mgau_seed_history(ps_search_acmod(ngs)->mgau, H - 1, cw) < 0) {
                    E_ERROR("psgpu device search: could not hand the scorer's history over to the second pass\n");
                    ckd_free(cw);
                    return -1;
                }
                ckd_free(cw);
            }
        }
    }
    if (ngs->fwdflat) {
        /* the reference's second pass over the injected table, exactly as ngram_search_finish runs it (ngram_search.c:791-808) */
        int i = 0;
        if (acmod_rewind(ps_search_acmod(ngs)) < 0) return -1;
        ngram_fwdflat_start(ngs);
        while (ps_search_acmod(ngs)->n_feat_frame > 0) {
            int k;
            if ((k = ngram_fwdflat_search(ngs, i)) < 0) return k;
            acmod_advance(ps_search_acmod(ngs));
            ++i;
        }
        ngram_fwdflat_finish(ngs);
    }
    ngs->done = TRUE;
    return 0;
}

/* the look-ahead search of the host has nothing to do: the device pipeline runs its own (psgpu_phone_loop_run_dev); how far
 * it has been stepped is what a partial read-out needs (the frames ahead of the n-gram search are still in acmod's ring) */
static int
dev_phone_loop_step(ps_search_t *search, int frame_idx)
{
    psgpu_device_decode_t *d = find_attached(search);
    if (d && frame_idx + 1 > d->pl_frames) d->pl_frames = frame_idx + 1;
    return 1;
}

/* frame t of the utterance in progress: kept by dev_search_step, or -- the look-ahead frames the phone loop has seen and the
 * n-gram search has not -- still in acmod's ring */
static int
copy_frame(psgpu_device_decode_t *d, acmod_t *acmod, int t, float *dst)
{
    int fi, s;
    if (t < d->n_feat) { memcpy(dst, d->h_feat + (size_t)t * d->veclen, d->veclen * sizeof(float)); return 0; }
    if ((fi = feat_ring_index(acmod, t)) < 0) return -1;
    for (s = 0; s < feat_dimension1(acmod->fcb); ++s) {
        memcpy(dst, acmod->feat_buf[fi][s], feat_dimension2(acmod->fcb, s) * sizeof(float));
        dst += feat_dimension2(acmod->fcb, s);
    }
    return 0;
}

/* The utterance in progress as a live utterance of the pipeline: the frames the pipeline has not seen yet -- up to frame T -- are
 * handed over and its search goes on to the frame the reference's has reached from where it stopped (`final`: to the utterance's end; the reference's does, ngram_search_fwdtree.c:1454-1495;
 * nothing is decoded twice).  Begun at the first call (with the session state the utterance started from); begun AGAIN with twice
 * the capacity, and all frames so far, when the utterance outgrows it.  Returns 1 when the pipeline's tables are the utterance's
 * at this moment, 0 when the pipeline does not do live utterances for this decoder (the caller decodes the prefix), -1 on error. */
static int
live_advance(psgpu_device_decode_t *d, ngram_search_t *ngs, int T, int final)
{
    acmod_t *acmod = ps_search_acmod(ngs);
    void *st = psgpu_hmm_ctx_stream(d->ctx);
    float *feat;
    int t, from, lag;
    if (d->live_off) return 0;
    if (!d->live_on || T > d->live_cap) {
        int cap = d->live_on ? 2 * d->live_cap : 3000;            /* (30 s; doubles) */
        while (cap < T) cap *= 2;
        if (refresh(d) < 0) return -1;
        if (!d->live_on && session_push(d, ngs) < 0) return -1;  /* (a restart begins from the same session state: the pipeline kept it) */
        if (d->live_on) ++d->live_restarts;
        if ((d->live_on ? psgpu_decode_live_restart(d->dec, cap, st) : psgpu_decode_live_begin(d->dec, cap, st)) != PSGPU_OK) {
            E_INFO("psgpu device search: no live utterance on the device (%s); results in mid-utterance decode the frames so far\n", psgpu_last_error());
            d->live_off = 1; d->live_on = 0;
            return 0;
        }
        d->live_on = 1; d->live_cap = cap; d->live_fed = 0;
    }
    from = d->live_fed;
    feat = ckd_calloc((size_t)(T - from) * d->veclen + 1, sizeof(float));
    for (t = from; t < T; ++t)
        if (copy_frame(d, acmod, t, feat + (size_t)(t - from) * d->veclen) < 0) { T = t; break; }
    /* (mid-utterance: the n-gram search has been stepped through n_feat frames, the phone loop through T -- pl_window frames
     *  more (pocketsphinx.c:1173-1197).  Should look-ahead frames be missing (copy_frame failed above), the device search stays
     *  pl_window frames behind what it was given rather than searching frames with clamped look-ahead penalties) */
    lag = final ? 0 : (T > d->n_feat ? T - d->n_feat : 0);
    /* (a step that is not the utterance's last never searches closer than pl_window frames to what it was given -- also when
     *  copy_frame took frames away and T fell to n_feat or below: the search portion of the step is then empty) */
    if (!final && lag < d->ps->pl_window) lag = d->ps->pl_window;
    if (psgpu_decode_live_step(d->dec, feat, T - from, lag, st) != PSGPU_OK) {
        E_ERROR("psgpu device search (live utterance): %s\n", psgpu_last_error());
        ckd_free(feat);
        return -1;
    }
    ckd_free(feat);
    d->live_fed = T; ++d->live_steps;
    return 1;
}

/* Results in mid-utterance (ps_get_hyp / ps_seg_iter between ps_process_raw calls, pocketsphinx.c:1372, ngram_search.c:845):
 * the reference's search has stepped through n_feat frames by now and its phone loop through up to pl_window more; the same
 * state on the device is the first pass over the frames seen so far with the search stopped n frames short
 * (psgpu_decode_search_lag), from the session state the utterance started with; its tables go where the reference's
 * read-out looks.  The utterance's final pass (dev_search_finish) starts over from the same session state. */
static int
partial_refresh(psgpu_device_decode_t *d, ngram_search_t *ngs)
{
    acmod_t *acmod = ps_search_acmod(ngs);
    int T = d->pl_frames > d->n_feat ? d->pl_frames : d->n_feat, t, s;
    int32_t off[2];
    float *feat;
    if (d->grp) {
        /* a group's member: one step of the group's pipeline if this decoder has frames the device has not seen (the other members'
         * pending frames go with them), then what its tables have grown by */
        const int32_t *res = d->grp->h_res + (size_t)d->gidx * 8;
        if (d->n_feat == 0) return 0;
        if (T > d->g_fed && group_step(d->grp) < 0) return -1;
        if (res[2] > 0) {
            if (fetch_and_inject_from(d, 0, d->inj_nb, d->inj_nh, d->inj_nfr) < 0) return -1;
            d->inj_nb = res[0]; d->inj_nh = res[1]; d->inj_nfr = res[2];
        }
        return 0;
    }
    if (d->n_feat == 0 || d->n_partial == d->n_feat) return 0;
    if ((t = live_advance(d, ngs, T, 0)) != 0) {
        if (t < 0 || fetch_summary(d, 1) < 0) return -1;
        /* (what the previous read-out of this utterance injected is still in the decoder: the rest) */
        if (d->h_res[2] > 0) {
            if (fetch_and_inject_from(d, 0, d->inj_nb, d->inj_nh, d->inj_nfr) < 0) return -1;
            d->inj_nb = d->h_res[0]; d->inj_nh = d->h_res[1]; d->inj_nfr = d->h_res[2];
        }
        d->n_partial = d->n_feat;
        return 0;
    }
    feat = ckd_calloc((size_t)T * d->veclen + 1, sizeof(float));
    memcpy(feat, d->h_feat, (size_t)d->n_feat * d->veclen * sizeof(float));
    for (t = d->n_feat; t < T; ++t) {                    /* the look-ahead frames: still in the ring */
        int fi = feat_ring_index(acmod, t);
        float *dst = feat + (size_t)t * d->veclen;
        if (fi < 0) { T = t; break; }
        for (s = 0; s < feat_dimension1(acmod->fcb); ++s) {
            memcpy(dst, acmod->feat_buf[fi][s], feat_dimension2(acmod->fcb, s) * sizeof(float));
            dst += feat_dimension2(acmod->fcb, s);
        }
    }
    off[0] = 0; off[1] = T;
    if (refresh(d) < 0 || session_push(d, ngs) < 0
        || psgpu_decode_search_lag(d->dec, T - d->n_feat) != PSGPU_OK
        || psgpu_decode_first_pass_feat(d->dec, feat, off, 1, psgpu_hmm_ctx_stream(d->ctx)) != PSGPU_OK) {
        E_ERROR("psgpu device search (partial result): %s\n", psgpu_last_error());
        ckd_free(feat);
        return -1;
    }
    ckd_free(feat);
    if (fetch_summary(d, 1) < 0) return -1;
    if (d->h_res[2] > 0 && fetch_and_inject(d, 0) < 0) return -1;
    d->n_partial = d->n_feat;
    return 0;
}

static char const *
dev_search_hyp(ps_search_t *search, int32 *out_score)
{
    psgpu_device_decode_t *d = find_attached(search);
    ngram_search_t *ngs = (ngram_search_t *)search;
    if (d == NULL) return NULL;
    if (!ngs->done && partial_refresh(d, ngs) < 0) return NULL;
    return d->orig_vt->hyp(search, out_score);
}

static ps_seg_t *
dev_search_seg_iter(ps_search_t *search)
{
    psgpu_device_decode_t *d = find_attached(search);
    ngram_search_t *ngs = (ngram_search_t *)search;
    if (d == NULL) return NULL;
    if (!ngs->done && partial_refresh(d, ngs) < 0) return NULL;
    return d->orig_vt->seg_iter(search);
}

/* ---- a group of live decoders -------------------------------------------------------------------------------------------------
 * N decoders whose n-gram searches are bound to the device (psgpu_device_search_attach each), ONE pipeline object in streams mode
 * (member 0's; psgpu_decode_streams_*): a step hands over every member's frames the device has not seen -- the search's own and the
 * phone loop's look-ahead, as live_advance does for one decoder -- in one launch set; every member's hyp / seg_iter then read its
 * tables as they stand.  A member's ps_end_utt ends its stream's utterance, its next ps_start_utt goes on as the same decoder
 * (psgpu_decode_streams_next_utt).  -fwdflat no (the reference's second pass reads host state the group does not keep up to date). */
static int
group_step(struct psgpu_live_group_s *g)
{
    int again;
    do {
        size_t at = 0;
        int u;
        again = 0;
        for (u = 0; u < g->n; ++u) {
            psgpu_device_decode_t *d = g->m[u];
            acmod_t *acmod = ps_search_acmod(d->ps->search);
            int T = d->g_final ? d->n_feat : (d->pl_frames > d->n_feat ? d->pl_frames : d->n_feat);
            int k = T - d->g_fed, t, whole = 1;
            if (d->g_final == 2 || k < 0) k = 0;
            if (k > g->step_cap) { k = g->step_cap; again = 1; whole = 0; }
            for (t = 0; t < k; ++t)
                if (copy_frame(d, acmod, d->g_fed + t, g->feat + (at + t) * d->veclen) < 0) { k = t; whole = 0; break; }
            g->n_new[u] = k;
            g->fin[u] = (uint8_t)(d->g_final == 1 && whole);
            at += (size_t)k;
        }
        if (psgpu_decode_streams_step(g->dec, g->feat, g->n_new, g->fin, g->st) != PSGPU_OK) {
            E_ERROR("psgpu live group: %s\n", psgpu_last_error());
            return -1;
        }
        for (u = 0; u < g->n; ++u) {
            g->m[u]->g_fed += g->n_new[u];
            if (g->n_new[u] > 0) g->m[u]->g_dirty = 1;
            if (g->fin[u]) { g->m[u]->g_final = 2; g->m[u]->g_done = 1; }
        }
    } while (again);
    if (psgpu_decode_fetch_hyps(g->dec, g->h_hn, NULL, g->h_res, g->st) != PSGPU_OK) {
        E_ERROR("psgpu live group: %s\n", psgpu_last_error());
        return -1;
    }
    ++g->steps;
    return 0;
}

int
psgpu_live_group_step(struct psgpu_live_group_s *g)
{
    return g ? group_step(g) : -1;
}

struct psgpu_live_group_s *
psgpu_live_group_create(psgpu_device_decode_t *const *members, int n, int max_frames, int max_step_frames)
{
    struct psgpu_live_group_s *g;
    int u;
    if (members == NULL || n < 1 || max_frames < 1 || max_step_frames < 1) return NULL;
    for (u = 0; u < n; ++u) {
        psgpu_device_decode_t *d = members[u];
        if (d == NULL || d->orig_vt == NULL || d->grp || d->veclen != members[0]->veclen || d->n_sen != members[0]->n_sen) {
            E_ERROR("psgpu live group: member %d is not a decoder bound with psgpu_device_search_attach, is in a group already, or has another model\n", u);
            return NULL;
        }
        if (((ngram_search_t *)d->ps->search)->fwdflat) {
            E_ERROR("psgpu live group: -fwdflat no (member %d)\n", u);
            return NULL;
        }
    }
    g = ckd_calloc(1, sizeof *g);
    g->n = n; g->cap = max_frames; g->step_cap = max_step_frames;
    g->m = ckd_calloc(n, sizeof *g->m);
    g->dec = members[0]->dec; g->st = psgpu_hmm_ctx_stream(members[0]->ctx);
    g->feat = ckd_calloc((size_t)n * max_step_frames * members[0]->veclen + 1, sizeof(float));
    g->n_new = ckd_calloc(n, sizeof *g->n_new); g->fin = ckd_calloc(n, 1);
    g->h_res = ckd_calloc((size_t)n * 8, 4); g->h_hn = ckd_calloc((size_t)n * 4, 4);
    if (refresh(members[0]) < 0 || psgpu_decode_streams_begin(g->dec, n, max_frames, max_step_frames, g->st) != PSGPU_OK) {
        E_ERROR("psgpu live group: %s\n", psgpu_last_error());
        ckd_free(g->m); ckd_free(g->feat); ckd_free(g->n_new); ckd_free(g->fin); ckd_free(g->h_res); ckd_free(g->h_hn); ckd_free(g);
        return NULL;
    }
    for (u = 0; u < n; ++u) {
        g->m[u] = members[u];
        members[u]->grp = g; members[u]->gidx = u; members[u]->g_fed = 0; members[u]->g_final = 0; members[u]->g_utts = 0; members[u]->g_dirty = 0; members[u]->g_done = 0;
    }
    return g;
}

void
psgpu_live_group_free(struct psgpu_live_group_s *g)
{
    int u;
    if (g == NULL) return;
    for (u = 0; u < g->n; ++u) if (g->m[u]) g->m[u]->grp = NULL;
    ckd_free(g->m); ckd_free(g->feat); ckd_free(g->n_new); ckd_free(g->fin); ckd_free(g->h_res); ckd_free(g->h_hn);
    ckd_free(g);
}

long
psgpu_live_group_stats(struct psgpu_live_group_s *g, long *steps)
{
    if (steps) *steps = g ? g->steps : 0;
    return g ? (long)psgpu_decode_live_frames_searched(g->dec) : 0;
}

void
psgpu_device_search_live_stats(psgpu_device_decode_t *d, long *frames_searched, long *steps, long *restarts)
{
    if (frames_searched) *frames_searched = d ? (long)psgpu_decode_live_frames_searched(d->dec) : 0;
    if (steps) *steps = d ? d->live_steps : 0;
    if (restarts) *restarts = d ? d->live_restarts : 0;
}

int
psgpu_device_search_attach(psgpu_device_decode_t *d)
{
    ps_search_t *s, *pl;
    ngram_search_t *ngs;
    if (d == NULL || d->orig_vt) return -1;
    s = d->ps->search; pl = d->ps->phone_loop;
    ngs = (ngram_search_t *)s;
    if (ngs->fwdflat && d->ff) {
        E_ERROR("psgpu device search: with PSGPU_DEVICE_SECOND_PASS=1 use psgpu_device_decode_utt (the vtable binding runs the "
                "reference's own second pass)\n");
        return -1;
    }
    if (psgpu_decode_session(d->dec, 1) != PSGPU_OK) {        /* one decoder, one utterance after another: see session_push */
        E_ERROR("psgpu device search: %s\n", psgpu_last_error());
        return -1;
    }
    d->orig_vt = s->vt; d->vt = *s->vt;
    d->vt.start = dev_search_start; d->vt.step = dev_search_step; d->vt.finish = dev_search_finish;
    d->vt.hyp = dev_search_hyp; d->vt.seg_iter = dev_search_seg_iter;     /* (the reference's, behind a mid-utterance refresh) */
    s->vt = &d->vt;                              /* reinit / free / lattice / prob stay the reference's */
    if (pl) {
        d->orig_pl_vt = pl->vt; d->pl_vt = *pl->vt;
        d->pl_vt.step = dev_phone_loop_step;
        pl->vt = &d->pl_vt;
    }
    return 0;
}

void
psgpu_device_search_detach(psgpu_device_decode_t *d)
{
    if (d == NULL || d->orig_vt == NULL) return;
    if (d->ps->search && d->ps->search->vt == &d->vt) d->ps->search->vt = d->orig_vt;
    if (d->ps->phone_loop && d->orig_pl_vt && d->ps->phone_loop->vt == &d->pl_vt) d->ps->phone_loop->vt = d->orig_pl_vt;
    d->orig_vt = NULL; d->orig_pl_vt = NULL;
}
