/* integration/psgpu_search_shim.c -- REFERENCE-SIDE BINDING.
 *
 * Host side of the batched Viterbi step: gathers the active HMM population of
 * one frame from the reference's own data structures (root channel array,
 * active non-root list, word channel lists, single-phone words; flat-lexicon
 * word chains; the CI phone loop), hands them to psgpu_hmm_vit_eval() as
 * 64-byte records and scatters the results back into the hmm_t objects.
 * No Viterbi arithmetic happens here.
 */
#include <string.h>

#include "psgpu_search_hooks.h"
#include "tmat.h"
#include "bin_mdef.h"
#include "psgpu.h"

typedef struct psgpu_search_ctx_s {
    psgpu_hmm_ctx_t *dev;
    hmm_t **ptr;                 /* population of the current frame */
    psgpu_hmm_rec_t *rec;
    int32 n, cap;
    int n_emit;
    long n_batches, n_hmms;
} psgpu_search_ctx_t;

static psgpu_search_ctx_t *
ctx_new(hmm_context_t *hc, bin_mdef_t *mdef, tmat_t *tmat)
{
    psgpu_search_ctx_t *c;
    int n_emit = hc->n_emit_state, n_tmat = tmat->n_tmat, n_sseq = bin_mdef_n_sseq(mdef);
    uint8 *tp;
    uint16 *sseq;
    int i, a, b, rc;

    if (n_emit != 3 && n_emit != 5) {
        E_ERROR("psgpu: %d-state HMMs are not supported on the device\n", n_emit);
        return NULL;
    }
    tp = ckd_calloc((size_t)n_tmat * n_emit * (n_emit + 1), 1);
    sseq = ckd_calloc((size_t)n_sseq * n_emit, sizeof(uint16));
    for (i = 0; i < n_tmat; ++i)
        for (a = 0; a < n_emit; ++a)
            for (b = 0; b <= n_emit; ++b)
                tp[((size_t)i * n_emit + a) * (n_emit + 1) + b] = hc->tp[i][a][b];
    for (i = 0; i < n_sseq; ++i)
        for (a = 0; a < n_emit; ++a)
            sseq[(size_t)i * n_emit + a] = hc->sseq[i][a];
    c = ckd_calloc(1, sizeof(*c));
    c->n_emit = n_emit;
    rc = psgpu_hmm_ctx_create(&c->dev, n_emit, n_tmat, tp, n_sseq, sseq, bin_mdef_n_sen(mdef));
    ckd_free(tp);
    ckd_free(sseq);
    if (rc != PSGPU_OK) {
        E_ERROR("psgpu_hmm_ctx_create failed (%d): %s\n", rc, psgpu_last_error());
        ckd_free(c);
        return NULL;
    }
    return c;
}

static void
ctx_free(psgpu_search_ctx_t *c)
{
    if (c == NULL)
        return;
    psgpu_hmm_ctx_free(c->dev);
    ckd_free(c->ptr);
    ckd_free(c->rec);
    ckd_free(c);
}

static void
ctx_add(psgpu_search_ctx_t *c, hmm_t *h)
{
    psgpu_hmm_rec_t *r;
    if (c->n == c->cap) {
        c->cap = c->cap ? c->cap * 2 : 1024;
        c->ptr = ckd_realloc(c->ptr, sizeof(*c->ptr) * c->cap);
        c->rec = ckd_realloc(c->rec, sizeof(*c->rec) * c->cap);
    }
    c->ptr[c->n] = h;
    r = &c->rec[c->n++];
    memcpy(r->score, h->score, sizeof r->score);
    memcpy(r->history, h->history, sizeof r->history);
    r->out_score = h->out_score;
    r->out_history = h->out_history;
    r->bestscore = h->bestscore;
    memcpy(r->senid, h->senid, sizeof r->senid);
    r->tmatid_mpx = (uint16)h->tmatid | (h->mpx ? PSGPU_HMM_MPX : 0);
}

static void
ctx_run(psgpu_search_ctx_t *c, int16 const *senscr)
{
    int32 i;
    int rc;
    if (c->n == 0)
        return;
    rc = psgpu_hmm_vit_eval(c->dev, c->rec, c->n, senscr, NULL);
    if (rc != PSGPU_OK)
        E_FATAL("psgpu_hmm_vit_eval failed (%d): %s\n", rc, psgpu_last_error());
    for (i = 0; i < c->n; ++i) {
        hmm_t *h = c->ptr[i];
        psgpu_hmm_rec_t *r = &c->rec[i];
        memcpy(h->score, r->score, sizeof(int32) * c->n_emit);
        memcpy(h->history, r->history, sizeof(int32) * c->n_emit);
        h->out_score = r->out_score;
        h->out_history = r->out_history;
        h->bestscore = r->bestscore;
        if (h->mpx)
            memcpy(h->senid, r->senid, sizeof(uint16) * c->n_emit);
    }
    ++c->n_batches;
    c->n_hmms += c->n;
    c->n = 0;
}

/* evaluate_channels (ngram_search_fwdtree.c:701-715): the populations of
 * eval_root_chan (:605-621), eval_nonroot_chan (:623-642), eval_word_chan
 * (:644-699), same order, same activity tests. */
void
psgpu_fwdtree_pre_evaluate(ngram_search_t *ngs, int16 const *senscr, int frame_idx)
{
    psgpu_search_ctx_t *c = ngs->hmmctx->udata;
    root_chan_t *rhmm;
    chan_t *hmm, **acl;
    int32 i, w, *awl;

    ngs->hmmctx->senscore = senscr;
    if (c == NULL)
        return;
    for (i = ngs->n_root_chan, rhmm = ngs->root_chan; i > 0; --i, rhmm++)
        if (hmm_frame(&rhmm->hmm) == frame_idx)
            ctx_add(c, &rhmm->hmm);
    acl = ngs->active_chan_list[frame_idx & 0x1];
    for (i = 0; i < ngs->n_active_chan[frame_idx & 0x1]; ++i)
        ctx_add(c, &acl[i]->hmm);
    awl = ngs->active_word_list[frame_idx & 0x1];
    for (i = 0; i < ngs->n_active_word[frame_idx & 0x1]; ++i)
        for (hmm = ngs->word_chan[awl[i]]; hmm; hmm = hmm->next)
            ctx_add(c, &hmm->hmm);
    for (i = 0; i < ngs->n_1ph_words; i++) {
        w = ngs->single_phone_wid[i];
        rhmm = (root_chan_t *) ngs->word_chan[w];
        if (hmm_frame(&rhmm->hmm) < frame_idx)
            continue;
        ctx_add(c, &rhmm->hmm);
    }
    ctx_run(c, senscr);
}

/* fwdflat_eval_chan (ngram_search_fwdflat.c:444-480) */
void
psgpu_fwdflat_pre_evaluate(ngram_search_t *ngs, int16 const *senscr, int frame_idx)
{
    psgpu_search_ctx_t *c = ngs->hmmctx->udata;
    root_chan_t *rhmm;
    chan_t *hmm;
    int32 i, nw, *awl;

    ngs->hmmctx->senscore = senscr;
    if (c == NULL)
        return;
    nw = ngs->n_active_word[frame_idx & 0x1];
    awl = ngs->active_word_list[frame_idx & 0x1];
    for (i = 0; i < nw; i++) {
        rhmm = (root_chan_t *) ngs->word_chan[awl[i]];
        if (hmm_frame(&rhmm->hmm) == frame_idx)
            ctx_add(c, &rhmm->hmm);
        for (hmm = rhmm->next; hmm; hmm = hmm->next)
            if (hmm_frame(&hmm->hmm) == frame_idx)
                ctx_add(c, &hmm->hmm);
    }
    ctx_run(c, senscr);
}

/* evaluate_hmms (phone_loop_search.c:202-222) */
void
psgpu_phone_loop_pre_evaluate(phone_loop_search_t *pls, int16 const *senscr, int frame_idx)
{
    psgpu_search_ctx_t *c = pls->hmmctx->udata;
    int i;

    pls->hmmctx->senscore = senscr;
    if (c == NULL)
        return;
    for (i = 0; i < pls->n_phones; ++i) {
        hmm_t *hmm = (hmm_t *)&pls->hmms[i];
        if (hmm_frame(hmm) < frame_idx)
            continue;
        ctx_add(c, hmm);
    }
    ctx_run(c, senscr);
}

/* fsg_search_hmm_eval (fsg_search.c:335-385): the active pnode list */
void
psgpu_fsg_pre_evaluate(fsg_search_t *fsgs, int16 const *senscr)
{
    psgpu_search_ctx_t *c = fsgs->hmmctx->udata;
    gnode_t *gn;

    fsgs->hmmctx->senscore = senscr;
    if (c == NULL)
        return;
    for (gn = fsgs->pnode_active; gn; gn = gnode_next(gn))
        ctx_add(c, fsg_pnode_hmmptr((fsg_pnode_t *) gnode_ptr(gn)));
    ctx_run(c, senscr);
}

/* phmm_eval_all (allphone_search.c:346-375) */
void
psgpu_allphone_pre_evaluate(allphone_search_t *allphs, int16 const *senscr)
{
    psgpu_search_ctx_t *c = allphs->hmmctx->udata;
    bin_mdef_t *mdef = ((ps_search_t *) allphs)->acmod->mdef;
    s3cipid_t ci;
    phmm_t *p;

    allphs->hmmctx->senscore = senscr;
    if (c == NULL)
        return;
    for (ci = 0; ci < mdef->n_ciphone; ci++)
        for (p = allphs->ci_phmm[(unsigned) ci]; p; p = p->next)
            if (hmm_frame(&(p->hmm)) == allphs->frame)
                ctx_add(c, &p->hmm);
    ctx_run(c, senscr);
}

/* kws_search_hmm_eval (kws_search.c:194-226): the phone loop, then active keyphrase HMMs */
void
psgpu_kws_pre_evaluate(kws_search_t *kwss, int16 const *senscr)
{
    psgpu_search_ctx_t *c = kwss->hmmctx->udata;
    gnode_t *gn;
    int32 i;

    kwss->hmmctx->senscore = senscr;
    if (c == NULL)
        return;
    for (i = 0; i < kwss->n_pl; ++i)
        ctx_add(c, &kwss->pl_hmms[i]);
    for (gn = kwss->keyphrases; gn; gn = gnode_next(gn)) {
        kws_keyphrase_t *keyphrase = gnode_ptr(gn);
        for (i = 0; i < keyphrase->n_hmms; i++)
            if (keyphrase->hmms[i].frame > 0)            /* hmm_is_active, kws_search.c:52 */
                ctx_add(c, &keyphrase->hmms[i]);
    }
    ctx_run(c, senscr);
}

/* evaluate_hmms (state_align_search.c:64-84) */
void
psgpu_state_align_pre_evaluate(state_align_search_t *sas, int16 const *senscr, int frame_idx)
{
    psgpu_search_ctx_t *c = sas->hmmctx->udata;
    int i;

    sas->hmmctx->senscore = senscr;
    if (c == NULL)
        return;
    for (i = 0; i < sas->n_phones; ++i) {
        hmm_t *hmm = sas->hmms + i;
        if (hmm_frame(hmm) < frame_idx)
            continue;
        ctx_add(c, hmm);
    }
    ctx_run(c, senscr);
}

/* the hmm_context_t of the active search (n-gram: shared by fwdtree and fwdflat) */
static hmm_context_t *
ngram_hmmctx(ps_decoder_t *ps)
{
    const char *type;
    if (ps->search == NULL)
        return NULL;
    type = ps_search_type(ps->search);
    if (0 == strcmp(type, PS_SEARCH_TYPE_NGRAM))
        return ((ngram_search_t *)ps->search)->hmmctx;
    if (0 == strcmp(type, PS_SEARCH_TYPE_FSG))
        return ((fsg_search_t *)ps->search)->hmmctx;
    if (0 == strcmp(type, PS_SEARCH_TYPE_ALLPHONE))
        return ((allphone_search_t *)ps->search)->hmmctx;
    if (0 == strcmp(type, PS_SEARCH_TYPE_KWS))
        return ((kws_search_t *)ps->search)->hmmctx;
    if (0 == strcmp(type, PS_SEARCH_TYPE_STATE_ALIGN))
        return ((state_align_search_t *)ps->search)->hmmctx;
    return NULL;
}

static hmm_context_t *
pl_hmmctx(ps_decoder_t *ps)
{
    return ps->phone_loop ? ((phone_loop_search_t *)ps->phone_loop)->hmmctx : NULL;
}

int
psgpu_search_attach(ps_decoder_t *ps)
{
    hmm_context_t *nc, *pc;
    psgpu_search_ctx_t *a = NULL, *b = NULL;

    if (ps == NULL || ps->acmod == NULL)
        return -1;
    nc = ngram_hmmctx(ps);
    pc = pl_hmmctx(ps);
    if (pc == nc)
        pc = NULL;
    if (nc == NULL) {
        E_ERROR("psgpu: the active search has no hmm_vit_eval loop known to the hooks\n");
        return -1;
    }
    if (nc->udata || (pc && pc->udata))
        return 0;                                   /* already attached */
    a = ctx_new(nc, ps->acmod->mdef, ps->acmod->tmat);
    if (a == NULL)
        return -1;
    if (pc) {
        b = ctx_new(pc, ps->acmod->mdef, ps->acmod->tmat);
        if (b == NULL) {
            ctx_free(a);
            return -1;
        }
        pc->udata = b;
    }
    nc->udata = a;
    return 0;
}

void
psgpu_search_detach(ps_decoder_t *ps)
{
    hmm_context_t *nc, *pc;
    if (ps == NULL)
        return;
    nc = ngram_hmmctx(ps);
    pc = pl_hmmctx(ps);
    if (nc && nc->udata) { ctx_free(nc->udata); nc->udata = NULL; }
    if (pc && pc->udata) { ctx_free(pc->udata); pc->udata = NULL; }
}

void
psgpu_search_stats(ps_decoder_t *ps, long *n_batches, long *n_hmms)
{
    hmm_context_t *hc[2];
    int i;
    *n_batches = *n_hmms = 0;
    hc[0] = ngram_hmmctx(ps);
    hc[1] = pl_hmmctx(ps);
    for (i = 0; i < 2; ++i)
        if (hc[i] && hc[i]->udata) {
            psgpu_search_ctx_t *c = hc[i]->udata;
            *n_batches += c->n_batches;
            *n_hmms += c->n_hmms;
        }
}
