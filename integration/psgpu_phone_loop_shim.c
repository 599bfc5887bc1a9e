/* integration/psgpu_phone_loop_shim.c -- REFERENCE-SIDE code.
 *
 * The phone-loop search (phone_loop_search.c) runs pl_window frames ahead of the
 * n-gram search and only produces pls->penalties, which fwdtree reads through
 * phone_loop_search_score() (phone_loop_search.h:99).  In full-utterance decoding all
 * features are in acmod->feat_buf before the first step, the psgpu scorer scores them
 * in one batched pass, and the whole phone loop of the utterance is ONE device launch
 * on those rows (psgpu_phone_loop_run_dev).  This file swaps the search's vtable
 * `step` (ps_searchfuncs_t, pocketsphinx_internal.h:86-97): step(t) copies the device
 * result of frame t into pls->penalties and tells the scorer that the fresh
 * frame_eval call of frame t (which the reference's step would have made, and which
 * fwdtree's call for frame t five frames later relies on) is accounted for.  Whenever
 * that is not possible the reference's own step runs, after the HMMs and the penalty
 * ring have been loaded with the device's state of the previous frame. */
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "util/ckd_alloc.h"
#include "acmod.h"
#include "hmm.h"
#include "phone_loop_search.h"

#include "psgpu.h"
#include "psgpu_mgau_shim.h"
#include "psgpu_phone_loop_shim.h"

/* device memory through the C ABI's own helpers: a C host needs no HIP headers */
#define dev_alloc(pp, n) psgpu_malloc((void **)(pp), (n))
#define h2d(dst, src, n) psgpu_memcpy_h2d((dst), (src), (n), NULL)
#define d2h_on(st, dst, src, n) psgpu_memcpy_d2h((dst), (src), (n), (st))

typedef struct pl_dev_s {
    ps_search_t *search;
    ps_searchfuncs_t vt, *orig;
    psgpu_hmm_ctx_t *ctx;
    psgpu_phone_loop_params_t par;
    uint16_t *d_ssid, *d_ci; int16_t *d_tmatid;
    int n_list, n_emit;
    int32_t *d_off, *d_pen, *d_now, *d_state;   /* device results of the current utterance */
    int32_t *pen, *now, *state;                 /* host copies */
    int cap, n_frames, active, host_ready, have_state;
    long n_device, n_host;
    struct pl_dev_s *next;
} pl_dev_t;

static pl_dev_t *g_list;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

static pl_dev_t *
find(ps_search_t *search)
{
    pl_dev_t *d;
    pthread_mutex_lock(&g_lock);
    for (d = g_list; d && d->search != search; d = d->next) ;
    pthread_mutex_unlock(&g_lock);
    return d;
}

/* one launch for the whole utterance; 0 if the results are on the host */
static int
run_utterance(pl_dev_t *d, acmod_t *acmod)
{
    const int16_t *raw; const int32_t *best;
    int f0, n, n_sen;
    int32_t off[2];
    void *st;
    size_t np = d->par.n_phones;

    if (psgpu_mgau_prefetch(acmod->mgau, 0, &raw, &best, &f0, &n, &n_sen) < 0 || f0 != 0)
        return -1;
    if (n > d->cap) {
        /* grow in big steps: device and page-locked allocations are expensive and synchronising */
        int cap = n < 1024 ? 1024 : n + n / 2;
        psgpu_free(d->d_pen); psgpu_free(d->d_now); psgpu_free(d->d_state);
        psgpu_host_free(d->pen); psgpu_host_free(d->now); psgpu_host_free(d->state);
        d->d_pen = d->d_now = d->d_state = NULL;
        d->pen = d->now = d->state = NULL;
        d->cap = 0;
        if (dev_alloc(&d->d_pen, sizeof(int32_t) * cap * np) || dev_alloc(&d->d_now, sizeof(int32_t) * cap * np)
            || dev_alloc(&d->d_state, sizeof(int32_t) * cap * np * 8)
            || psgpu_host_alloc((void **)&d->pen, sizeof(int32_t) * cap * np)
            || psgpu_host_alloc((void **)&d->now, sizeof(int32_t) * cap * np)
            || psgpu_host_alloc((void **)&d->state, sizeof(int32_t) * cap * np * 8))
            return -1;
        d->cap = cap;
    }
    off[0] = 0; off[1] = n;
    st = psgpu_hmm_ctx_stream(d->ctx);            /* this decoder's own stream: decoders do not serialise */
    if (psgpu_memcpy_h2d(d->d_off, off, sizeof off, st)) return -1;
    if (psgpu_phone_loop_run_dev(d->ctx, &d->par, d->d_ssid, d->d_tmatid,
                                 acmod->compallsen ? NULL : d->d_ci, acmod->compallsen ? 0 : d->n_list,
                                 raw, n_sen, acmod->compallsen ? best : NULL, d->d_off, 1, n,
                                 d->d_pen, d->d_now, d->d_state, st) != PSGPU_OK) {
        E_ERROR("psgpu_phone_loop_run_dev: %s\n", psgpu_last_error());
        return -1;
    }
    /* the penalties are all a decode needs; the ring entries and HMM states stay on the device
     * unless the reference's own step has to take over (load_host_state) */
    if (d2h_on(st, d->pen, d->d_pen, sizeof(int32_t) * n * np) || psgpu_stream_sync(st))
        return -1;
    d->n_frames = n;
    d->have_state = 0;
    return 0;
}

/* give the reference's HMMs and penalty ring the state the device left after frame t-1 */
static void
load_host_state(pl_dev_t *d, phone_loop_search_t *pls, int t)
{
    int i, k, w;
    size_t np = d->par.n_phones;
    if (t == 0) return;                           /* phone_loop_search_start's state is the right one */
    if (!d->have_state) {
        void *st = psgpu_hmm_ctx_stream(d->ctx);
        if (d2h_on(st, d->now, d->d_now, sizeof(int32_t) * d->n_frames * np)
            || d2h_on(st, d->state, d->d_state, sizeof(int32_t) * d->n_frames * np * 8) || psgpu_stream_sync(st))
            E_FATAL("psgpu phone loop: cannot read the device state back: %s\n", psgpu_last_error());
        d->have_state = 1;
    }
    for (i = 0; i < pls->n_phones; ++i) {
        hmm_t *h = (hmm_t *)&pls->hmms[i];
        const int32_t *st = d->state + ((size_t)(t - 1) * np + i) * 8;
        for (k = 0; k < d->n_emit; ++k) hmm_score(h, k) = st[k];
        hmm_out_score(h) = st[5]; h->bestscore = st[6]; hmm_frame(h) = st[7];
    }
    /* store_scores (phone_loop_search.c:223-245) has run t times: ring entry (k mod window) = frame k */
    for (w = 0; w < pls->window; ++w)
        memset(pls->pen_buf[w], 0, sizeof(int32) * pls->n_phones);
    for (k = (t > pls->window ? t - pls->window : 0); k < t; ++k)
        memcpy(pls->pen_buf[k % pls->window], d->now + (size_t)k * np, sizeof(int32) * pls->n_phones);
    pls->pen_buf_ptr = (int16)(t % pls->window);
    memcpy(pls->penalties, d->pen + (size_t)(t - 1) * np, sizeof(int32) * pls->n_phones);
    /* pls->best_score of frame t-1: the best of the HMMs' bestscores */
    {
        int32 bs = WORST_SCORE;
        const int32_t *st = d->state + (size_t)(t - 1) * np * 8;
        (void)st;
        /* bestscore fields of pruned HMMs were cleared, the survivors hold the frame's values; the
         * frame's maximum is the bestscore of whichever HMM had it, and that one survives its own beam */
        for (i = 0; i < pls->n_phones; ++i)
            if (d->state[((size_t)(t - 1) * np + i) * 8 + 6] BETTER_THAN bs)
                bs = d->state[((size_t)(t - 1) * np + i) * 8 + 6];
        pls->best_score = bs;
    }
}

static int
pl_step(ps_search_t *search, int frame_idx)
{
    pl_dev_t *d = find(search);
    phone_loop_search_t *pls = (phone_loop_search_t *)search;
    acmod_t *acmod = ps_search_acmod(search);

    if (frame_idx == 0) {
        d->active = (run_utterance(d, acmod) == 0);
        d->host_ready = 1;                        /* frame 0: the host state is phone_loop_search_start's */
    }
    {   /* test hook: PSGPU_PL_BREAK_AT=t makes frame t fall back to the host mid-utterance */
        static int break_at = -2;
        if (break_at == -2) { const char *e = getenv("PSGPU_PL_BREAK_AT"); break_at = e ? atoi(e) : -1; }
        if (d->active && frame_idx == break_at) { load_host_state(d, pls, frame_idx); d->host_ready = 1; d->active = 0; }
    }
    if (d->active && frame_idx < d->n_frames && psgpu_mgau_mark_fresh(acmod->mgau, frame_idx) == 0) {
        memcpy(pls->penalties, d->pen + (size_t)frame_idx * d->par.n_phones, sizeof(int32) * pls->n_phones);
        d->host_ready = 0;
        ++d->n_device;
        return 0;
    }
    if (d->active && !d->host_ready) {
        load_host_state(d, pls, frame_idx);
        d->host_ready = 1;
    }
    d->active = 0;
    ++d->n_host;
    return d->orig->step(search, frame_idx);
}

int
psgpu_phone_loop_attach(ps_decoder_t *ps)
{
    phone_loop_search_t *pls;
    acmod_t *acmod;
    pl_dev_t *d;
    uint16_t *ssid, *ci; int16_t *tm;
    uint8 *flags;
    int i, k, n_sen, n_emit, last, n_list = 0;

    if (ps == NULL || ps->phone_loop == NULL || ps->acmod == NULL) return -1;
    if (find(ps->phone_loop)) return -1;
    pls = (phone_loop_search_t *)ps->phone_loop;
    acmod = ps->acmod;
    n_emit = bin_mdef_n_emit_state(acmod->mdef);
    n_sen = bin_mdef_n_sen(acmod->mdef);
    if (pls->n_phones < 1 || pls->n_phones > 64 || pls->window < 1 || pls->window > 32 || (n_emit != 3 && n_emit != 5))
        return -1;
    d = ckd_calloc(1, sizeof *d);
    d->search = ps->phone_loop;
    d->n_emit = n_emit;
    {
        /* transition matrices and senone sequences, as psgpu_search_shim.c hands them over */
        int n_tmat = acmod->tmat->n_tmat, n_sseq = bin_mdef_n_sseq(acmod->mdef), s;
        uint8 *tp = ckd_calloc((size_t)n_tmat * n_emit * (n_emit + 1), 1);
        uint16 *sseq = ckd_calloc((size_t)n_sseq * n_emit, sizeof(uint16));
        hmm_context_t *hc = pls->hmmctx;
        int b;
        for (i = 0; i < n_tmat; ++i)
            for (k = 0; k < n_emit; ++k)
                for (b = 0; b <= n_emit; ++b)
                    tp[((size_t)i * n_emit + k) * (n_emit + 1) + b] = hc->tp[i][k][b];
        for (s = 0; s < n_sseq; ++s)
            for (k = 0; k < n_emit; ++k)
                sseq[(size_t)s * n_emit + k] = hc->sseq[s][k];
        i = psgpu_hmm_ctx_create(&d->ctx, n_emit, n_tmat, tp, n_sseq, sseq, n_sen);
        ckd_free(tp); ckd_free(sseq);
        if (i != PSGPU_OK) { E_ERROR("psgpu_hmm_ctx_create: %s\n", psgpu_last_error()); ckd_free(d); return -1; }
    }
    d->par.n_phones = pls->n_phones; d->par.window = pls->window;
    d->par.beam = pls->beam; d->par.pbeam = pls->pbeam; d->par.pip = pls->pip;
    d->par.penalty_weight = pls->penalty_weight;
    /* the CI phones' HMM definitions, and the senone list acmod_flags2list builds when all of
     * them are active (acmod.c:1223-1275): deltas above 255 are bridged with entries that are
     * scored and normalised over like any other */
    ssid = ckd_calloc(pls->n_phones, sizeof *ssid);
    tm = ckd_calloc(pls->n_phones, sizeof *tm);
    flags = ckd_calloc(n_sen, 1);
    for (i = 0; i < pls->n_phones; ++i) {
        hmm_t *h = (hmm_t *)&pls->hmms[i];
        ssid[i] = hmm_nonmpx_ssid(h);
        tm[i] = (int16_t)h->tmatid;
        for (k = 0; k < n_emit; ++k) flags[hmm_nonmpx_senid(h, k)] = 1;
    }
    ci = ckd_calloc(n_sen, sizeof *ci);
    for (last = 0, i = 0; i < n_sen; ++i) {
        if (!flags[i]) continue;
        while (i - last > 255) { last += 255; ci[n_list++] = (uint16_t)last; }
        ci[n_list++] = (uint16_t)i;
        last = i;
    }
    d->n_list = n_list;
    if (dev_alloc(&d->d_ssid, sizeof *ssid * pls->n_phones) || dev_alloc(&d->d_tmatid, sizeof *tm * pls->n_phones)
        || dev_alloc(&d->d_ci, sizeof *ci * (n_list ? n_list : 1)) || dev_alloc(&d->d_off, 2 * sizeof(int32_t))
        || h2d(d->d_ssid, ssid, sizeof *ssid * pls->n_phones) || h2d(d->d_tmatid, tm, sizeof *tm * pls->n_phones)
        || h2d(d->d_ci, ci, sizeof *ci * n_list) || psgpu_stream_sync(NULL)) {
        E_ERROR("psgpu_phone_loop_attach: device allocation failed\n");
        ckd_free(ssid); ckd_free(tm); ckd_free(flags); ckd_free(ci);
        psgpu_hmm_ctx_free(d->ctx); ckd_free(d);
        return -1;
    }
    ckd_free(ssid); ckd_free(tm); ckd_free(flags); ckd_free(ci);
    d->orig = ps->phone_loop->vt;
    d->vt = *d->orig;
    d->vt.step = pl_step;
    ps->phone_loop->vt = &d->vt;
    pthread_mutex_lock(&g_lock);
    d->next = g_list; g_list = d;
    pthread_mutex_unlock(&g_lock);
    return 0;
}

void
psgpu_phone_loop_detach(ps_decoder_t *ps)
{
    pl_dev_t *d, **pp;
    if (ps == NULL || ps->phone_loop == NULL) return;
    pthread_mutex_lock(&g_lock);
    for (pp = &g_list; *pp && (*pp)->search != ps->phone_loop; pp = &(*pp)->next) ;
    d = *pp;
    if (d) *pp = d->next;
    pthread_mutex_unlock(&g_lock);
    if (!d) return;
    ps->phone_loop->vt = d->orig;
    psgpu_free(d->d_ssid); psgpu_free(d->d_tmatid); psgpu_free(d->d_ci); psgpu_free(d->d_off);
    psgpu_free(d->d_pen); psgpu_free(d->d_now); psgpu_free(d->d_state);
    psgpu_host_free(d->pen); psgpu_host_free(d->now); psgpu_host_free(d->state);
    psgpu_hmm_ctx_free(d->ctx);
    ckd_free(d);
}

void
psgpu_phone_loop_stats(ps_decoder_t *ps, long *n_device, long *n_host)
{
    pl_dev_t *d = (ps && ps->phone_loop) ? find(ps->phone_loop) : NULL;
    if (n_device) *n_device = d ? d->n_device : 0;
    if (n_host) *n_host = d ? d->n_host : 0;
}
