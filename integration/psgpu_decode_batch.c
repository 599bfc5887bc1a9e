/* integration/psgpu_decode_batch.c -- REFERENCE-SIDE code.
 *
 * The additive batch call of SURVEY 8(b).  The reference decodes on one thread
 * and has no batched entry; its decoder objects are independent, so a batch is
 * B utterances handed to n_workers decoders, each on its own host thread with
 * its own psgpu scorer (model + state + HIP stream) on the same MI355X.  With
 * PSGPU_BATCH_DEVICE_FE the cepstra of the WHOLE batch come out of one device
 * call before the workers start.  Utterances are independent: no
 * shared mutable state, no collective; the work queue is one atomic counter. */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "util/ckd_alloc.h"
#include "ptm_mgau.h"
#include "ngram_search.h"
#include "hmm.h"
#include "fe/fe_noise.h"
#include "fe/fe_internal.h"

#include "psgpu.h"
#include "psgpu_mgau_shim.h"
#include "psgpu_fe_shim.h"
#include "psgpu_phone_loop_shim.h"
#include "psgpu_device_decode.h"
#include "psgpu_decode_batch.h"
#ifdef PSGPU_SEARCH_HOOKS
#include "psgpu_search_hooks.h"
#endif

struct psgpu_batch_s {
    int n_workers;
    unsigned flags;
    int device;                    /* the device the decoders' psgpu objects live on */
    ps_decoder_t **ps;
    psgpu_fe_t *fe;                /* batch front end (tables of worker 0's fe_t) */
    psgpu_device_decode_t *dd;     /* PSGPU_BATCH_DEVICE_FIRST_PASS: the device pipeline, bound to worker 0's decoder */
    void *rd_stream;               /* ... ONE stream for the workers' table fetches (made at the first call: making a stream costs ~18 ms, and
                                    *     every further stream alive in the process slows the pipeline's own launches and waits) */
    psgpu_dd_stage_t *rd_stage;    /* ... and their staging buffers */
    int out_dim;
    /* one call's work */
    const int16 *const *pcm;
    const size_t *n;
    int B;
    psgpu_batch_result_t *out;
    float *cep;                    /* [total frames][out_dim] when the device front end is on */
    int32_t *frame_off;
    int next;                      /* work queue */
    int failed;
};

typedef struct worker_arg_s {
    psgpu_batch_t *b;
    int w;
} worker_arg_t;

static char *
dup_str(const char *s)
{
    size_t n = strlen(s) + 1;
    char *d = malloc(n);
    memcpy(d, s, n);
    return d;
}

void
psgpu_batch_result_clear(psgpu_batch_result_t *r)
{
    int i;
    if (!r) return;
    for (i = 0; i < r->n_seg; ++i) free(r->seg[i].word);
    free(r->seg); free(r->hyp);
    memset(r, 0, sizeof *r);
}

/* hmm_clear (hmm.c:181-196) resets scores, histories and the frame stamp of an HMM but
 * not the per-state senone-sequence ids a multiplex HMM picked up along the winning arcs
 * (hmm.c:609-707); hmm_init (hmm.c:146-168) starts them at BAD_SSID.  The lexicon tree's
 * permanently allocated multiplex HMMs (root channels, single-phone words) therefore keep
 * the ids of the previous utterance, which puts extra senones on the active list of the
 * first frames (acmod_activate_hmm, acmod.c:1179-1221) and so moves the per-frame
 * normalisation -- a decoder's scores depend on what it decoded before.  A batch call
 * must not: put them back to what hmm_init leaves. */
static void
fresh_mpx_ssids(hmm_t *h)
{
    int i;
    if (hmm_is_mpx(h))
        for (i = 1; i < hmm_n_emit_state(h); ++i)
            h->senid[i] = BAD_SSID;
}

static void
reset_search(ps_decoder_t *ps)
{
    ps_search_t *search = ps->search;
    if (search && !strcmp(ps_search_type(search), PS_SEARCH_TYPE_NGRAM)) {
        ngram_search_t *ngs = (ngram_search_t *)search;
        int i;
        if (ngs->fwdtree && ngs->root_chan)
            for (i = 0; i < ngs->n_root_chan; ++i)
                fresh_mpx_ssids(&ngs->root_chan[i].hmm);
        if (ngs->word_chan && ngs->single_phone_wid)       /* fwdtree and fwdflat share these (ngram_search_fwdflat.c:160-190) */
            for (i = 0; i < ngs->n_1ph_words; ++i)
                if (ngs->word_chan[ngs->single_phone_wid[i]])
                    fresh_mpx_ssids(&((root_chan_t *)ngs->word_chan[ngs->single_phone_wid[i]])->hmm);
    }
}

/* the state a decoder has after ps_start_stream() on its first utterance */
static int
reset_decoder(psgpu_batch_t *b, ps_decoder_t *ps)
{
    ps_start_stream(ps);                                   /* fe_reset_noisestats (pocketsphinx.c:1081) */
    reset_search(ps);
    if (b->flags & PSGPU_BATCH_CPU_ONLY) {
        if (!strcmp(ps->acmod->mgau->vt->name, "ptm"))
            ptm_mgau_reset_fast_hist(ps->acmod->mgau);
        return 0;
    }
    return psgpu_mgau_reset(ps->acmod->mgau);
}

static int
collect(ps_decoder_t *ps, psgpu_batch_result_t *r)
{
    const char *hyp;
    ps_seg_t *seg;
    int cap = 0;
    hyp = ps_get_hyp(ps, &r->score);
    r->hyp = dup_str(hyp ? hyp : "");
    for (seg = ps_seg_iter(ps); seg; seg = ps_seg_next(seg)) {
        psgpu_batch_seg_t *s;
        if (r->n_seg == cap) {
            cap = cap ? 2 * cap : 16;
            r->seg = realloc(r->seg, cap * sizeof *r->seg);
        }
        s = &r->seg[r->n_seg++];
        s->word = dup_str(ps_seg_word(seg));
        ps_seg_frames(seg, &s->sf, &s->ef);
        ps_seg_prob(seg, &s->ascr, &s->lscr, &s->lback);
    }
    return 0;
}

static int
decode_one(psgpu_batch_t *b, ps_decoder_t *ps, int u)
{
    psgpu_batch_result_t *r = &b->out[u];
    int rv;

    memset(r, 0, sizeof *r);
    if (reset_decoder(b, ps) < 0) return -1;
    if (ps_start_utt(ps) < 0) return -1;
    if (b->cep) {
        /* rows of the batch's cepstra; ps_process_cep normalises them in place (CMN), they are ours */
        int nfr = b->frame_off[u + 1] - b->frame_off[u], t;
        mfcc_t **rows = ckd_calloc(nfr ? nfr : 1, sizeof *rows);
        for (t = 0; t < nfr; ++t)
            rows[t] = b->cep + (size_t)(b->frame_off[u] + t) * b->out_dim;
        rv = ps_process_cep(ps, rows, nfr, FALSE, TRUE);
        ckd_free(rows);
    }
    else
        rv = ps_process_raw(ps, b->pcm[u], b->n[u], FALSE, TRUE);
    if (rv < 0) { ps_end_utt(ps); return -1; }
    if (ps_end_utt(ps) < 0) return -1;
    r->n_frames = ps_get_n_frames(ps);
    return collect(ps, r);
}

static void *
worker(void *arg)
{
    worker_arg_t *a = arg;
    psgpu_batch_t *b = a->b;
    /* HIP's current device is per thread: a worker must be on the device its decoder's
     * model, state and stream were created on (one process per GPU, or psgpu_set_device
     * before psgpu_batch_init in a multi-GPU process) */
    if (b->device >= 0) psgpu_set_device(b->device);
    for (;;) {
        int u = __atomic_fetch_add(&b->next, 1, __ATOMIC_RELAXED);
        if (u >= b->B) break;
        if (decode_one(b, b->ps[a->w], u) < 0)
            __atomic_store_n(&b->failed, 1, __ATOMIC_RELAXED);
    }
    return NULL;
}

/* PSGPU_BATCH_DEVICE_FIRST_PASS: the batch's searches ran on the device in one launch set; what is left per utterance is the
 * reference's own read-out -- the tables injected into a decoder, ps_get_hyp / ps_seg_iter on them, with -bestpath yes the lattice
 * and its best path (ngram_search.c:782, ps_lattice.c:1216: ~15 ms of host work for 30 s of audio) -- independent of one another:
 * every worker reads utterances out into ITS decoder (staging buffers and a stream of its own) */
static void *
readout_worker(void *arg)
{
    worker_arg_t *a = arg;
    psgpu_batch_t *b = a->b;
    psgpu_dd_stage_t *st = &b->rd_stage[a->w];
    void *stream = b->rd_stream;
    if (b->device >= 0) psgpu_set_device(b->device);
    for (;;) {
        int u = __atomic_fetch_add(&b->next, 1, __ATOMIC_RELAXED), nfr;
        psgpu_batch_result_t *r;
        if (u >= b->B) break;
        r = &b->out[u];
        memset(r, 0, sizeof *r);
        if ((nfr = psgpu_device_decode_batch_select_into(b->dd, u, b->ps[a->w], st, stream)) < 0) {
            r->hyp = dup_str("");
            __atomic_store_n(&b->failed, 1, __ATOMIC_RELAXED);
            continue;
        }
        r->n_frames = psgpu_device_decode_batch_n_frames(b->dd, u) + 1;           /* ps_get_n_frames: output_frame + 1 */
        collect(b->ps[a->w], r);
    }
    return NULL;
}

psgpu_batch_t *
psgpu_batch_init(ps_config_t *config, int n_workers, unsigned flags)
{
    psgpu_batch_t *b;
    int w;

    if (config == NULL || n_workers < 1) return NULL;
#ifndef PSGPU_SEARCH_HOOKS
    if (flags & PSGPU_BATCH_DEVICE_SEARCH) {
        E_ERROR("PSGPU_BATCH_DEVICE_SEARCH needs the library built with the search hooks\n");
        return NULL;
    }
#endif
    b = calloc(1, sizeof *b);
    b->n_workers = n_workers; b->flags = flags;
    b->device = (flags & PSGPU_BATCH_CPU_ONLY) ? -1 : psgpu_get_device();
    b->ps = calloc(n_workers, sizeof *b->ps);
    b->rd_stage = calloc(n_workers, sizeof *b->rd_stage);
    for (w = 0; w < n_workers; ++w) {
        b->ps[w] = ps_init(config);                        /* ps_init retains the config */
        if (b->ps[w] == NULL) goto fail;
        if (flags & PSGPU_BATCH_CPU_ONLY) continue;
        if (psgpu_mgau_attach(b->ps[w]) < 0) {
            E_ERROR("psgpu_mgau_attach failed: %s\n", psgpu_last_error());
            goto fail;
        }
#ifdef PSGPU_SEARCH_HOOKS
        if ((flags & PSGPU_BATCH_DEVICE_SEARCH) && psgpu_search_attach(b->ps[w]) < 0) goto fail;
#endif
        if ((flags & PSGPU_BATCH_DEVICE_PHONE_LOOP) && psgpu_phone_loop_attach(b->ps[w]) < 0) {
            E_ERROR("psgpu_phone_loop_attach failed\n");
            goto fail;
        }
    }
    if ((flags & PSGPU_BATCH_DEVICE_FIRST_PASS) && !(flags & PSGPU_BATCH_CPU_ONLY)) {
        /* the pipeline reads the search tables out of worker 0's decoder; a call's results are injected into the workers' decoders,
         * every worker reading utterances out side by side (readout_worker) */
        if (ps_config_bool(ps_get_config(b->ps[0]), "fwdflat")
            && !(getenv("PSGPU_DEVICE_SECOND_PASS") && atoi(getenv("PSGPU_DEVICE_SECOND_PASS")))) {
            /* (refused here, at initialisation, not at the first decode: the reference's own second pass wants the utterance's
             *  feature vectors in acmod, which this path -- front end on the device -- never fills; with the device second pass
             *  both passes of the batch run on the device and the second pass's tables are injected) */
            E_ERROR("PSGPU_BATCH_DEVICE_FIRST_PASS with -fwdflat yes needs PSGPU_DEVICE_SECOND_PASS=1 (both passes on the device), or "
                    "-fwdflat no, or the device search bound behind ps_decode_raw (psgpu_device_search_attach), which runs the "
                    "reference's second pass after the device's first\n");
            goto fail;
        }
        b->dd = psgpu_device_decode_attach(b->ps[0]);
        if (b->dd == NULL) goto fail;
    }
    if ((flags & PSGPU_BATCH_DEVICE_FE) && !(flags & PSGPU_BATCH_CPU_ONLY) && !(flags & PSGPU_BATCH_DEVICE_FIRST_PASS)) {
        /* one front end for the batch: psgpu_fe_wrap's table read-out, kept as the raw object */
        psgpu_fe_shim_t *s = psgpu_fe_wrap(b->ps[0]->acmod->fe);
        if (s == NULL) goto fail;
        b->fe = psgpu_fe_shim_release(s);
        b->out_dim = psgpu_fe_out_dim(b->fe);
    }
    return b;
fail:
    psgpu_batch_free(b);
    return NULL;
}

void
psgpu_batch_free(psgpu_batch_t *b)
{
    int w;
    if (!b) return;
    psgpu_device_decode_detach(b->dd);
    for (w = 0; w < b->n_workers; ++w) {
        if (!b->ps[w]) continue;
#ifdef PSGPU_SEARCH_HOOKS
        if (b->flags & PSGPU_BATCH_DEVICE_SEARCH) psgpu_search_detach(b->ps[w]);
#endif
        psgpu_phone_loop_detach(b->ps[w]);
        ps_free(b->ps[w]);
    }
    for (w = 0; w < b->n_workers; ++w) if (b->rd_stage) psgpu_dd_stage_release(&b->rd_stage[w]);
    if (b->rd_stream) psgpu_stream_destroy(b->rd_stream);
    free(b->rd_stage);
    psgpu_fe_free(b->fe);
    free(b->ps);
    free(b);
}

int
psgpu_decode_batch(psgpu_batch_t *b, const int16 *const pcm[], const size_t n[], int B,
                   psgpu_batch_result_t out[])
{
    pthread_t *tid;
    worker_arg_t *args;
    int w, nw, rc = 0, *started;

    if (b == NULL || B < 0 || (B > 0 && (!pcm || !n || !out))) return -1;
    if (B == 0) return 0;
    b->pcm = pcm; b->n = n; b->B = B; b->out = out; b->next = 0; b->failed = 0;
    b->cep = NULL; b->frame_off = NULL;
    if (b->dd) {
        /* the whole first pass of the batch in one launch set; then the reference's own read-out per utterance */
        int u;
        struct timespec ts0, ts1, ts2;
        clock_gettime(CLOCK_MONOTONIC, &ts0);
        if (psgpu_device_decode_batch_run(b->dd, pcm, n, B) < 0) return -1;
        clock_gettime(CLOCK_MONOTONIC, &ts1);
        nw = b->n_workers < B ? b->n_workers : B;
        if (nw > 1) {                                      /* the read-outs side by side, one decoder a worker */
            if (b->rd_stream == NULL && psgpu_stream_create_dedicated(&b->rd_stream) != PSGPU_OK) b->rd_stream = NULL;   /* (none: the default stream) */
            tid = calloc(nw, sizeof *tid);
            started = calloc(nw, sizeof *started);
            args = calloc(nw, sizeof *args);
            for (w = 0; w < nw; ++w) {
                args[w].b = b; args[w].w = w;
                if (w == nw - 1) readout_worker(&args[w]);  /* the caller's thread is the last worker */
                else if (pthread_create(&tid[w], NULL, readout_worker, &args[w]) == 0) started[w] = 1;
                else readout_worker(&args[w]);
            }
            for (w = 0; w + 1 < nw; ++w) if (started[w]) pthread_join(tid[w], NULL);
            free(started); free(tid); free(args);
            if (getenv("PSGPU_BATCH_TIMING")) {            /* (where a call's time goes: stderr) */
                clock_gettime(CLOCK_MONOTONIC, &ts2);
                fprintf(stderr, "psgpu_decode_batch: %d utterances: device passes %.1f ms, read-outs on %d threads %.1f ms\n", B,
                        1e3 * (ts1.tv_sec - ts0.tv_sec) + 1e-6 * (ts1.tv_nsec - ts0.tv_nsec), nw,
                        1e3 * (ts2.tv_sec - ts1.tv_sec) + 1e-6 * (ts2.tv_nsec - ts1.tv_nsec));
            }
            return b->failed ? -1 : 0;
        }
        for (u = 0; u < B; ++u) {
            psgpu_batch_result_t *r = &out[u];
            int nfr;
            memset(r, 0, sizeof *r);
            if ((nfr = psgpu_device_decode_batch_select(b->dd, u)) < 0) { r->hyp = dup_str(""); rc = -1; continue; }
            r->n_frames = psgpu_device_decode_batch_n_frames(b->dd, u) + 1;    /* ps_get_n_frames: output_frame + 1 */
            collect(b->ps[0], r);
        }
        return rc;
    }
    if (b->fe) {
        /* the whole batch through the device front end in one call, every utterance from
         * reset noise statistics (noise arrays NULL) */
        int64_t *soff = malloc(sizeof *soff * ((size_t)B + 1));
        int64_t total = 0, frames = 0;
        int16 *all;
        int u;
        soff[0] = 0;
        for (u = 0; u < B; ++u) {
            soff[u + 1] = soff[u] + (int64_t)n[u];
            frames += psgpu_fe_n_frames(b->fe, (int64_t)n[u]);
        }
        total = soff[B];
        all = malloc(sizeof *all * (size_t)(total ? total : 1));
        for (u = 0; u < B; ++u) memcpy(all + soff[u], pcm[u], sizeof *all * n[u]);
        b->cep = malloc(sizeof(float) * (size_t)(frames ? frames : 1) * b->out_dim);
        b->frame_off = malloc(sizeof(int32_t) * ((size_t)B + 1));
        if (psgpu_fe_process_utts(b->fe, all, soff, B, NULL, NULL, b->cep, b->frame_off) != PSGPU_OK) {
            E_ERROR("psgpu_fe_process_utts: %s\n", psgpu_last_error());
            rc = -1;
        }
        free(all); free(soff);
    }
    if (rc == 0) {
        nw = b->n_workers < B ? b->n_workers : B;
        tid = calloc(nw, sizeof *tid);
        started = calloc(nw, sizeof *started);
        args = calloc(nw, sizeof *args);
        for (w = 0; w < nw; ++w) {
            args[w].b = b; args[w].w = w;
            started[w] = 0;
            if (w == nw - 1) worker(&args[w]);             /* the caller's thread is the last worker */
            else if (pthread_create(&tid[w], NULL, worker, &args[w]) == 0) started[w] = 1;
            else worker(&args[w]);                         /* no thread to be had: its share runs here */
        }
        for (w = 0; w + 1 < nw; ++w) if (started[w]) pthread_join(tid[w], NULL);
        free(started);
        free(tid); free(args);
        if (b->failed) rc = -1;
    }
    free(b->cep); free(b->frame_off);
    b->cep = NULL; b->frame_off = NULL;
    return rc;
}

/* ---- several devices ------------------------------------------------------------------------------------------------ */
struct psgpu_multi_s {
    int n;
    int *device;
    psgpu_batch_t **b;
};

psgpu_multi_t *
psgpu_multi_init(ps_config_t *config, const int devices[], int n_devices, int n_workers, unsigned flags)
{
    psgpu_multi_t *m;
    int k;
    if (config == NULL || devices == NULL || n_devices < 1) return NULL;
    m = ckd_calloc(1, sizeof *m);
    m->n = n_devices;
    m->device = ckd_calloc(n_devices, sizeof *m->device);
    m->b = ckd_calloc(n_devices, sizeof *m->b);
    for (k = 0; k < n_devices; ++k) {
        m->device[k] = devices[k];
        if (!(flags & PSGPU_BATCH_CPU_ONLY) && psgpu_set_device(devices[k]) != PSGPU_OK) {
            E_ERROR("psgpu_multi_init: device %d: %s\n", devices[k], psgpu_last_error());
            psgpu_multi_free(m);
            return NULL;
        }
        m->b[k] = psgpu_batch_init(config, n_workers, flags);
        if (m->b[k] == NULL) { psgpu_multi_free(m); return NULL; }
    }
    return m;
}

void
psgpu_multi_free(psgpu_multi_t *m)
{
    int k;
    if (m == NULL) return;
    for (k = 0; k < m->n; ++k)
        if (m->b[k]) { psgpu_set_device(m->device[k]); psgpu_batch_free(m->b[k]); }
    ckd_free(m->b); ckd_free(m->device); ckd_free(m);
}

int psgpu_multi_n_devices(const psgpu_multi_t *m) { return m ? m->n : 0; }

typedef struct multi_arg_s {
    psgpu_multi_t *m;
    int k, b0, b1, rc;
    const int16 *const *pcm;
    const size_t *n;
    psgpu_batch_result_t *out;
} multi_arg_t;

static void *
multi_worker(void *p)
{
    multi_arg_t *a = p;
    /* the device is a property of the calling thread */
    if (psgpu_set_device(a->m->device[a->k]) != PSGPU_OK) { a->rc = -1; return NULL; }
    a->rc = a->b1 > a->b0 ? psgpu_decode_batch(a->m->b[a->k], a->pcm + a->b0, a->n + a->b0, a->b1 - a->b0, a->out + a->b0) : 0;
    return NULL;
}

int
psgpu_decode_batch_multi(psgpu_multi_t *m, const int16 *const pcm[], const size_t n[], int B, psgpu_batch_result_t out[])
{
    multi_arg_t *a;
    pthread_t *tid;
    int *started, k, u = 0, rc = 0;
    double total = 0, acc = 0;
    if (m == NULL || B < 0 || (B > 0 && (pcm == NULL || n == NULL || out == NULL))) return -1;
    a = ckd_calloc(m->n, sizeof *a); tid = ckd_calloc(m->n, sizeof *tid); started = ckd_calloc(m->n, sizeof *started);
    for (k = 0; k < B; ++k) total += (double)n[k];
    /* consecutive blocks of about equal audio length (utterance boundaries; the same split as pocketsphinx_amd/batch.py) */
    for (k = 0; k < m->n; ++k) {
        a[k].m = m; a[k].k = k; a[k].pcm = pcm; a[k].n = n; a[k].out = out; a[k].b0 = u;
        if (k == m->n - 1) u = B;
        else while (u < B && acc + (double)n[u] / 2 <= total * (k + 1) / m->n) acc += (double)n[u++];
        a[k].b1 = u;
    }
    for (k = 0; k < m->n; ++k) {
        if (pthread_create(&tid[k], NULL, multi_worker, &a[k]) != 0) { multi_worker(&a[k]); continue; }   /* (no thread: in line) */
        started[k] = 1;
    }
    for (k = 0; k < m->n; ++k) {
        if (started[k]) pthread_join(tid[k], NULL);
        if (a[k].rc < 0) rc = -1;
    }
    ckd_free(a); ckd_free(tid); ckd_free(started);
    return rc;
}
