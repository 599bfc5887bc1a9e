/* integration/psgpu_search_tables.c -- REFERENCE-SIDE code: see psgpu_search_tables.h.
 *
 * The one place where the reference's search structures become index arrays.  Callers: psgpu_device_decode.c (live
 * decoder -> psgpu_fwdtree_create / psgpu_fwdflat_create), psgpu_export_tables.c (-> table file), oracle/ref_dump.c
 * (test harness: the same arrays in front of its traces). */
#include <stdlib.h>
#include <string.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "util/ckd_alloc.h"
#include "acmod.h"
#include "ngram_search.h"
#include "ngram_search_fwdtree.h"
#include "ngram_search_fwdflat.h"
#include "phone_loop_search.h"
#include "dict2pid.h"
#include "lm/ngram_model.h"
#include "lm/ngram_model_set.h"

#include "psgpu_lm_tables.h"
#include "psgpu_search_tables.h"

/* the tree in the flattened numbering: roots 0..R-1 (ngs->root_chan order), then every other channel depth-first,
 * siblings (->alt) in list order, a channel's subtree (->next) right behind it */
static int
number_nodes(chan_t *first, chan_t **nodes, int n)
{
    chan_t *h;
    for (h = first; h; h = h->alt) { nodes[n++] = h; n = number_nodes(h->next, nodes, n); }
    return n;
}
/* index of a tree channel in that numbering: the pointers are sorted once (the tree of a large dictionary has a quarter of
 * a million channels) */
typedef struct { chan_t *h; int idx; } node_ref_t;
static int
node_ref_cmp(const void *a, const void *b)
{
    const chan_t *x = ((const node_ref_t *)a)->h, *y = ((const node_ref_t *)b)->h;
    return x < y ? -1 : x > y;
}
static int
node_index(node_ref_t *refs, int n, chan_t *h, int base)
{
    node_ref_t key, *r;
    if (h == NULL) return -1;
    key.h = h; key.idx = 0;
    r = bsearch(&key, refs, n, sizeof *refs, node_ref_cmp);
    return r ? base + r->idx : -1;
}

psgpu_search_tables_t *
psgpu_search_tables_collect(ps_decoder_t *ps, int want_flat)
{
    psgpu_search_tables_t *t;
    ngram_search_t *ngs;
    acmod_t *acmod;
    bin_mdef_t *mdef;
    dict_t *dict;
    dict2pid_t *d2p;
    chan_t **nodes;
    node_ref_t *refs;
    int n_ci, n_emit, n_w, R, M, N, n1, i, j, k, w;

    if (ps == NULL || ps->search == NULL || ps->acmod == NULL) return NULL;
    if (strcmp(ps_search_type(ps->search), PS_SEARCH_TYPE_NGRAM)) { E_ERROR("psgpu search tables: not an n-gram search\n"); return NULL; }
    ngs = (ngram_search_t *)ps->search;
    if (!ngs->fwdtree) { E_ERROR("psgpu search tables: needs -fwdtree yes (the lexicon tree is what is flattened)\n"); return NULL; }
    if (want_flat && !ngs->fwdflat) { E_ERROR("psgpu search tables: the second pass's tables need -fwdflat yes\n"); return NULL; }
    acmod = ps->acmod; mdef = acmod->mdef; dict = ps_search_dict(ngs); d2p = ps_search_dict2pid(ngs);
    n_ci = bin_mdef_n_ciphone(mdef); n_emit = bin_mdef_n_emit_state(mdef); n_w = dict_size(dict);
    t = ckd_calloc(1, sizeof *t);
    /* ---- the tree (create_search_channels, ngram_search_fwdtree.c:181-336) */
    R = ngs->n_root_chan;
    nodes = ckd_calloc(ngs->n_nonroot_chan + 16, sizeof *nodes);
    for (M = 0, i = 0; i < R; ++i) M = number_nodes(ngs->root_chan[i].next, nodes, M);
    N = R + M; n1 = ngs->n_1ph_words;
    t->n_ci = n_ci; t->n_emit = n_emit; t->n_w = n_w; t->R = R; t->M = M; t->N = N; t->n1 = n1;
    refs = ckd_calloc(M + 1, sizeof *refs);
    for (i = 0; i < M; ++i) { refs[i].h = nodes[i]; refs[i].idx = i; }
    qsort(refs, M, sizeof *refs, node_ref_cmp);
    t->node_ci = ckd_calloc(N + 1, 4); t->node_ci2 = ckd_calloc(N + 1, 4); t->node_ssid = ckd_calloc(N + 1, 4);
    t->node_tmat = ckd_calloc(N + 1, 4); t->node_child = ckd_calloc(N + 1, 4); t->node_sib = ckd_calloc(N + 1, 4);
    t->node_penult_wid = ckd_calloc(N + 1, 4);
    for (i = 0; i < R; ++i) {
        root_chan_t *r = &ngs->root_chan[i];
        t->node_ci[i] = r->ciphone; t->node_ci2[i] = r->ci2phone; t->node_ssid[i] = hmm_mpx_ssid(&r->hmm, 0);
        t->node_tmat[i] = r->hmm.tmatid; t->node_child[i] = node_index(refs, M, r->next, R); t->node_sib[i] = -1;
        t->node_penult_wid[i] = r->penult_phn_wid;
    }
    for (i = 0; i < M; ++i) {
        chan_t *h = nodes[i];
        t->node_ci[R + i] = h->ciphone; t->node_ci2[R + i] = -1; t->node_ssid[R + i] = hmm_nonmpx_ssid(&h->hmm);
        t->node_tmat[R + i] = h->hmm.tmatid; t->node_child[R + i] = node_index(refs, M, h->next, R);
        t->node_sib[R + i] = node_index(refs, M, h->alt, R); t->node_penult_wid[R + i] = h->info.penult_phn_wid;
    }
    ckd_free(refs); ckd_free(nodes);
    t->homophone_set = ngs->homophone_set;
    /* ---- single-phone words: permanent channels (ngram_fwdtree_init, :380-) */
    t->w1_wid = ckd_calloc(n1 + 1, 4); t->w1_ci = ckd_calloc(n1 + 1, 4); t->w1_ci2 = ckd_calloc(n1 + 1, 4);
    t->w1_ssid = ckd_calloc(n1 + 1, 4); t->w1_tmat = ckd_calloc(n1 + 1, 4); t->w1_mpx = ckd_calloc(n1 + 1, 4);
    for (i = 0; i < n1; ++i) {
        root_chan_t *r = (root_chan_t *)ngs->word_chan[ngs->single_phone_wid[i]];
        t->w1_wid[i] = ngs->single_phone_wid[i]; t->w1_ci[i] = r->ciphone; t->w1_ci2[i] = r->ci2phone;
        t->w1_mpx[i] = hmm_is_mpx(&r->hmm);
        t->w1_ssid[i] = t->w1_mpx[i] ? hmm_mpx_ssid(&r->hmm, 0) : hmm_nonmpx_ssid(&r->hmm);
        t->w1_tmat[i] = r->hmm.tmatid;
    }
    /* ---- dictionary columns */
    t->dict_pronlen = ckd_calloc(n_w + 1, 4); t->dict_first = ckd_calloc(n_w + 1, 4); t->dict_last = ckd_calloc(n_w + 1, 4);
    t->dict_last2 = ckd_calloc(n_w + 1, 4); t->dict_basewid = ckd_calloc(n_w + 1, 4); t->dict_filler = ckd_calloc(n_w + 1, 4);
    t->dict_real = ckd_calloc(n_w + 1, 4);
    for (w = 0; w < n_w; ++w) {
        t->dict_pronlen[w] = dict_pronlen(dict, w); t->dict_first[w] = dict_first_phone(dict, w);
        t->dict_last[w] = dict_last_phone(dict, w);
        t->dict_last2[w] = t->dict_pronlen[w] > 1 ? dict_second_last_phone(dict, w) : -1;
        t->dict_basewid[w] = dict_basewid(dict, w); t->dict_filler[w] = dict_filler_word(dict, w);
        t->dict_real[w] = dict_real_word(dict, w);
    }
    /* ---- dict2pid: right-context tables for every (last phone, second-last phone), root entry ssids */
    t->rssid_n = ckd_calloc((size_t)n_ci * n_ci, 4); t->rssid_ssid = ckd_calloc((size_t)n_ci * n_ci * n_ci, 4);
    t->rssid_cimap = ckd_calloc((size_t)n_ci * n_ci * n_ci, 4); t->ldiph_lc = ckd_calloc((size_t)n_ci * n_ci * n_ci, 4);
    for (i = 0; i < n_ci; ++i)
        for (j = 0; j < n_ci; ++j) {
            xwdssid_t *x = dict2pid_rssid(d2p, i, j);
            t->rssid_n[i * n_ci + j] = x->n_ssid;
            for (k = 0; k < n_ci; ++k) {
                const size_t at = ((size_t)i * n_ci + j) * n_ci + k;
                t->rssid_ssid[at] = (x->ssid && k < x->n_ssid) ? x->ssid[k] : -1;
                t->rssid_cimap[at] = x->cimap ? x->cimap[k] : -1;
                t->ldiph_lc[at] = d2p->ldiph_lc[i][j][k];
            }
        }
    /* ---- HMM topology */
    t->n_tmat = acmod->tmat->n_tmat; t->n_sseq = bin_mdef_n_sseq(mdef);
    t->tp = ckd_calloc((size_t)t->n_tmat * n_emit * (n_emit + 1) + 1, 1);
    t->sseq = ckd_calloc((size_t)t->n_sseq * n_emit + 1, 2);
    t->ci_tmat = ckd_calloc(n_ci + 1, 4);
    for (i = 0; i < t->n_tmat; ++i) for (j = 0; j < n_emit; ++j) for (k = 0; k <= n_emit; ++k)
        t->tp[((size_t)i * n_emit + j) * (n_emit + 1) + k] = acmod->tmat->tp[i][j][k];
    for (i = 0; i < t->n_sseq; ++i) for (j = 0; j < n_emit; ++j) t->sseq[(size_t)i * n_emit + j] = mdef->sseq[i][j];
    for (i = 0; i < n_ci; ++i) t->ci_tmat[i] = bin_mdef_pid2tmatid(mdef, i);
    /* ---- sizes, beams, penalties, special word ids */
    {
        int32 *par = t->par;
        par[0] = n_ci; par[1] = n_emit; par[2] = bin_mdef_n_sen(mdef); par[3] = n_w; par[4] = R; par[5] = M; par[6] = n1;
        par[7] = ngs->n_1ph_LMwords; par[8] = ngs->beam; par[9] = ngs->pbeam; par[10] = ngs->lpbeam; par[11] = ngs->lponlybeam;
        par[12] = ngs->wbeam; par[13] = ngs->pip; par[14] = ngs->nwpen; par[15] = ngs->silpen; par[16] = ngs->fillpen;
        par[17] = ngs->maxhmmpf; par[18] = ngs->maxwpf; par[19] = dict_startwid(dict); par[20] = dict_finishwid(dict);
        par[21] = dict_silwid(dict); par[22] = dict_filler_start(dict); par[23] = dict_filler_end(dict); par[24] = mdef->sil;
        par[25] = ps_search_lookahead(ngs) != NULL; par[26] = acmod->compallsen;
    }
    /* ---- what the second pass adds: pronunciations as word-internal ssids, CI ssids, LM membership, its beams */
    if (want_flat) {
        int64_t tot = 0, o = 0;
        for (w = 0; w < n_w; ++w) tot += dict_pronlen(dict, w);
        t->has_flat = 1; t->pron_total = tot;
        t->pron_off = ckd_calloc(n_w + 1, 4); t->pron_ci = ckd_calloc(tot + 1, 4); t->pron_ssid = ckd_calloc(tot + 1, 4);
        t->ci_ssid = ckd_calloc(n_ci + 1, 4); t->lm_known = ckd_calloc(n_w + 1, 4);
        for (w = 0; w < n_w; ++w) {
            int len = dict_pronlen(dict, w);
            t->pron_off[w] = (int32)o;
            for (k = 0; k < len; ++k, ++o) {
                t->pron_ci[o] = dict_pron(dict, w, k);
                t->pron_ssid[o] = (k >= 1 && k < len - 1) ? dict2pid_internal(d2p, w, k) : -1;
            }
            t->lm_known[w] = ngram_model_set_known_wid(ngs->lmset, dict_basewid(dict, w)) ? 1 : 0;
        }
        t->pron_off[n_w] = (int32)o;
        for (i = 0; i < n_ci; ++i) t->ci_ssid[i] = bin_mdef_pid2ssid(mdef, i);
        t->flat_par[0] = ngs->fwdflatbeam; t->flat_par[1] = ngs->fwdflatwbeam; t->flat_par[2] = ngs->min_ef_width;
        t->flat_par[3] = ngs->max_sf_win;
        t->flat_lwf = ngs->fwdflat_fwdtree_lw_ratio;
    }
    /* ---- the phone loop feeding the look-ahead penalties (phone_loop_search.h:75-94) */
    if (ps->phone_loop) {
        phone_loop_search_t *pls = (phone_loop_search_t *)ps->phone_loop;
        t->has_pl = 1;
        t->pl_par[0] = pls->n_phones; t->pl_par[1] = pls->window; t->pl_par[2] = pls->beam; t->pl_par[3] = pls->pbeam;
        t->pl_par[4] = pls->pip; t->pl_par[5] = ps->pl_window;
        t->pl_weight = pls->penalty_weight;
        t->pl_ssid = ckd_calloc(pls->n_phones + 1, 4); t->pl_tmat = ckd_calloc(pls->n_phones + 1, 4);
        for (i = 0; i < pls->n_phones; ++i) { t->pl_ssid[i] = hmm_nonmpx_ssid(&pls->hmms[i]); t->pl_tmat[i] = pls->hmms[i].tmatid; }
    }
    return t;
}

void
psgpu_search_tables_free(psgpu_search_tables_t *t)
{
    if (!t) return;
    ckd_free(t->node_ci); ckd_free(t->node_ci2); ckd_free(t->node_ssid); ckd_free(t->node_tmat); ckd_free(t->node_child);
    ckd_free(t->node_sib); ckd_free(t->node_penult_wid);
    ckd_free(t->w1_wid); ckd_free(t->w1_ci); ckd_free(t->w1_ci2); ckd_free(t->w1_ssid); ckd_free(t->w1_tmat); ckd_free(t->w1_mpx);
    ckd_free(t->dict_pronlen); ckd_free(t->dict_first); ckd_free(t->dict_last); ckd_free(t->dict_last2); ckd_free(t->dict_basewid);
    ckd_free(t->dict_filler); ckd_free(t->dict_real);
    ckd_free(t->rssid_n); ckd_free(t->rssid_ssid); ckd_free(t->rssid_cimap); ckd_free(t->ldiph_lc);
    ckd_free(t->tp); ckd_free(t->sseq); ckd_free(t->ci_tmat);
    ckd_free(t->pron_off); ckd_free(t->pron_ci); ckd_free(t->pron_ssid); ckd_free(t->ci_ssid); ckd_free(t->lm_known);
    ckd_free(t->pl_ssid); ckd_free(t->pl_tmat);
    ckd_free(t);
}

static void
e1(psgpu_table_emit_fn emit, void *ctx, const char *name, char dt, int64_t a, const void *d)
{
    emit(ctx, name, dt, 1, &a, d);
}
static void
e2(psgpu_table_emit_fn emit, void *ctx, const char *name, char dt, int64_t a, int64_t b, const void *d)
{
    int64_t dims[2]; dims[0] = a; dims[1] = b;
    emit(ctx, name, dt, 2, dims, d);
}
static void
e3(psgpu_table_emit_fn emit, void *ctx, const char *name, char dt, int64_t a, int64_t b, int64_t c, const void *d)
{
    int64_t dims[3]; dims[0] = a; dims[1] = b; dims[2] = c;
    emit(ctx, name, dt, 3, dims, d);
}

void
psgpu_search_tables_emit(const psgpu_search_tables_t *t, psgpu_table_emit_fn emit, void *ctx)
{
    const int N = t->N, n_w = t->n_w, n1 = t->n1, n_ci = t->n_ci, ne = t->n_emit;
    e1(emit, ctx, "node_ci", 'i', N, t->node_ci); e1(emit, ctx, "node_ci2", 'i', N, t->node_ci2);
    e1(emit, ctx, "node_ssid", 'i', N, t->node_ssid); e1(emit, ctx, "node_tmat", 'i', N, t->node_tmat);
    e1(emit, ctx, "node_child", 'i', N, t->node_child); e1(emit, ctx, "node_sib", 'i', N, t->node_sib);
    e1(emit, ctx, "node_penult_wid", 'i', N, t->node_penult_wid);
    e1(emit, ctx, "homophone_set", 'i', n_w, t->homophone_set);
    e1(emit, ctx, "w1_wid", 'i', n1, t->w1_wid); e1(emit, ctx, "w1_ci", 'i', n1, t->w1_ci); e1(emit, ctx, "w1_ci2", 'i', n1, t->w1_ci2);
    e1(emit, ctx, "w1_ssid", 'i', n1, t->w1_ssid); e1(emit, ctx, "w1_tmat", 'i', n1, t->w1_tmat); e1(emit, ctx, "w1_mpx", 'i', n1, t->w1_mpx);
    e1(emit, ctx, "dict_pronlen", 'i', n_w, t->dict_pronlen); e1(emit, ctx, "dict_first", 'i', n_w, t->dict_first);
    e1(emit, ctx, "dict_last", 'i', n_w, t->dict_last); e1(emit, ctx, "dict_last2", 'i', n_w, t->dict_last2);
    e1(emit, ctx, "dict_basewid", 'i', n_w, t->dict_basewid); e1(emit, ctx, "dict_filler", 'i', n_w, t->dict_filler);
    e1(emit, ctx, "dict_real", 'i', n_w, t->dict_real);
    e2(emit, ctx, "rssid_n", 'i', n_ci, n_ci, t->rssid_n); e3(emit, ctx, "rssid_ssid", 'i', n_ci, n_ci, n_ci, t->rssid_ssid);
    e3(emit, ctx, "rssid_cimap", 'i', n_ci, n_ci, n_ci, t->rssid_cimap); e3(emit, ctx, "ldiph_lc", 'i', n_ci, n_ci, n_ci, t->ldiph_lc);
    e3(emit, ctx, "tp", 'B', t->n_tmat, ne, ne + 1, t->tp); e2(emit, ctx, "sseq", 'H', t->n_sseq, ne, t->sseq);
    e1(emit, ctx, "ci_tmat", 'i', n_ci, t->ci_tmat);
    e1(emit, ctx, "par", 'i', 32, t->par);
    if (t->has_flat) {
        e1(emit, ctx, "pron_off", 'i', n_w + 1, t->pron_off); e1(emit, ctx, "pron_ci", 'i', t->pron_total, t->pron_ci);
        e1(emit, ctx, "pron_ssid", 'i', t->pron_total, t->pron_ssid);
        e1(emit, ctx, "ci_ssid", 'i', n_ci, t->ci_ssid); e1(emit, ctx, "lm_known", 'i', n_w, t->lm_known);
        e1(emit, ctx, "flat_par", 'i', 16, t->flat_par); e1(emit, ctx, "flat_lwf", 'f', 1, &t->flat_lwf);
    }
    if (t->has_pl) {
        e1(emit, ctx, "pl_par", 'i', 8, t->pl_par); e1(emit, ctx, "pl_weight", 'd', 1, &t->pl_weight);
        e1(emit, ctx, "pl_ssid", 'i', t->pl_par[0], t->pl_ssid); e1(emit, ctx, "pl_tmat", 'i', t->pl_par[0], t->pl_tmat);
    }
}

void
psgpu_search_tables_view(const psgpu_search_tables_t *t, psgpu_fwdtree_tables_t *ft, psgpu_fwdflat_tables_t *ff)
{
    memset(ft, 0, sizeof *ft);
    ft->par = t->par; ft->node_ci = t->node_ci; ft->node_ci2 = t->node_ci2; ft->node_ssid = t->node_ssid; ft->node_tmat = t->node_tmat;
    ft->node_child = t->node_child; ft->node_sib = t->node_sib; ft->node_penult_wid = t->node_penult_wid;
    ft->homophone_set = t->homophone_set;
    ft->w1_wid = t->w1_wid; ft->w1_ci = t->w1_ci; ft->w1_ci2 = t->w1_ci2; ft->w1_ssid = t->w1_ssid; ft->w1_tmat = t->w1_tmat;
    ft->w1_mpx = t->w1_mpx;
    ft->dict_pronlen = t->dict_pronlen; ft->dict_first = t->dict_first; ft->dict_last = t->dict_last; ft->dict_last2 = t->dict_last2;
    ft->dict_basewid = t->dict_basewid; ft->dict_filler = t->dict_filler;
    ft->rssid_n = t->rssid_n; ft->rssid_ssid = t->rssid_ssid; ft->rssid_cimap = t->rssid_cimap; ft->ldiph_lc = t->ldiph_lc;
    ft->tp = t->tp; ft->sseq = t->sseq; ft->ci_tmat = t->ci_tmat; ft->lm = NULL; ft->n_tmat = t->n_tmat; ft->n_sseq = t->n_sseq;
    if (ff) {
        memset(ff, 0, sizeof *ff);
        ff->ft = ft;
        if (t->has_flat) {
            ff->pron_off = t->pron_off; ff->pron_ci = t->pron_ci; ff->pron_ssid = t->pron_ssid; ff->ci_ssid = t->ci_ssid;
            ff->lm_known = t->lm_known;
            ff->fwdflatbeam = t->flat_par[0]; ff->fwdflatwbeam = t->flat_par[1]; ff->min_ef_width = t->flat_par[2];
            ff->max_sf_win = t->flat_par[3]; ff->lwf = t->flat_lwf;
        }
    }
}

int32_t *
psgpu_search_tables_dense_lm(ps_decoder_t *ps, int fixed_point)
{
    ngram_search_t *ngs = (ngram_search_t *)ps->search;
    dict_t *dict = ps_search_dict(ngs);
    const int n_w = dict_size(dict);
    const size_t nn = (size_t)n_w + 1;
    int32 *lm = ckd_calloc((size_t)n_w * nn * nn, 4);
    int i, j, k;
    if (fixed_point) {
        /* The trie's back-off cache (lm_trie.c:775-811) starts zeroed: a full-history look-up whose model history is (0, 0)
         * matches the zeroed key and is answered with zero back-off weights until any OTHER history has filled the cache.
         * The table below is the cache's fixed point (what every look-up returns once it has been filled), so it is filled
         * first, with a history of two different words (their model ids cannot both be 0).  A search never meets the initial
         * state: its first full-history look-up has the history (w, <s>) with w != <s>. */
        int a = -1, b = -1;
        for (i = 0; i < n_w && b < 0; ++i)
            if (!dict_filler_word(dict, i) && dict_basewid(dict, i) == i && ngram_model_set_known_wid(ngs->lmset, i)) {
                if (a < 0) a = i; else b = i;
            }
        if (b >= 0) { int32 nu; (void)ngram_tg_score(ngs->lmset, a, b, a, &nu); (void)ngram_tg_score(ngs->lmset, a, a, b, &nu); }
    }
    for (i = 0; i < n_w; ++i)
        if (!dict_filler_word(dict, i) && dict_basewid(dict, i) == i)
            for (j = -1; j < n_w; ++j)
                for (k = -1; k < n_w; ++k) {
                    int32 nu;
                    lm[((size_t)i * nn + (j + 1)) * nn + (k + 1)] = ngram_tg_score(ngs->lmset, i, j, k, &nu) >> SENSCR_SHIFT;
                }
    return lm;
}

int
psgpu_lm_tables_emit(ngram_model_t *lmset, psgpu_table_emit_fn emit, void *ctx)
{
    psgpu_lm_tables_t t;
    uint32_t lev[PSGPU_LM_MAX_LEVELS * 7];
    int32_t v;
    int l, w;
    size_t nb = 0;
    char *words;
    if (psgpu_lm_tables_read(lmset, &t) < 0) return -1;
    v = t.order; e1(emit, ctx, "order", 'i', 1, &v);
    v = t.n_unigrams; e1(emit, ctx, "n_unigrams", 'i', 1, &v);
    v = t.n_words; e1(emit, ctx, "n_words", 'i', 1, &v);
    e2(emit, ctx, "unigrams", 'i', t.n_unigrams + 1, 3, t.unigrams);
    e1(emit, ctx, "ngram_mem", 'B', (int64_t)t.ngram_mem_size, t.ngram_mem ? (const void *)t.ngram_mem : (const void *)"");
    for (l = 0; l < t.order - 1; ++l) {
        lev[7 * l] = t.level_offset[l]; lev[7 * l + 1] = t.total_bits[l]; lev[7 * l + 2] = t.word_bits[l];
        lev[7 * l + 3] = t.word_mask[l]; lev[7 * l + 4] = t.max_vocab[l]; lev[7 * l + 5] = t.next_bits[l];
        lev[7 * l + 6] = t.next_mask[l];
    }
    e2(emit, ctx, "levels", 'i', t.order - 1, 7, lev);
    if (t.order > 1) e2(emit, ctx, "quant", 'f', 2 * (t.order - 2) + 1, 65536, t.quant);
    e1(emit, ctx, "lw", 'f', 1, &t.lw);
    v = t.log_wip; e1(emit, ctx, "log_wip", 'i', 1, &v);
    v = t.log_zero; e1(emit, ctx, "log_zero", 'i', 1, &v);
    e1(emit, ctx, "widmap", 'i', t.n_words, t.widmap);
    for (w = 0; w < t.n_words; ++w) nb += strlen(ngram_word(lmset, w)) + 1;
    words = ckd_calloc(nb + 1, 1);
    for (w = 0, nb = 0; w < t.n_words; ++w) {
        const char *s = ngram_word(lmset, w);
        memcpy(words + nb, s, strlen(s)); nb += strlen(s); words[nb++] = '\n';
    }
    e1(emit, ctx, "words", 'B', (int64_t)nb, words);
    ckd_free(words);
    psgpu_lm_tables_release(&t);
    return 0;
}
