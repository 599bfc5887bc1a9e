/* integration/psgpu_device_decode.c -- REFERENCE-SIDE code.
 *
 * The first pass of an utterance entirely on the MI355X, behind the reference's own
 * result API: PCM -> MFCC (psgpu_fe) -> 1s_c_d_dd features -> PTM senone scores
 * (un-normalised rows) -> phone-loop search -> lexicon-tree search, then the
 * back-pointer table, right-context score stack and frame marks are copied into the
 * decoder's ngram_search_t in the reference's own layout (bptbl_t, ngram_search.h:112-124;
 * SURVEY 8f-2), so that ps_get_hyp(), ps_seg_iter() and friends run unchanged on them.
 *
 * What is read out of the decoder, once (psgpu_device_decode_attach): the search tree
 * create_search_channels built (flattened: roots, then depth-first), single-phone word
 * channels, dictionary and dict2pid tables, beams and penalties, the phone loop's HMMs
 * and beams, and the language model as a dense table over dictionary word ids
 * (ngram_tg_score for every triple) -- which limits this binding to small vocabularies
 * until the trie lookup itself is on the device.  Requires the n-gram search with
 * -fwdflat no (pass 1 on the device; -bestpath yes builds the word lattice from the
 * injected table on the host, ngram_search.c:1212, and searches it as usual), the PTM scorer (psgpu_mgau_attach first) and
 * the 1s_c_d_dd feature type. */
#include <stdlib.h>
#include <string.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "util/ckd_alloc.h"
#include "acmod.h"
#include "ngram_search.h"
#include "phone_loop_search.h"
#include "dict2pid.h"
#include "lm/ngram_model.h"
#include "fe/fe_internal.h"

#include "psgpu.h"
#include "psgpu_mgau_shim.h"
#include "psgpu_fe_shim.h"
#include "psgpu_lm_tables.h"
#include "psgpu_device_decode.h"

struct psgpu_device_decode_s {
    ps_decoder_t *ps;
    psgpu_fwdtree_t *ft;
    psgpu_lm_t *lm;                    /* the trie on the device (NULL: dense table inside ft) */
    psgpu_hmm_ctx_t *ctx;
    psgpu_fe_t *fe;
    psgpu_ptm_model_t *model;          /* borrowed from the attached scorer */
    psgpu_phone_loop_params_t plpar;
    int pl_window, n_ci, n_sen, n_chain, topn, cepsize, n_list;
    uint16_t *d_ssid, *d_ci; int16_t *d_tmatid;
    /* per-utterance device buffers, grown on demand */
    int cap_frames; size_t cap_samples;
    int16_t *d_pcm; float *d_cep, *d_feat; int32_t *d_off, *d_tsc; uint8_t *d_tcw; int16_t *d_rows; int32_t *d_best;
    int32_t *d_pen, *d_now, *d_state, *d_bp, *d_bss, *d_idx, *d_step, *d_res;
    int32_t *h_bp, *h_bss, *h_idx;
    int bp_cap, bss_cap;
    /* the second pass on the device as well (PSGPU_DEVICE_SECOND_PASS=1 with -fwdflat yes) */
    psgpu_fwdflat_t *ff;
    psgpu_ptm_view_t view;
    int n_fast_hist, n1, n_emit;
    int32_t *d_w1, *d_seed, *d_bp2, *d_bss2, *d_idx2, *d_step2, *d_res2, *h_seed;
    uint8_t *h_tcw;
};

static int
number_nodes(chan_t *first, chan_t **nodes, int n)
{
    chan_t *h;
    for (h = first; h; h = h->alt) { nodes[n++] = h; n = number_nodes(h->next, nodes, n); }
    return n;
}
/* index of a tree channel in the flattened numbering: the pointers are sorted once (the tree of a large
 * dictionary has a quarter of a million channels) */
typedef struct { chan_t *h; int idx; } node_ref_t;
static int
node_ref_cmp(const void *a, const void *b)
{
    const chan_t *x = ((const node_ref_t *)a)->h, *y = ((const node_ref_t *)b)->h;
    return x < y ? -1 : x > y;
}
static node_ref_t *g_refs;       /* set for the duration of one attach (single-threaded, like ps_init) */
static int
node_index(chan_t **nodes, int n, chan_t *h, int base)
{
    node_ref_t key, *r;
    (void)nodes;
    if (h == NULL) return -1;
    key.h = h; key.idx = 0;
    r = bsearch(&key, g_refs, n, sizeof *g_refs, node_ref_cmp);
    return r ? base + r->idx : -1;
}

psgpu_device_decode_t *
psgpu_device_decode_attach(ps_decoder_t *ps)
{
    psgpu_device_decode_t *d;
    ngram_search_t *ngs;
    acmod_t *acmod;
    bin_mdef_t *mdef;
    dict_t *dict;
    dict2pid_t *d2p;
    phone_loop_search_t *pls;
    psgpu_fwdtree_tables_t t;
    chan_t **nodes;
    int n_ci, n_emit, n_w, R, M, N, n1, i, j, k, w, n_tmat, n_sseq, lm_ok;
    int32 par[32];
    int32 *ci, *ci2, *ssid, *tm, *child, *sib, *pw, *sw, *sci, *sci2, *sss, *stm, *smpx;
    int32 *pl, *p0, *pz, *py, *bw, *fl, *rn, *rs, *rm, *ld, *ptm, *lm;
    uint8 *tp; uint16 *sq;
    psgpu_fe_shim_t *fes;
    psgpu_lm_tables_t lt;

    if (ps == NULL || ps->search == NULL || ps->acmod == NULL) return NULL;
    if (strcmp(ps_search_type(ps->search), PS_SEARCH_TYPE_NGRAM)) { E_ERROR("psgpu device decode: not an n-gram search\n"); return NULL; }
    ngs = (ngram_search_t *)ps->search;
    if (!ngs->fwdtree || (ngs->fwdflat && !(getenv("PSGPU_DEVICE_SECOND_PASS") && atoi(getenv("PSGPU_DEVICE_SECOND_PASS"))))) {
        E_ERROR("psgpu device decode: needs -fwdtree yes -fwdflat no (pass 1 on the device, the lattice pass on the host); "
                "-fwdflat yes with PSGPU_DEVICE_SECOND_PASS=1 in the environment runs the second pass on the device too\n");
        return NULL;
    }
    acmod = ps->acmod; mdef = acmod->mdef; dict = ps_search_dict(ngs); d2p = ps_search_dict2pid(ngs);
    n_ci = bin_mdef_n_ciphone(mdef); n_emit = bin_mdef_n_emit_state(mdef); n_w = dict_size(dict);
    if (strcmp(feat_name(acmod->fcb), "1s_c_d_dd") || acmod->fcb->lda || acmod->compallsen || acmod->fcb->cmn != CMN_BATCH
        || acmod->fcb->agc != AGC_NONE || acmod->fcb->varnorm) {
        E_ERROR("psgpu device decode: needs the 1s_c_d_dd feature type with -cmn batch, no AGC / variance normalisation / LDA, "
                "and -compallsen no\n");
        return NULL;
    }
    d = ckd_calloc(1, sizeof *d);
    d->ps = ps;
    d->model = psgpu_mgau_ptm_model(acmod->mgau);
    if (d->model == NULL) { E_ERROR("psgpu device decode: attach the psgpu PTM scorer first\n"); ckd_free(d); return NULL; }
    d->n_ci = n_ci; d->n_sen = bin_mdef_n_sen(mdef);
    d->n_chain = psgpu_ptm_n_chain(d->model); d->topn = psgpu_ptm_topn(d->model);
    d->cepsize = feat_cepsize(acmod->fcb);
    /* ---- the search tables (cf. oracle/ref_dump.c cmd_fwdtree, which writes the same arrays to a file) */
    R = ngs->n_root_chan;
    nodes = ckd_calloc(ngs->n_nonroot_chan + 16, sizeof *nodes);
    for (M = 0, i = 0; i < R; ++i) M = number_nodes(ngs->root_chan[i].next, nodes, M);
    N = R + M; n1 = ngs->n_1ph_words;
    g_refs = ckd_calloc(M + 1, sizeof *g_refs);
    for (i = 0; i < M; ++i) { g_refs[i].h = nodes[i]; g_refs[i].idx = i; }
    qsort(g_refs, M, sizeof *g_refs, node_ref_cmp);
    ci = ckd_calloc(N, 4); ci2 = ckd_calloc(N, 4); ssid = ckd_calloc(N, 4); tm = ckd_calloc(N, 4); child = ckd_calloc(N, 4);
    sib = ckd_calloc(N, 4); pw = ckd_calloc(N, 4);
    for (i = 0; i < R; ++i) {
        root_chan_t *r = &ngs->root_chan[i];
        ci[i] = r->ciphone; ci2[i] = r->ci2phone; ssid[i] = hmm_mpx_ssid(&r->hmm, 0); tm[i] = r->hmm.tmatid;
        child[i] = node_index(nodes, M, r->next, R); sib[i] = -1; pw[i] = r->penult_phn_wid;
    }
    for (i = 0; i < M; ++i) {
        chan_t *h = nodes[i];
        ci[R + i] = h->ciphone; ci2[R + i] = -1; ssid[R + i] = hmm_nonmpx_ssid(&h->hmm); tm[R + i] = h->hmm.tmatid;
        child[R + i] = node_index(nodes, M, h->next, R); sib[R + i] = node_index(nodes, M, h->alt, R);
        pw[R + i] = h->info.penult_phn_wid;
    }
    ckd_free(g_refs); g_refs = NULL;
    sw = ckd_calloc(n1 + 1, 4); sci = ckd_calloc(n1 + 1, 4); sci2 = ckd_calloc(n1 + 1, 4); sss = ckd_calloc(n1 + 1, 4);
    stm = ckd_calloc(n1 + 1, 4); smpx = ckd_calloc(n1 + 1, 4);
    for (i = 0; i < n1; ++i) {
        root_chan_t *r = (root_chan_t *)ngs->word_chan[ngs->single_phone_wid[i]];
        sw[i] = ngs->single_phone_wid[i]; sci[i] = r->ciphone; sci2[i] = r->ci2phone; smpx[i] = hmm_is_mpx(&r->hmm);
        sss[i] = smpx[i] ? hmm_mpx_ssid(&r->hmm, 0) : hmm_nonmpx_ssid(&r->hmm);
        stm[i] = r->hmm.tmatid;
    }
    pl = ckd_calloc(n_w, 4); p0 = ckd_calloc(n_w, 4); pz = ckd_calloc(n_w, 4); py = ckd_calloc(n_w, 4); bw = ckd_calloc(n_w, 4);
    fl = ckd_calloc(n_w, 4);
    for (w = 0; w < n_w; ++w) {
        pl[w] = dict_pronlen(dict, w); p0[w] = dict_first_phone(dict, w); pz[w] = dict_last_phone(dict, w);
        py[w] = pl[w] > 1 ? dict_second_last_phone(dict, w) : -1; bw[w] = dict_basewid(dict, w); fl[w] = dict_filler_word(dict, w);
    }
    rn = ckd_calloc((size_t)n_ci * n_ci, 4); rs = ckd_calloc((size_t)n_ci * n_ci * n_ci, 4);
    rm = ckd_calloc((size_t)n_ci * n_ci * n_ci, 4); ld = ckd_calloc((size_t)n_ci * n_ci * n_ci, 4);
    for (i = 0; i < n_ci; ++i)
        for (j = 0; j < n_ci; ++j) {
            xwdssid_t *x = dict2pid_rssid(d2p, i, j);
            rn[i * n_ci + j] = x->n_ssid;
            for (k = 0; k < n_ci; ++k) {
                rs[((size_t)i * n_ci + j) * n_ci + k] = (x->ssid && k < x->n_ssid) ? x->ssid[k] : -1;
                rm[((size_t)i * n_ci + j) * n_ci + k] = x->cimap ? x->cimap[k] : -1;
                ld[((size_t)i * n_ci + j) * n_ci + k] = d2p->ldiph_lc[i][j][k];
            }
        }
    n_tmat = acmod->tmat->n_tmat; n_sseq = bin_mdef_n_sseq(mdef);
    tp = ckd_calloc((size_t)n_tmat * n_emit * (n_emit + 1), 1);
    sq = ckd_calloc((size_t)n_sseq * n_emit, 2);
    ptm = ckd_calloc(n_ci, 4);
    for (i = 0; i < n_tmat; ++i) for (j = 0; j < n_emit; ++j) for (k = 0; k <= n_emit; ++k)
        tp[((size_t)i * n_emit + j) * (n_emit + 1) + k] = acmod->tmat->tp[i][j][k];
    for (i = 0; i < n_sseq; ++i) for (j = 0; j < n_emit; ++j) sq[(size_t)i * n_emit + j] = mdef->sseq[i][j];
    for (i = 0; i < n_ci; ++i) ptm[i] = bin_mdef_pid2tmatid(mdef, i);
    memset(par, 0, sizeof par);
    par[0] = n_ci; par[1] = n_emit; par[2] = d->n_sen; par[3] = n_w; par[4] = R; par[5] = M; par[6] = n1;
    par[7] = ngs->n_1ph_LMwords; par[8] = ngs->beam; par[9] = ngs->pbeam; par[10] = ngs->lpbeam; par[11] = ngs->lponlybeam;
    par[12] = ngs->wbeam; par[13] = ngs->pip; par[14] = ngs->nwpen; par[15] = ngs->silpen; par[16] = ngs->fillpen;
    par[17] = ngs->maxhmmpf; par[18] = ngs->maxwpf; par[19] = dict_startwid(dict); par[20] = dict_finishwid(dict);
    par[21] = dict_silwid(dict); par[22] = dict_filler_start(dict); par[23] = dict_filler_end(dict); par[24] = mdef->sil;
    par[25] = ps_search_lookahead(ngs) != NULL; par[26] = acmod->compallsen;
    /* language scores: the model's own trie on the device when it is one trie model without classes
     * (psgpu_lm_tables.c), else -- small vocabularies only -- every ngram_tg_score in a dense table */
    lm = NULL;
    if (psgpu_lm_tables_read(ngs->lmset, &lt) == 0) {
        if (psgpu_lm_create(&d->lm, &lt) != PSGPU_OK) {
            E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
            d->lm = NULL;
        }
        psgpu_lm_tables_release(&lt);
    }
    lm_ok = d->lm != NULL || n_w <= 400;
    if (!lm_ok)
        E_ERROR("psgpu device decode: %d words and no trie model -- the dense LM table is for small vocabularies\n", n_w);
    else if (d->lm == NULL) {
        size_t nn = (size_t)n_w + 1;
        lm = ckd_calloc((size_t)n_w * nn * nn, 4);
        for (i = 0; i < n_w; ++i)
            if (!dict_filler_word(dict, i) && dict_basewid(dict, i) == i)
                for (j = -1; j < n_w; ++j)
                    for (k = -1; k < n_w; ++k) {
                        int32 nu;
                        lm[((size_t)i * nn + (j + 1)) * nn + (k + 1)] = ngram_tg_score(ngs->lmset, i, j, k, &nu) >> SENSCR_SHIFT;
                    }
    }
    memset(&t, 0, sizeof t);
    t.par = par; t.node_ci = ci; t.node_ci2 = ci2; t.node_ssid = ssid; t.node_tmat = tm; t.node_child = child; t.node_sib = sib;
    t.node_penult_wid = pw; t.homophone_set = ngs->homophone_set; t.w1_wid = sw; t.w1_ci = sci; t.w1_ci2 = sci2; t.w1_ssid = sss;
    t.w1_tmat = stm; t.w1_mpx = smpx; t.dict_pronlen = pl; t.dict_first = p0; t.dict_last = pz; t.dict_last2 = py;
    t.dict_basewid = bw; t.dict_filler = fl; t.rssid_n = rn; t.rssid_ssid = rs; t.rssid_cimap = rm; t.ldiph_lc = ld;
    t.tp = tp; t.sseq = sq; t.ci_tmat = ptm; t.lm = lm; t.n_tmat = n_tmat; t.n_sseq = n_sseq;
    i = lm_ok ? psgpu_fwdtree_create(&d->ft, &t) : PSGPU_EINVAL;
    if (i == PSGPU_OK && d->lm) i = psgpu_fwdtree_set_lm(d->ft, d->lm);
    if (i == PSGPU_OK && ngs->fwdflat) {
        /* ---- what the second pass adds (cf. oracle/ref_dump.c cmd_fwdtree(.., flat = 1)): pronunciations as word-internal
         *      ssids, the CI phones' ssids, which words the language model knows, its beams and windows */
        psgpu_fwdflat_tables_t t2;
        int64_t tot = 0, o = 0;
        int32 *off = ckd_calloc(n_w + 1, 4), *pci, *pss, *cis = ckd_calloc(n_ci, 4), *known = ckd_calloc(n_w, 4);
        for (w = 0; w < n_w; ++w) tot += dict_pronlen(dict, w);
        pci = ckd_calloc(tot + 1, 4); pss = ckd_calloc(tot + 1, 4);
        for (w = 0; w < n_w; ++w) {
            int len = dict_pronlen(dict, w);
            off[w] = (int32)o;
            for (k = 0; k < len; ++k, ++o) {
                pci[o] = dict_pron(dict, w, k);
                pss[o] = (k >= 1 && k < len - 1) ? dict2pid_internal(d2p, w, k) : -1;
            }
            known[w] = ngram_model_set_known_wid(ngs->lmset, dict_basewid(dict, w)) ? 1 : 0;
        }
        off[n_w] = (int32)o;
        for (j = 0; j < n_ci; ++j) cis[j] = bin_mdef_pid2ssid(mdef, j);
        memset(&t2, 0, sizeof t2);
        t2.ft = &t; t2.pron_off = off; t2.pron_ci = pci; t2.pron_ssid = pss; t2.ci_ssid = cis; t2.lm_known = known;
        t2.fwdflatbeam = ngs->fwdflatbeam; t2.fwdflatwbeam = ngs->fwdflatwbeam; t2.min_ef_width = ngs->min_ef_width;
        t2.max_sf_win = ngs->max_sf_win; t2.lwf = ngs->fwdflat_fwdtree_lw_ratio;
        i = psgpu_fwdflat_create(&d->ff, &t2);
        if (i == PSGPU_OK && d->lm) i = psgpu_fwdflat_set_lm(d->ff, d->lm);
        if (i == PSGPU_OK) i = psgpu_ptm_model_view(d->model, &d->view);
        d->n_fast_hist = ps->pl_window + 2;               /* ptm_mgau.c:884 */
        d->n1 = ngs->n_1ph_words; d->n_emit = n_emit;
        if (i == PSGPU_OK && (psgpu_malloc((void **)&d->d_w1, 4 * (size_t)d->n1 * n_emit + 4)
                              || psgpu_malloc((void **)&d->d_seed, 4 * (size_t)d->n_chain * d->topn + 4)))
            i = PSGPU_ENOMEM;
        d->h_seed = ckd_calloc((size_t)d->n_chain * d->topn + 1, 4);
        ckd_free(off); ckd_free(pci); ckd_free(pss); ckd_free(cis); ckd_free(known);
    }
    /* PSGPU_FWDTREE_MODE=active_list: the large-vocabulary formulation of the kernel (psgpu.h, psgpu_fwdtree_set_mode);
     * same tables, per-frame work proportional to the active channels */
    if (i == PSGPU_OK && getenv("PSGPU_FWDTREE_MODE") && !strcmp(getenv("PSGPU_FWDTREE_MODE"), "active_list"))
        i = psgpu_fwdtree_set_mode(d->ft, PSGPU_FWDTREE_ACTIVE_LIST);
    if (i == PSGPU_OK) i = psgpu_hmm_ctx_create(&d->ctx, n_emit, n_tmat, tp, n_sseq, sq, d->n_sen);
    /* ---- the phone loop (cf. psgpu_phone_loop_shim.c) */
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
        d->n_list = nl;
        d->plpar.n_phones = pls->n_phones; d->plpar.window = pls->window; d->plpar.beam = pls->beam; d->plpar.pbeam = pls->pbeam;
        d->plpar.pip = pls->pip; d->plpar.penalty_weight = pls->penalty_weight;
        d->pl_window = ps->pl_window;
        if (psgpu_malloc((void **)&d->d_ssid, 2 * pls->n_phones) || psgpu_malloc((void **)&d->d_tmatid, 2 * pls->n_phones)
            || psgpu_malloc((void **)&d->d_ci, 2 * (nl ? nl : 1)) || psgpu_memcpy_h2d(d->d_ssid, ps_ssid, 2 * pls->n_phones, NULL)
            || psgpu_memcpy_h2d(d->d_tmatid, ps_tm, 2 * pls->n_phones, NULL) || psgpu_memcpy_h2d(d->d_ci, cil, 2 * nl, NULL)
            || psgpu_stream_sync(NULL))
            i = PSGPU_EHIP;
        ckd_free(ps_ssid); ckd_free(cil); ckd_free(ps_tm); ckd_free(flags);
    }
    else if (i == PSGPU_OK) {
        E_ERROR("psgpu device decode: needs the phone-loop look-ahead (pl_window > 0, <= 64 CI phones)\n");
        i = PSGPU_EINVAL;
    }
    ckd_free(nodes); ckd_free(ci); ckd_free(ci2); ckd_free(ssid); ckd_free(tm); ckd_free(child); ckd_free(sib); ckd_free(pw);
    ckd_free(sw); ckd_free(sci); ckd_free(sci2); ckd_free(sss); ckd_free(stm); ckd_free(smpx);
    ckd_free(pl); ckd_free(p0); ckd_free(pz); ckd_free(py); ckd_free(bw); ckd_free(fl); ckd_free(rn); ckd_free(rs); ckd_free(rm);
    ckd_free(ld); ckd_free(tp); ckd_free(sq); ckd_free(ptm); ckd_free(lm);
    if (i == PSGPU_OK) {
        fes = psgpu_fe_wrap(acmod->fe);
        if (fes) d->fe = psgpu_fe_shim_release(fes); else i = PSGPU_EINVAL;
    }
    if (i != PSGPU_OK) {
        E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
        psgpu_device_decode_detach(d);
        return NULL;
    }
    d->bp_cap = 16384; d->bss_cap = 1 << 19;
    return d;
}

void
psgpu_device_decode_detach(psgpu_device_decode_t *d)
{
    if (!d) return;
    psgpu_fwdflat_free(d->ff); psgpu_free(d->d_w1); psgpu_free(d->d_seed); psgpu_free(d->d_bp2); psgpu_free(d->d_bss2);
    psgpu_free(d->d_idx2); psgpu_free(d->d_step2); psgpu_free(d->d_res2); ckd_free(d->h_seed); ckd_free(d->h_tcw);
    psgpu_fwdtree_free(d->ft); psgpu_lm_free(d->lm); psgpu_hmm_ctx_free(d->ctx); psgpu_fe_free(d->fe);
    psgpu_free(d->d_ssid); psgpu_free(d->d_tmatid); psgpu_free(d->d_ci);
    psgpu_free(d->d_pcm); psgpu_free(d->d_cep); psgpu_free(d->d_feat); psgpu_free(d->d_off); psgpu_free(d->d_tsc); psgpu_free(d->d_tcw);
    psgpu_free(d->d_rows); psgpu_free(d->d_best); psgpu_free(d->d_pen); psgpu_free(d->d_now); psgpu_free(d->d_state);
    psgpu_free(d->d_bp); psgpu_free(d->d_bss); psgpu_free(d->d_idx); psgpu_free(d->d_step); psgpu_free(d->d_res);
    ckd_free(d->h_bp); ckd_free(d->h_bss); ckd_free(d->h_idx);
    ckd_free(d);
}

static int
grow(psgpu_device_decode_t *d, size_t n_samples, int T)
{
    if (n_samples > d->cap_samples) {
        psgpu_free(d->d_pcm); d->d_pcm = NULL;
        if (psgpu_malloc((void **)&d->d_pcm, 2 * n_samples)) return -1;
        d->cap_samples = n_samples;
    }
    if (T > d->cap_frames) {
        size_t t = (size_t)T + T / 2 + 64, ne = t * d->n_chain * d->topn;
        psgpu_free(d->d_cep); psgpu_free(d->d_feat); psgpu_free(d->d_tsc); psgpu_free(d->d_tcw); psgpu_free(d->d_rows);
        psgpu_free(d->d_best); psgpu_free(d->d_pen); psgpu_free(d->d_now); psgpu_free(d->d_state); psgpu_free(d->d_idx);
        psgpu_free(d->d_step); psgpu_free(d->d_off); psgpu_free(d->d_bp); psgpu_free(d->d_bss); psgpu_free(d->d_res);
        ckd_free(d->h_bp); ckd_free(d->h_bss); ckd_free(d->h_idx);
        if (d->ff) {
            psgpu_free(d->d_bp2); psgpu_free(d->d_bss2); psgpu_free(d->d_idx2); psgpu_free(d->d_step2); psgpu_free(d->d_res2);
            ckd_free(d->h_tcw);
            d->d_bp2 = d->d_bss2 = d->d_idx2 = d->d_step2 = d->d_res2 = NULL; d->h_tcw = NULL;
            if (psgpu_malloc((void **)&d->d_bp2, 4 * (size_t)10 * d->bp_cap) || psgpu_malloc((void **)&d->d_bss2, 4 * (size_t)d->bss_cap)
                || psgpu_malloc((void **)&d->d_idx2, 4 * (t + 2)) || psgpu_malloc((void **)&d->d_step2, 4 * t * 4)
                || psgpu_malloc((void **)&d->d_res2, 32))
                return -1;
            d->h_tcw = ckd_calloc(ne + 1, 1);
        }
        d->cap_frames = 0;
        if (psgpu_malloc((void **)&d->d_cep, 4 * t * d->cepsize) || psgpu_malloc((void **)&d->d_feat, 4 * t * 3 * d->cepsize)
            || psgpu_malloc((void **)&d->d_tsc, 4 * ne) || psgpu_malloc((void **)&d->d_tcw, ne)
            || psgpu_malloc((void **)&d->d_rows, 2 * t * d->n_sen) || psgpu_malloc((void **)&d->d_best, 4 * t)
            || psgpu_malloc((void **)&d->d_pen, 4 * t * d->n_ci) || psgpu_malloc((void **)&d->d_now, 4 * t * d->n_ci)
            || psgpu_malloc((void **)&d->d_state, 4 * t * d->n_ci * 8) || psgpu_malloc((void **)&d->d_idx, 4 * (t + 2))
            || psgpu_malloc((void **)&d->d_step, 4 * t * 4) || psgpu_malloc((void **)&d->d_off, 8)
            || psgpu_malloc((void **)&d->d_bp, 4 * (size_t)10 * d->bp_cap) || psgpu_malloc((void **)&d->d_bss, 4 * (size_t)d->bss_cap)
            || psgpu_malloc((void **)&d->d_res, 32))
            return -1;
        d->h_bp = ckd_calloc((size_t)10 * d->bp_cap, 4); d->h_bss = ckd_calloc(d->bss_cap, 4); d->h_idx = ckd_calloc(t + 2, 4);
        d->cap_frames = (int)t;
    }
    return 0;
}

int
psgpu_device_decode_utt(psgpu_device_decode_t *d, int16 const *pcm, size_t n_samples)
{
    ps_decoder_t *ps = d->ps;
    ngram_search_t *ngs = (ngram_search_t *)ps->search;
    int64_t soff[2] = { 0, (int64_t)n_samples };
    int32_t fo[2], res[8];
    int T = (int)psgpu_fe_n_frames(d->fe, (int64_t)n_samples), i, nb, nh, nfr;
    void *st = psgpu_hmm_ctx_stream(d->ctx);       /* one stream for the whole chain */

    if (grow(d, n_samples ? n_samples : 1, T ? T : 1) < 0) { E_ERROR("psgpu device decode: %s\n", psgpu_last_error()); return -1; }
    /* the reference's own start / end-of-utterance housekeeping, without any frame going through its search */
    if (ps_start_utt(ps) < 0) return -1;
    if (ps_end_utt(ps) < 0) return -1;
    if (T == 0) return 0;
    if (d->ff && psgpu_fwdtree_set_w1_ssid_out(d->ft, d->d_w1)) { E_ERROR("psgpu device decode: %s\n", psgpu_last_error()); return -1; }
    if (psgpu_memcpy_h2d(d->d_pcm, pcm, 2 * n_samples, st)
        || psgpu_fe_process_utts_dev(d->fe, d->d_pcm, soff, 1, NULL, NULL, d->d_cep, d->d_off, fo, st)
        || psgpu_feat_1s_c_d_dd_dev(d->d_cep, d->d_off, 1, d->cepsize, d->d_feat, st)
        || psgpu_ptm_score_batch_dev(d->model, d->d_feat, d->d_off, 1, T, NULL, NULL, d->d_tsc, d->d_tcw, d->d_rows, d->d_best,
                                     PSGPU_PTM_RAW_SCORES, st)
        || psgpu_phone_loop_run_dev(d->ctx, &d->plpar, d->d_ssid, d->d_tmatid, d->d_ci, d->n_list, d->d_rows, d->n_sen, NULL,
                                    d->d_off, 1, T, d->d_pen, d->d_now, d->d_state, st)
        || psgpu_fwdtree_search_dev(d->ft, d->d_rows, d->n_sen, d->d_pen, d->d_off, 1, T, d->bp_cap, d->bss_cap, d->d_bp, d->d_bss,
                                    d->d_idx, d->d_step, d->d_res, 1, d->pl_window, st)
        || psgpu_memcpy_d2h(res, d->d_res, sizeof res, st) || psgpu_stream_sync(st)) {
        E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
        return -1;
    }
    if (res[3]) { E_ERROR("psgpu device decode: back-pointer table or score stack full\n"); return -1; }
    if (d->ff) {
        /* ---- the second pass (ngram_search_finish, ngram_search.c:791-808, does it inside ps_end_utt on the host): the
         *      flat-lexicon search over the first pass's device-resident table, scoring its own senones from the feature
         *      rows; its scorer state starts from the lists pass 1 left in history slot n_fast_hist - 1 (ptm_mgau.c:425-441),
         *      i.e. the batch scorer's lists (chain-major [n_chain][T][topn]) of the last frame ts with ts % H == H - 1 */
        int H = d->n_fast_hist, ts = T - 1, c;
        while (ts >= 0 && ts % H != H - 1) --ts;
        if (psgpu_memcpy_d2h(d->h_tcw, d->d_tcw, (size_t)d->n_chain * T * d->topn, st) || psgpu_stream_sync(st)) {
            E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
            return -1;
        }
        for (c = 0; c < d->n_chain; ++c)
            for (i = 0; i < d->topn; ++i)      /* a shorter utterance than H frames never wrote that slot: ptm_mgau_init's i-th codeword */
                d->h_seed[c * d->topn + i] = ts >= 0 ? d->h_tcw[((size_t)c * T + ts) * d->topn + i] : i;
        if (psgpu_memcpy_h2d(d->d_seed, d->h_seed, 4 * (size_t)d->n_chain * d->topn, st)
            || psgpu_fwdflat_search_feats_dev(d->ff, &d->view, d->d_feat, d->d_seed, d->d_off, 1, T, d->bp_cap, d->d_bp, d->d_res,
                                              d->d_w1, d->bp_cap, d->bss_cap, d->d_bp2, d->d_bss2, d->d_idx2, d->d_step2, d->d_res2, st)
            || psgpu_memcpy_d2h(res, d->d_res2, sizeof res, st) || psgpu_stream_sync(st)) {
            E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
            return -1;
        }
        if (res[3]) { E_ERROR("psgpu device decode: second pass: back-pointer table or score stack full\n"); return -1; }
    }
    nb = res[0]; nh = res[1]; nfr = res[2];
    if (psgpu_memcpy_d2h(d->h_bp, d->ff ? d->d_bp2 : d->d_bp, 4 * (size_t)10 * d->bp_cap, st)
        || psgpu_memcpy_d2h(d->h_bss, d->ff ? d->d_bss2 : d->d_bss, 4 * (size_t)(nh ? nh : 1), st)
        || psgpu_memcpy_d2h(d->h_idx, d->ff ? d->d_idx2 : d->d_idx, 4 * ((size_t)nfr + 1), st) || psgpu_stream_sync(st)) {
        E_ERROR("psgpu device decode: %s\n", psgpu_last_error());
        return -1;
    }
    /* ---- the tables in the reference's layout (ngram_search.h:112-124, ngram_search.c:301-339, 445-497) */
    if (nb > ngs->bp_table_size) {
        ngs->bp_table_size = nb + nb / 2;
        ngs->bp_table = ckd_realloc(ngs->bp_table, ngs->bp_table_size * sizeof(*ngs->bp_table));
    }
    if (nh + d->n_ci >= ngs->bscore_stack_size) {
        ngs->bscore_stack_size = nh + d->n_ci + nh / 2 + 1;
        ngs->bscore_stack = ckd_realloc(ngs->bscore_stack, ngs->bscore_stack_size * sizeof(*ngs->bscore_stack));
    }
    if (nfr + 1 >= ngs->n_frame_alloc) {
        ngs->n_frame_alloc = nfr + 2;
        ngs->bp_table_idx = (int32 *)ckd_realloc(ngs->bp_table_idx - 1, (ngs->n_frame_alloc + 1) * sizeof(*ngs->bp_table_idx)) + 1;
    }
    for (i = 0; i < nb; ++i) {
        bptbl_t *e = &ngs->bp_table[i];
#define COL(c) d->h_bp[(size_t)(c) * d->bp_cap + i]
        e->frame = COL(0); e->valid = (uint8)COL(1); e->refcnt = 0; e->wid = COL(2); e->bp = COL(3); e->score = COL(4);
        e->s_idx = COL(5); e->real_wid = COL(6); e->prev_real_wid = COL(7); e->last_phone = (int16)COL(8); e->last2_phone = (int16)COL(9);
#undef COL
    }
    memcpy(ngs->bscore_stack, d->h_bss, sizeof(int32) * nh);
    memcpy(ngs->bp_table_idx, d->h_idx, sizeof(int32) * (nfr + 1));
    ngs->bpidx = nb; ngs->bss_head = nh; ngs->n_frame = nfr;
    ngs->best_score = res[4];        /* ngram_search_lattice (ngram_search.c:1226) refuses an utterance whose best score is WORST_SCORE */
    return nfr;
}
