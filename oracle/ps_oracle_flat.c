/* oracle/ps_oracle_flat.c -- TEST INFRASTRUCTURE (CPU oracle, part 4): the flat-lexicon second pass.
 *
 * A plain-C restatement of the reference's ngram_search_fwdflat.c (ngram_fwdflat_start,
 * ngram_fwdflat_search per frame, ngram_fwdflat_finish) with the back-pointer helpers of ngram_search.c it
 * shares with the first pass, on flat index-based tables: the utterance's word list and per-word channel chains
 * are laid out contiguously ([root][word-internal phones][right-context fan-out]) -- the layout the device
 * kernel csrc/psgpu_flat.hip uses.  Everything static comes out of the unmodified reference through
 * `ref_dump fwdflat`; per frame the oracle is handed the senone scores the reference's pass 2 was handed.
 * Only tests/ may call it.
 *
 * Pinned by tests/test_oracle_flat.py: per-frame active senone lists, best scores and back-pointer counts, and
 * the final back-pointer table / score stack / frame marks identical to the reference's.  Every function cites
 * the reference lines it restates (paths relative to /root/reference/src).
 */
#include <stdlib.h>
#include <string.h>
#include "ps_oracle.h"
#include "ps_oracle_search.h"
#include "ps_oracle_flat.h"
#include "ps_oracle_lm.h"

#define WORST ((int32_t)0xE0000000)     /* WORST_SCORE, hmm.h:84 */
#define NO_BP (-1)
#define BAD_SSID 0xffff

struct pso_ff_s {
    pso_ff_tables_t t;
    const pso_lm_t *trie;
    int n_ci, n_emit, n_sen, n_w, n1;
    int32_t beam, pip, silpen, fillpen, fwdflatbeam, fwdflatwbeam, min_ef_width, max_sf_win;
    int32_t startwid, finishwid, silwid, filler_start, filler_end, sil_ci;
    float lwf;
    pso_hmm_ctx_t ctx;
    int16_t *senscr;
    uint8_t *sen_active;
    int32_t *w1_of_word;               /* [n_w] index of the permanent single-phone channel, or -1 */
    pso_hmm_t *w1;                     /* [n1] */
    /* the utterance's vocabulary (build_fwdflat_wordlist) and channels (build_fwdflat_chan) */
    int32_t *node_sf, *node_wid, *node_fef, *node_lef, *node_next, *frm_head; int n_node, n_frame;
    int32_t *wordlist; int nwd;        /* fwdflat_wordlist */
    int32_t *wchain;                   /* [n_w] offset of the word's chain in chan[], -1 if it has none */
    int32_t *wlen;                     /* [n_w] chain length: 1 + (pronlen - 2) + n right contexts */
    pso_hmm_t *chan; int32_t *chan_rc; int n_chan;      /* chan_rc: right-context id, or -1 (root / word-internal) */
    int32_t *awl[2]; int32_t n_awl[2];
    uint8_t *word_active, *expand_flag;
    int32_t *expand; int n_expand;
    /* back-pointer table */
    pso_bp_t *bp; int32_t bpidx, bp_cap;
    int32_t *bss; int32_t bss_head, bss_cap;
    int32_t *word_lat_idx;
    int32_t *bp_table_idx; int32_t n_frame_alloc;
    int32_t best_score;
    int64_t n_eval, n_word_transition;
};

/* ---- hmm.c helpers ---- */
static void h_clear_scores(pso_hmm_t *h)                           /* hmm_clear_scores, hmm.c:167-179 */
{
    int i;
    for (i = 0; i < h->n_emit_state; ++i) h->score[i] = WORST;
    h->out_score = WORST; h->bestscore = WORST;
}
static void h_clear(pso_hmm_t *h)                                  /* hmm_clear, hmm.c:181-198 */
{
    int i;
    for (i = 0; i < h->n_emit_state; ++i) { h->score[i] = WORST; h->history[i] = -1; }
    h->out_score = WORST; h->out_history = -1; h->bestscore = WORST; h->frame = -1;
}
static void h_init(const pso_ff_t *s, pso_hmm_t *h, int mpx, int ssid, int tmatid)    /* hmm_init */
{
    int i;
    memset(h, 0, sizeof *h);
    h->mpx = (uint8_t)mpx; h->n_emit_state = (uint8_t)s->n_emit;
    if (mpx) {
        h->ssid = BAD_SSID; h->senid[0] = (uint16_t)ssid;
        for (i = 1; i < s->n_emit; ++i) h->senid[i] = BAD_SSID;
    }
    else {
        h->ssid = (uint16_t)ssid;
        for (i = 0; i < s->n_emit; ++i) h->senid[i] = s->t.ft.sseq[(size_t)ssid * s->n_emit + i];
    }
    h->tmatid = (int16_t)tmatid;
    h_clear(h);
}
static void h_enter(pso_hmm_t *h, int32_t score, int32_t hist, int frame) { h->score[0] = score; h->history[0] = hist; h->frame = frame; }
static void h_normalize(pso_hmm_t *h, int32_t norm)
{
    int i;
    for (i = 0; i < h->n_emit_state; ++i) if (h->score[i] > WORST) h->score[i] -= norm;
    if (h->out_score > WORST) h->out_score -= norm;
}

static int rs_n(const pso_ff_t *s, int last, int last2) { return s->t.ft.rssid_n[last * s->n_ci + last2]; }
static const int32_t *rs_cimap(const pso_ff_t *s, int last, int last2) { return s->t.ft.rssid_cimap + ((size_t)last * s->n_ci + last2) * s->n_ci; }

static int32_t lm_score(const pso_ff_t *s, int w3, int w2, int w1)    /* ngram_tg_score(...) >> SENSCR_SHIFT */
{
    const size_t n1 = (size_t)s->n_w + 1;
    if (s->trie) return pso_lm_tg_score(s->trie, w3, w2, w1, NULL) >> 10;
    return s->t.ft.lm[((size_t)w3 * n1 + (size_t)(w2 + 1)) * n1 + (size_t)(w1 + 1)];
}

/* set_real_wid, ngram_search.c:341-372 */
static void set_real_wid(pso_ff_t *s, int bp)
{
    pso_bp_t *e = &s->bp[bp], *prev = e->bp == NO_BP ? NULL : &s->bp[e->bp];
    if (s->t.ft.dict_filler[e->wid]) {
        if (prev) { e->real_wid = prev->real_wid; e->prev_real_wid = prev->prev_real_wid; }
        else { e->real_wid = s->t.ft.dict_basewid[e->wid]; e->prev_real_wid = -1; }
    }
    else {
        e->real_wid = s->t.ft.dict_basewid[e->wid];
        e->prev_real_wid = prev ? prev->real_wid : -1;
    }
}

/* ngram_search_save_bp, ngram_search.c:376-498 */
static void save_bp(pso_ff_t *s, int frame, int w, int32_t score, int32_t path, int rc)
{
    int bp = s->word_lat_idx[w];
    if (bp != NO_BP) {
        pso_bp_t *e = &s->bp[bp];
        if (e->score < score) {
            if (e->bp != path) {
                int32_t bplh[2], newlh[2];
                bplh[0] = e->bp == -1 ? -1 : s->bp[e->bp].prev_real_wid;
                bplh[1] = e->bp == -1 ? -1 : s->bp[e->bp].real_wid;
                newlh[0] = path == -1 ? -1 : s->bp[path].prev_real_wid;
                newlh[1] = path == -1 ? -1 : s->bp[path].real_wid;
                if (bplh[0] != newlh[0] || bplh[1] != newlh[1]) set_real_wid(s, bp);    /* with the old e->bp still in place */
                e->bp = path;
            }
            e->score = score;
        }
        if (e->s_idx != -1) s->bss[e->s_idx + rc] = score;
        return;
    }
    if (s->bpidx >= s->bp_cap) { s->bp_cap *= 2; s->bp = realloc(s->bp, sizeof *s->bp * s->bp_cap); }
    if (s->bss_head >= s->bss_cap - s->n_ci) { s->bss_cap *= 2; s->bss = realloc(s->bss, sizeof *s->bss * s->bss_cap); }
    {
        pso_bp_t *e = &s->bp[s->bpidx];
        int rcsize = 0, i;
        s->word_lat_idx[w] = s->bpidx;
        e->wid = w; e->frame = frame; e->bp = path; e->score = score; e->s_idx = s->bss_head; e->valid = 1;
        e->last_phone = s->t.ft.dict_last[w];
        if (s->t.ft.dict_pronlen[w] == 1) { e->last2_phone = -1; e->s_idx = -1; }
        else { e->last2_phone = s->t.ft.dict_last2[w]; rcsize = rs_n(s, e->last_phone, e->last2_phone); }
        for (i = 0; i < rcsize; ++i) s->bss[s->bss_head + i] = WORST;
        if (rcsize) s->bss[s->bss_head + rc] = score;
        set_real_wid(s, s->bpidx);
        s->bpidx++;
        s->bss_head += rcsize;
    }
}

static void mark_bptable(pso_ff_t *s, int frame)                   /* ngram_search_mark_bptable, ngram_search.c:301-339 */
{
    if (frame >= s->n_frame_alloc) {
        s->n_frame_alloc = (frame + 1) * 2;
        s->bp_table_idx = realloc(s->bp_table_idx, sizeof(int32_t) * (s->n_frame_alloc + 1));
    }
    s->bp_table_idx[frame] = s->bpidx;
}

pso_ff_t *pso_ff_new(const pso_ff_tables_t *t)
{
    pso_ff_t *s = calloc(1, sizeof *s);
    const int32_t *p = t->ft.par, *q = t->flat_par;
    int i, w;
    s->t = *t;
    s->n_ci = p[0]; s->n_emit = p[1]; s->n_sen = p[2]; s->n_w = p[3]; s->n1 = p[6];
    s->beam = p[8]; s->pip = p[13]; s->silpen = p[15]; s->fillpen = p[16];
    s->startwid = p[19]; s->finishwid = p[20]; s->silwid = p[21]; s->filler_start = p[22]; s->filler_end = p[23]; s->sil_ci = p[24];
    s->fwdflatbeam = q[0]; s->fwdflatwbeam = q[1]; s->min_ef_width = q[2]; s->max_sf_win = q[3];
    s->lwf = t->lwf;
    s->ctx.n_emit_state = s->n_emit; s->ctx.tp = t->ft.tp; s->ctx.sseq = t->ft.sseq;
    s->senscr = calloc(s->n_sen, sizeof(int16_t)); s->ctx.senscore = s->senscr;
    s->sen_active = calloc(s->n_sen, 1);
    s->w1_of_word = malloc(sizeof(int32_t) * s->n_w);
    for (w = 0; w < s->n_w; ++w) s->w1_of_word[w] = -1;
    s->w1 = calloc(s->n1 + 1, sizeof *s->w1);
    for (i = 0; i < s->n1; ++i) {
        s->w1_of_word[t->ft.w1_wid[i]] = i;
        h_init(s, &s->w1[i], t->ft.w1_mpx[i], t->ft.w1_ssid[i], t->ft.w1_tmat[i]);
    }
    s->wchain = malloc(sizeof(int32_t) * s->n_w); s->wlen = calloc(s->n_w, sizeof(int32_t));
    s->wordlist = calloc(s->n_w + 1, sizeof(int32_t));
    for (i = 0; i < 2; ++i) s->awl[i] = calloc(s->n_w + 1, sizeof(int32_t));
    s->word_active = calloc(s->n_w, 1); s->expand_flag = calloc(s->n_w, 1);
    s->expand = calloc(s->n_w + 1, sizeof(int32_t));
    s->bp_cap = 2048; s->bp = calloc(s->bp_cap, sizeof *s->bp);
    s->bss_cap = 2048 * 20; s->bss = calloc(s->bss_cap, sizeof(int32_t));
    s->word_lat_idx = calloc(s->n_w, sizeof(int32_t));
    s->n_frame_alloc = 256; s->bp_table_idx = calloc(s->n_frame_alloc + 1, sizeof(int32_t));
    return s;
}

static void free_utt(pso_ff_t *s)
{
    free(s->node_sf); free(s->node_wid); free(s->node_fef); free(s->node_lef); free(s->node_next); free(s->frm_head);
    free(s->chan); free(s->chan_rc);
    s->node_sf = s->node_wid = s->node_fef = s->node_lef = s->node_next = s->frm_head = NULL; s->chan = NULL; s->chan_rc = NULL;
}

void pso_ff_free(pso_ff_t *s)
{
    int i;
    if (!s) return;
    free_utt(s);
    free(s->senscr); free(s->sen_active); free(s->w1_of_word); free(s->w1); free(s->wchain); free(s->wlen); free(s->wordlist);
    for (i = 0; i < 2; ++i) free(s->awl[i]);
    free(s->word_active); free(s->expand_flag); free(s->expand); free(s->bp); free(s->bss); free(s->word_lat_idx);
    free(s->bp_table_idx);
    free(s);
}

void pso_ff_set_lm(pso_ff_t *s, const struct pso_lm_s *lm) { s->trie = lm; }

/* build_fwdflat_wordlist, ngram_search_fwdflat.c:223-300: one node per (start frame, word) of the first pass's
 * back-pointer table, new nodes at the HEAD of their start frame's list; nodes with too few end points (and </s>
 * not ending in the last frame) dropped; the vocabulary in order of first appearance walking the frames */
static void build_wordlist(pso_ff_t *s, const int32_t *bp1, int nb1)
{
    int i, f, n = 0;
    s->node_sf = malloc(sizeof(int32_t) * (nb1 + 1)); s->node_wid = malloc(sizeof(int32_t) * (nb1 + 1));
    s->node_fef = malloc(sizeof(int32_t) * (nb1 + 1)); s->node_lef = malloc(sizeof(int32_t) * (nb1 + 1));
    s->node_next = malloc(sizeof(int32_t) * (nb1 + 1)); s->frm_head = malloc(sizeof(int32_t) * (s->n_frame + 2));
    for (f = 0; f <= s->n_frame; ++f) s->frm_head[f] = -1;
    for (i = 0; i < nb1; ++i) {
        const int32_t *b = bp1 + (size_t)i * 10;
        const int sf = b[3] < 0 ? 0 : bp1[(size_t)b[3] * 10] + 1, ef = b[0], wid = b[2];
        int nd;
        if (!s->t.lm_known[wid]) continue;                          /* ngram_model_set_known_wid(basewid) */
        for (nd = s->frm_head[sf]; nd >= 0 && s->node_wid[nd] != wid; nd = s->node_next[nd]);
        if (nd >= 0) s->node_lef[nd] = ef;
        else {
            s->node_sf[n] = sf; s->node_wid[n] = wid; s->node_fef[n] = s->node_lef[n] = ef;
            s->node_next[n] = s->frm_head[sf]; s->frm_head[sf] = n;
            ++n;
        }
    }
    s->n_node = n;
    for (f = 0; f < s->n_frame; ++f) {
        int prev = -1, nd, next;
        for (nd = s->frm_head[f]; nd >= 0; nd = next) {
            next = s->node_next[nd];
            if (s->node_lef[nd] - s->node_fef[nd] < s->min_ef_width ||
                (s->node_wid[nd] == s->finishwid && s->node_lef[nd] < s->n_frame - 1)) {
                if (prev < 0) s->frm_head[f] = next; else s->node_next[prev] = next;
            }
            else prev = nd;
        }
    }
    s->nwd = 0;
    memset(s->word_active, 0, s->n_w);
    for (f = 0; f < s->n_frame; ++f) {
        int nd;
        for (nd = s->frm_head[f]; nd >= 0; nd = s->node_next[nd])
            if (!s->word_active[s->node_wid[nd]]) { s->word_active[s->node_wid[nd]] = 1; s->wordlist[s->nwd++] = s->node_wid[nd]; }
    }
    s->wordlist[s->nwd] = -1;
}

/* build_fwdflat_chan, :305-368: per multi-phone word of the vocabulary a multiplex root for the first phone,
 * one channel per word-internal phone, and the right-context fan-out of the last phone (ngram_search_alloc_all_rc,
 * ngram_search.c:583-633), here contiguous in that order */
static void build_chan(pso_ff_t *s)
{
    const pso_ft_tables_t *t = &s->t.ft;
    int i, tot = 0, w;
    for (w = 0; w < s->n_w; ++w) s->wchain[w] = -1;
    for (i = 0; i < s->nwd; ++i) {
        w = s->wordlist[i];
        if (t->dict_pronlen[w] == 1) continue;
        s->wlen[w] = 1 + (t->dict_pronlen[w] - 2) + rs_n(s, t->dict_last[w], t->dict_last2[w]);
        s->wchain[w] = tot; tot += s->wlen[w];
    }
    s->n_chan = tot;
    s->chan = calloc(tot + 1, sizeof *s->chan); s->chan_rc = malloc(sizeof(int32_t) * (tot + 1));
    for (i = 0; i < s->nwd; ++i) {
        int len, p, o, r, nrc, last, last2;
        w = s->wordlist[i];
        if ((o = s->wchain[w]) < 0) continue;
        len = t->dict_pronlen[w]; last = t->dict_last[w]; last2 = t->dict_last2[w]; nrc = rs_n(s, last, last2);
        h_init(s, &s->chan[o], 1, s->t.ci_ssid[t->dict_first[w]], t->ci_tmat[t->dict_first[w]]); s->chan_rc[o] = -1;
        ++o;
        for (p = 1; p < len - 1; ++p, ++o) {
            const int k = s->t.pron_off[w] + p;
            h_init(s, &s->chan[o], 0, s->t.pron_ssid[k], t->ci_tmat[s->t.pron_ci[k]]); s->chan_rc[o] = -1;
        }
        for (r = 0; r < nrc; ++r, ++o) {
            h_init(s, &s->chan[o], 0, t->rssid_ssid[((size_t)last * s->n_ci + last2) * s->n_ci + r], t->ci_tmat[last]);
            s->chan_rc[o] = r;
        }
    }
}

/* ngram_fwdflat_start, :370-414.  bp1 [nb1][10]: the first pass's back-pointer table (ref_dump's column order);
 * w1_ssid [n1][n_emit]: the per-state ssids the permanent single-phone channels hold when the first pass ends
 * (hmm_clear resets scores, histories and the frame, not these) */
void pso_ff_start(pso_ff_t *s, const int32_t *bp1, int nb1, int n_frame, const int32_t *w1_ssid)
{
    int i, k;
    free_utt(s);
    s->n_frame = n_frame;
    build_wordlist(s, bp1, nb1);
    build_chan(s);
    s->bpidx = 0; s->bss_head = 0;
    for (i = 0; i < s->n_w; ++i) s->word_lat_idx[i] = NO_BP;
    for (i = 0; i < s->n1; ++i) {
        h_clear(&s->w1[i]);
        if (w1_ssid && s->w1[i].mpx) for (k = 0; k < s->n_emit; ++k) s->w1[i].senid[k] = (uint16_t)w1_ssid[i * s->n_emit + k];
    }
    h_enter(&s->w1[s->w1_of_word[s->startwid]], 0, NO_BP, 0);
    s->awl[0][0] = s->startwid; s->n_awl[0] = 1; s->n_awl[1] = 0;
    s->best_score = 0;
    s->n_eval = s->n_word_transition = 0;
}

static pso_hmm_t *root_of(pso_ff_t *s, int w, int *len)
{
    if (s->wchain[w] >= 0) { *len = s->wlen[w]; return &s->chan[s->wchain[w]]; }
    *len = 1;
    return &s->w1[s->w1_of_word[w]];
}

static void activate(pso_ff_t *s, const pso_hmm_t *h)                /* acmod_activate_hmm, acmod.c:1179-1221 */
{
    int i;
    if (h->mpx) {
        for (i = 0; i < s->n_emit; ++i)
            if (h->senid[i] != BAD_SSID) s->sen_active[s->t.ft.sseq[(size_t)h->senid[i] * s->n_emit + i]] = 1;
    }
    else for (i = 0; i < s->n_emit; ++i) s->sen_active[h->senid[i]] = 1;
}

/* compute_fwdflat_sen_active, :416-442, + acmod_flags2list's bridging entries (acmod.c:1223-1275) */
int pso_ff_active_list(pso_ff_t *s, int frame, int32_t *out)
{
    int i, k, n = 0, last = 0, len;
    memset(s->sen_active, 0, s->n_sen);
    for (i = 0; i < s->n_awl[frame & 1]; ++i) {
        pso_hmm_t *h = root_of(s, s->awl[frame & 1][i], &len);
        for (k = 0; k < len; ++k) if (h[k].frame == frame) activate(s, &h[k]);
    }
    for (i = 0; i < s->n_sen; ++i) {
        if (!s->sen_active[i]) continue;
        while (i - last > 255) { last += 255; out[n++] = last; }
        out[n++] = i; last = i;
    }
    return n;
}

/* fwdflat_eval_chan, :444-480 */
static void eval_chan(pso_ff_t *s, int frame)
{
    int i, k, len;
    int32_t best = WORST;
    for (i = 0; i < s->n_awl[frame & 1]; ++i) {
        const int w = s->awl[frame & 1][i];
        pso_hmm_t *h = root_of(s, w, &len);
        for (k = 0; k < len; ++k) {
            int32_t sc;
            if (h[k].frame != frame) continue;
            sc = pso_hmm_vit_eval(&s->ctx, &h[k]);
            if (sc > best && !(k == 0 && w == s->finishwid)) best = sc;
            ++s->n_eval;
        }
    }
    s->best_score = best;
}

static void enter_if_better(pso_hmm_t *h, int32_t score, int32_t hist, int cf)
{
    if (h->frame < cf || score > h->score[0]) h_enter(h, score, hist, cf + 1);
}

/* fwdflat_prune_chan, :482-607 */
static void prune_chan(pso_ff_t *s, int cf)
{
    const int nf = cf + 1;
    const int32_t thresh = s->best_score + s->fwdflatbeam, wordthresh = s->best_score + s->fwdflatwbeam;
    int i, k, j, len;
    memset(s->word_active, 0, s->n_w);
    for (i = 0; i < s->n_awl[cf & 1]; ++i) {
        const int w = s->awl[cf & 1][i];
        pso_hmm_t *h = root_of(s, w, &len);
        const int32_t *rc = s->wchain[w] >= 0 ? s->chan_rc + s->wchain[w] : NULL;
        if (h[0].frame == cf && h[0].bestscore > thresh) {
            int32_t newscore = h[0].out_score;
            h[0].frame = nf;
            s->word_active[w] = 1;
            if (len > 1) {
                newscore += s->pip;
                if (newscore > thresh) {
                    if (rc[1] >= 0) for (j = 1; j < len; ++j) enter_if_better(&h[j], newscore, h[0].out_history, cf);
                    else enter_if_better(&h[1], newscore, h[0].out_history, cf);
                }
            }
            else if (newscore > wordthresh) save_bp(s, cf, w, newscore, h[0].out_history, 0);
        }
        for (k = 1; k < len; ++k) {
            if (h[k].frame < cf) continue;
            if (h[k].bestscore > thresh) {
                int32_t newscore = h[k].out_score;
                h[k].frame = nf;
                s->word_active[w] = 1;
                if (rc[k] < 0) {
                    newscore += s->pip;
                    if (newscore > thresh) {
                        if (rc[k + 1] >= 0) for (j = k + 1; j < len; ++j) enter_if_better(&h[j], newscore, h[k].out_history, cf);
                        else enter_if_better(&h[k + 1], newscore, h[k].out_history, cf);
                    }
                }
                else if (newscore > wordthresh) save_bp(s, cf, w, newscore, h[k].out_history, rc[k]);
            }
            else if (h[k].frame != nf) h_clear_scores(&h[k]);
        }
    }
}

/* get_expand_wordlist, :609-640 */
static void get_expand_wordlist(pso_ff_t *s, int frm, int win)
{
    int f, nd, sf = frm - win, ef = frm + win;
    if (sf < 0) sf = 0;
    if (ef > s->n_frame) ef = s->n_frame;
    memset(s->expand_flag, 0, s->n_w);
    s->n_expand = 0;
    for (f = sf; f < ef; ++f)
        for (nd = s->frm_head[f]; nd >= 0; nd = s->node_next[nd])
            if (!s->expand_flag[s->node_wid[nd]]) { s->expand[s->n_expand++] = s->node_wid[nd]; s->expand_flag[s->node_wid[nd]] = 1; }
    s->expand[s->n_expand] = -1;
    s->n_word_transition += s->n_expand;
}

/* fwdflat_word_transition, :642-782 */
static void word_transition(pso_ff_t *s, int cf)
{
    const pso_ft_tables_t *t = &s->t.ft;
    const int nf = cf + 1;
    const int32_t thresh = s->best_score + s->fwdflatbeam;
    int32_t best_silrc_score = WORST, best_silrc_bp = 0, newscore;
    int b, i, w, len;
    get_expand_wordlist(s, cf, s->max_sf_win);
    for (b = s->bp_table_idx[cf]; b < s->bpidx; ++b) {
        const pso_bp_t *e = &s->bp[b];
        const int32_t *rcss = s->bss + e->s_idx, *cimap = e->last2_phone == -1 ? NULL : rs_cimap(s, e->last_phone, e->last2_phone);
        int32_t silscore;
        s->word_lat_idx[e->wid] = NO_BP;
        if (e->wid == s->finishwid) continue;
        for (i = 0; i < s->n_expand; ++i) {
            pso_hmm_t *rh;
            w = s->expand[i];
            newscore = cimap ? rcss[cimap[t->dict_first[w]]] : e->score;
            if (newscore == WORST) continue;
            /* "newscore += lwf * (ngram_tg_score(...) >> SENSCR_SHIFT)": float arithmetic, truncated back to int32 (:700-706) */
            {
                volatile float prod = s->lwf * (float)lm_score(s, t->dict_basewid[w], e->real_wid, e->prev_real_wid);
                volatile float sum = (float)newscore + prod;
                newscore = (int32_t)sum;
            }
            newscore += s->pip;
            if (newscore > thresh) {
                rh = root_of(s, w, &len);
                if (rh->frame < cf || newscore > rh->score[0]) {
                    h_enter(rh, newscore, b, nf);
                    /* the root's (first phone, second phone) as build_fwdflat_chan / the single-phone set-up give them */
                    rh->senid[0] = (uint16_t)t->ldiph_lc[((size_t)t->dict_first[w] * s->n_ci +
                        (s->wchain[w] >= 0 ? s->t.pron_ci[s->t.pron_off[w] + 1] : t->w1_ci2[s->w1_of_word[w]])) * s->n_ci + t->dict_last[e->wid]];
                    s->word_active[w] = 1;
                }
            }
        }
        silscore = cimap ? rcss[cimap[s->sil_ci]] : e->score;
        if (silscore > best_silrc_score) { best_silrc_score = silscore; best_silrc_bp = b; }
    }
    newscore = best_silrc_score + s->silpen + s->pip;
    if (newscore > thresh && newscore > WORST) {
        pso_hmm_t *rh = &s->w1[s->w1_of_word[s->silwid]];
        if (rh->frame < cf || newscore > rh->score[0]) { h_enter(rh, newscore, best_silrc_bp, nf); s->word_active[s->silwid] = 1; }
    }
    newscore = best_silrc_score + s->fillpen + s->pip;
    if (newscore > thresh && newscore > WORST)
        for (w = s->filler_start; w <= s->filler_end; ++w) {
            pso_hmm_t *rh;
            if (w == s->silwid) continue;
            if (s->w1_of_word[w] < 0) continue;                  /* noise words that are not a single phone have no channel */
            rh = &s->w1[s->w1_of_word[w]];
            if (rh->frame < cf || newscore > rh->score[0]) { h_enter(rh, newscore, best_silrc_bp, nf); s->word_active[w] = 1; }
        }
    /* reset the initial channels of words that stayed inactive (:771-781) */
    for (i = 0; i < s->n_awl[cf & 1]; ++i) {
        pso_hmm_t *rh = root_of(s, s->awl[cf & 1][i], &len);
        if (rh->frame == cf) h_clear_scores(rh);
    }
}

static void renormalize(pso_ff_t *s, int cf, int32_t norm)           /* fwdflat_renormalize_scores, :784-810 */
{
    int i, k, len;
    for (i = 0; i < s->n_awl[cf & 1]; ++i) {
        pso_hmm_t *h = root_of(s, s->awl[cf & 1][i], &len);
        for (k = 0; k < len; ++k) if (h[k].frame == cf) h_normalize(&h[k], norm);
    }
}

/* ngram_fwdflat_search, :812-877, with the frame's scores given as (listed ids, their scores, the value of
 * every other entry) */
int pso_ff_step(pso_ff_t *s, int frame, const int32_t *ids, const int16_t *scr, int n, int16_t rest)
{
    int i, j = 0;
    int32_t *nawl;
    for (i = 0; i < s->n_sen; ++i) s->senscr[i] = rest;
    for (i = 0; i < n; ++i) s->senscr[ids[i]] = scr[i];
    mark_bptable(s, frame);
    if (s->best_score == WORST || s->best_score < WORST) return 0;
    if (s->best_score + 2 * s->beam < WORST) renormalize(s, frame, s->best_score);
    s->best_score = WORST;
    eval_chan(s, frame);
    prune_chan(s, frame);
    word_transition(s, frame);
    nawl = s->awl[(frame + 1) & 1];
    for (i = 0; i < s->nwd; ++i) {
        const int wid = s->wordlist[i];
        if (s->word_active[wid] && wid < s->startwid) nawl[j++] = wid;
    }
    for (i = s->startwid; i < s->n_w; ++i) if (s->word_active[i]) nawl[j++] = i;
    s->n_awl[(frame + 1) & 1] = j;
    return 1;
}

void pso_ff_finish(pso_ff_t *s, int n_frames) { mark_bptable(s, n_frames); }     /* ngram_fwdflat_finish, :925-960 */

int32_t pso_ff_best_score(const pso_ff_t *s) { return s->best_score; }
int32_t pso_ff_bpidx(const pso_ff_t *s) { return s->bpidx; }
int32_t pso_ff_bss_head(const pso_ff_t *s) { return s->bss_head; }
const pso_bp_t *pso_ff_bp(const pso_ff_t *s) { return s->bp; }
const int32_t *pso_ff_bss(const pso_ff_t *s) { return s->bss; }
const int32_t *pso_ff_bp_table_idx(const pso_ff_t *s) { return s->bp_table_idx; }
int32_t pso_ff_n_words(const pso_ff_t *s) { return s->nwd; }
const int32_t *pso_ff_wordlist(const pso_ff_t *s) { return s->wordlist; }
int32_t pso_ff_n_chan(const pso_ff_t *s) { return s->n_chan; }

/* test/debug aid: frame, score[0], history[0], senid[0] of word w's first channel */
void pso_ff_root_state(pso_ff_t *s, int w, int32_t *out)
{
    int len;
    const pso_hmm_t *h = root_of(s, w, &len);
    out[0] = h->frame; out[1] = h->score[0]; out[2] = h->history[0]; out[3] = h->senid[0]; out[4] = len;
}
/* test/debug aid: per channel of word w's chain: frame, score[0], bestscore, out_score */
int pso_ff_chain_state(pso_ff_t *s, int w, int32_t *out)
{
    int len, k;
    const pso_hmm_t *h = root_of(s, w, &len);
    for (k = 0; k < len; ++k) { out[4 * k] = h[k].frame; out[4 * k + 1] = h[k].score[0]; out[4 * k + 2] = h[k].bestscore; out[4 * k + 3] = h[k].out_score; }
    return len;
}
