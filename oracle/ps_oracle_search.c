/* oracle/ps_oracle_search.c -- TEST INFRASTRUCTURE (CPU oracle, part 2).
 *
 * A plain-C restatement of the reference's lexicon-tree search
 * (ngram_search_fwdtree.c, the back-pointer helpers of ngram_search.c) on flat,
 * index-based tables -- the layout a device-resident search would use.  It is
 * the oracle for SURVEY 8a row 17 (prune / transitions / bp table), which the
 * product does not implement yet; only tests/ may call it.  Everything static
 * (the tree create_search_channels built, dictionary, dict2pid, beams, the
 * language model as a dense table) comes out of the unmodified reference
 * through `ref_dump fwdtree`; per frame the oracle is handed the senone scores
 * and phone-loop penalties the reference's search was handed.
 *
 * Pinned by tests/test_oracle_search.py: back-pointer tables, score stacks,
 * per-frame marks and best scores identical to the reference's on the bundled
 * recordings.  Every function cites the reference lines it restates.
 */
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include "ps_oracle.h"
#include "ps_oracle_search.h"
#include "ps_oracle_lm.h"

#define WORST ((int32_t)0xE0000000)     /* WORST_SCORE, hmm.h:84 */
#define NO_BP (-1)
#define BAD_SSID 0xffff

typedef struct { int32_t wid, score, bp, next; } cand_t;          /* lastphn_cand_t, ngram_search.h:141-146 */
typedef struct { int32_t sf, dscr, bp; } ltrans_t;                /* last_ltrans_t :154-158 */
typedef struct { int32_t bp_ef, cand; } candsf_t;                 /* cand_sf_t :161-164 */
typedef struct { int32_t score, path, lc; } bestrc_t;             /* bestbp_rc_t :172-176 */

struct pso_ft_s {
    pso_ft_tables_t t;
    const pso_lm_t *trie;              /* optional: language scores from ps_oracle_lm.c instead of t.lm */
    int n_ci, n_emit, n_sen, n_w, R, M, N, n1, n1lm;
    int32_t beam, pbeam, lpbeam, lponlybeam, wbeam, pip, nwpen, silpen, fillpen, maxhmmpf, maxwpf;
    int32_t startwid, finishwid, silwid, filler_start, filler_end, sil_ci, has_pl;
    pso_hmm_ctx_t ctx;
    int16_t *senscr;                   /* [n_sen] the frame's scores */
    pso_hmm_t *node;                   /* [N] tree channels (roots are multiplex) */
    pso_hmm_t *w1;                     /* [n1] single-phone word channels */
    /* last-phone channels: word w owns slots wc_off[w] .. wc_off[w] + rssid_n - 1, one per right-context id */
    int32_t *wc_off;                   /* [n_w + 1] */
    pso_hmm_t *wc;
    uint8_t *wc_present;
    int32_t *acl[2]; int32_t n_acl[2]; /* active_chan_list */
    int32_t *awl[2]; int32_t n_awl[2]; /* active_word_list */
    uint8_t *word_active;
    cand_t *cand; int32_t n_cand;
    ltrans_t *ltrans;
    candsf_t *candsf;
    bestrc_t *bestrc;
    /* back-pointer table */
    pso_bp_t *bp; int32_t bpidx, bp_cap;
    int32_t *bss; int32_t bss_head, bss_cap;
    int32_t *word_lat_idx;
    int32_t *bp_table_idx; int32_t n_frame_alloc;
    int32_t best_score, last_phone_best_score, dynamic_beam, n_frame;
    int64_t n_root_eval, n_nonroot_eval;       /* ngram_search_stats_t counters the histogram pruning looks at */
    uint8_t *sen_active;               /* compute_sen_active flags */
    /* data-parallel formulation of the tree pruning (pso_ft_set_parallel) */
    int par_mode;
    int32_t *parent;                   /* [N] tree parent of a non-root node */
    int32_t *pos, *o_frame, *o_s0, *o_best, *o_out, *o_outh, *cnt, *offs;
    uint8_t *fire, *selfapp, *ret;
};

/* ---- hmm.c helpers on pso_hmm_t ---- */
static void h_clear(pso_hmm_t *h)                                  /* hmm_clear, hmm.c:181-196 */
{
    int i;
    for (i = 0; i < h->n_emit_state; ++i) { h->score[i] = WORST; h->history[i] = -1; }
    h->out_score = WORST; h->out_history = -1; h->bestscore = WORST; h->frame = -1;
}
static void h_init(pso_ft_t *s, pso_hmm_t *h, int mpx, int ssid, int tmatid)   /* hmm_init, hmm.c:146-168 */
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
        for (i = 0; i < s->n_emit; ++i) h->senid[i] = s->t.sseq[(size_t)ssid * s->n_emit + i];
    }
    h->tmatid = (int16_t)tmatid;
    h_clear(h);
}
static void h_enter(pso_hmm_t *h, int32_t score, int32_t hist, int frame)   /* hmm_enter, hmm.c:198-204 */
{
    h->score[0] = score; h->history[0] = hist; h->frame = frame;
}
static void h_normalize(pso_hmm_t *h, int32_t norm)               /* hmm_normalize, hmm.c:206-217 */
{
    int i;
    for (i = 0; i < h->n_emit_state; ++i) if (h->score[i] > WORST) h->score[i] -= norm;
    if (h->out_score > WORST) h->out_score -= norm;
}

static const int32_t *rs_cimap(const pso_ft_t *s, int last, int last2)
{
    return s->t.rssid_cimap + ((size_t)last * s->n_ci + last2) * s->n_ci;
}
static int rs_n(const pso_ft_t *s, int last, int last2) { return s->t.rssid_n[last * s->n_ci + last2]; }
static int32_t pen(const pso_ft_t *s, const int32_t *p, int ci) { return s->has_pl ? p[ci] : 0; }   /* phone_loop_search_score */

/* ngram_tg_score(lmset, w3, w2, w1) >> SENSCR_SHIFT: from the trie oracle (pso_ft_set_lm), else the dense table */
static int32_t lm_score(const pso_ft_t *s, int w3, int w2, int w1)
{
    const size_t n1 = (size_t)s->n_w + 1;
    if (s->trie)
        return pso_lm_tg_score(s->trie, w3, w2, w1, NULL) >> 10;
    return s->t.lm[((size_t)w3 * n1 + (size_t)(w2 + 1)) * n1 + (size_t)(w1 + 1)];
}

/* ngram_search_exit_score, ngram_search.c:653-674 */
static int32_t exit_score(const pso_ft_t *s, const pso_bp_t *e, int rcphone)
{
    if (e->last2_phone == -1) return e->score;
    return s->bss[e->s_idx + rs_cimap(s, e->last_phone, e->last2_phone)[rcphone]];
}

/* set_real_wid, ngram_search.c:341-372 */
static void set_real_wid(pso_ft_t *s, int bp)
{
    pso_bp_t *e = &s->bp[bp], *prev = e->bp == NO_BP ? NULL : &s->bp[e->bp];
    if (s->t.dict_filler[e->wid]) {
        if (prev) { e->real_wid = prev->real_wid; e->prev_real_wid = prev->prev_real_wid; }
        else { e->real_wid = s->t.dict_basewid[e->wid]; e->prev_real_wid = -1; }
    }
    else {
        e->real_wid = s->t.dict_basewid[e->wid];
        e->prev_real_wid = prev ? prev->real_wid : -1;
    }
}

/* ngram_search_save_bp, ngram_search.c:376-498 */
static void save_bp(pso_ft_t *s, int frame, int w, int32_t score, int32_t path, int rc)
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
                if (bplh[0] != newlh[0] || bplh[1] != newlh[1])
                    set_real_wid(s, bp);          /* NB: with the OLD e->bp still in place, as the reference does */
                e->bp = path;
            }
            e->score = score;
        }
        if (e->s_idx != -1) s->bss[e->s_idx + rc] = score;
        return;
    }
    if (s->bpidx >= s->bp_cap) {
        s->bp_cap *= 2;
        s->bp = realloc(s->bp, sizeof *s->bp * s->bp_cap);
    }
    if (s->bss_head >= s->bss_cap - s->n_ci) {
        s->bss_cap *= 2;
        s->bss = realloc(s->bss, sizeof *s->bss * s->bss_cap);
    }
    {
        pso_bp_t *e = &s->bp[s->bpidx];
        int rcsize = 0, i;
        s->word_lat_idx[w] = s->bpidx;
        e->wid = w; e->frame = frame; e->bp = path; e->score = score; e->s_idx = s->bss_head; e->valid = 1;
        e->last_phone = s->t.dict_last[w];
        if (s->t.dict_pronlen[w] == 1) { e->last2_phone = -1; e->s_idx = -1; }
        else {
            e->last2_phone = s->t.dict_last2[w];
            rcsize = rs_n(s, e->last_phone, e->last2_phone);
        }
        for (i = 0; i < rcsize; ++i) s->bss[s->bss_head + i] = WORST;
        if (rcsize) s->bss[s->bss_head + rc] = score;
        set_real_wid(s, s->bpidx);
        s->bpidx++;
        s->bss_head += rcsize;
    }
}

/* ngram_search_mark_bptable, ngram_search.c:301-339 */
static void mark_bptable(pso_ft_t *s, int frame)
{
    if (frame >= s->n_frame_alloc) {
        s->n_frame_alloc = (frame + 1) * 2;
        s->bp_table_idx = realloc(s->bp_table_idx, sizeof(int32_t) * (s->n_frame_alloc + 1));
    }
    s->bp_table_idx[frame] = s->bpidx;
}

/* ngram_search_alloc_all_rc, ngram_search.c:583-633: every right-context channel of w that is
 * not there yet is created fresh; the list order is the right-context id order */
static void alloc_all_rc(pso_ft_t *s, int w)
{
    int last = s->t.dict_last[w], last2 = s->t.dict_last2[w], n = rs_n(s, last, last2), i;
    const int32_t *ssid = s->t.rssid_ssid + ((size_t)last * s->n_ci + last2) * s->n_ci;
    for (i = 0; i < n; ++i) {
        int slot = s->wc_off[w] + i;
        if (!s->wc_present[slot]) {
            h_init(s, &s->wc[slot], 0, ssid[i], s->t.ci_tmat[last]);
            s->wc_present[slot] = 1;
        }
    }
}

pso_ft_t *pso_ft_new(const pso_ft_tables_t *t)
{
    pso_ft_t *s = calloc(1, sizeof *s);
    const int32_t *p = t->par;
    int i, w, tot = 0;
    s->t = *t;
    s->n_ci = p[0]; s->n_emit = p[1]; s->n_sen = p[2]; s->n_w = p[3]; s->R = p[4]; s->M = p[5]; s->N = s->R + s->M;
    s->n1 = p[6]; s->n1lm = p[7]; s->beam = p[8]; s->pbeam = p[9]; s->lpbeam = p[10]; s->lponlybeam = p[11];
    s->wbeam = p[12]; s->pip = p[13]; s->nwpen = p[14]; s->silpen = p[15]; s->fillpen = p[16]; s->maxhmmpf = p[17];
    s->maxwpf = p[18]; s->startwid = p[19]; s->finishwid = p[20]; s->silwid = p[21]; s->filler_start = p[22];
    s->filler_end = p[23]; s->sil_ci = p[24]; s->has_pl = p[25];
    s->ctx.n_emit_state = s->n_emit; s->ctx.tp = t->tp; s->ctx.sseq = t->sseq;
    s->senscr = calloc(s->n_sen, sizeof(int16_t));
    s->ctx.senscore = s->senscr;
    s->node = calloc(s->N, sizeof *s->node);
    for (i = 0; i < s->N; ++i) h_init(s, &s->node[i], i < s->R, t->node_ssid[i], t->node_tmat[i]);
    s->w1 = calloc(s->n1 + 1, sizeof *s->w1);
    for (i = 0; i < s->n1; ++i) h_init(s, &s->w1[i], t->w1_mpx[i], t->w1_ssid[i], t->w1_tmat[i]);
    s->wc_off = calloc(s->n_w + 1, sizeof(int32_t));
    for (w = 0; w < s->n_w; ++w) {
        s->wc_off[w] = tot;
        if (t->dict_pronlen[w] > 1) tot += rs_n(s, t->dict_last[w], t->dict_last2[w]);
    }
    s->wc_off[s->n_w] = tot;
    s->wc = calloc(tot + 1, sizeof *s->wc);
    s->wc_present = calloc(tot + 1, 1);
    for (i = 0; i < 2; ++i) { s->acl[i] = calloc(s->N + 1, sizeof(int32_t)); s->awl[i] = calloc(s->n_w + 1, sizeof(int32_t)); }
    s->word_active = calloc(s->n_w, 1);
    s->cand = calloc(s->n_w + 1, sizeof *s->cand);
    s->ltrans = calloc(s->n_w, sizeof *s->ltrans);
    s->candsf = calloc(s->n_w + 1, sizeof *s->candsf);
    s->bestrc = calloc(s->n_ci, sizeof *s->bestrc);
    s->bp_cap = 2048; s->bp = calloc(s->bp_cap, sizeof *s->bp);
    s->bss_cap = 2048 * 20; s->bss = calloc(s->bss_cap, sizeof(int32_t));
    s->word_lat_idx = calloc(s->n_w, sizeof(int32_t));
    s->n_frame_alloc = 256; s->bp_table_idx = calloc(s->n_frame_alloc + 1, sizeof(int32_t));
    s->sen_active = calloc(s->n_sen, 1);
    s->parent = calloc(s->N, sizeof(int32_t));
    for (i = 0; i < s->N; ++i) s->parent[i] = -1;
    for (i = 0; i < s->N; ++i) {
        int c;
        for (c = t->node_child[i]; c >= 0; c = t->node_sib[c]) s->parent[c] = i;
    }
    s->pos = calloc(s->N, 4); s->o_frame = calloc(s->N, 4); s->o_s0 = calloc(s->N, 4); s->o_best = calloc(s->N, 4);
    s->o_out = calloc(s->N, 4); s->o_outh = calloc(s->N, 4); s->cnt = calloc(s->N + 1, 4); s->offs = calloc(s->N + 2, 4);
    s->fire = calloc(s->N, 1); s->selfapp = calloc(s->N, 1); s->ret = calloc(s->N, 1);
    return s;
}

void pso_ft_free(pso_ft_t *s)
{
    int i;
    if (!s) return;
    free(s->senscr); free(s->node); free(s->w1); free(s->wc_off); free(s->wc); free(s->wc_present);
    for (i = 0; i < 2; ++i) { free(s->acl[i]); free(s->awl[i]); }
    free(s->word_active); free(s->cand); free(s->ltrans); free(s->candsf); free(s->bestrc);
    free(s->bp); free(s->bss); free(s->word_lat_idx); free(s->bp_table_idx); free(s->sen_active);
    free(s->parent); free(s->pos); free(s->o_frame); free(s->o_s0); free(s->o_best); free(s->o_out); free(s->o_outh);
    free(s->cnt); free(s->offs); free(s->fire); free(s->selfapp); free(s->ret);
    free(s);
}

static int w1_index(const pso_ft_t *s, int w)
{
    int i;
    for (i = 0; i < s->n1; ++i) if (s->t.w1_wid[i] == w) return i;
    return -1;
}

/* ngram_fwdtree_start, ngram_search_fwdtree.c:469-520 */
void pso_ft_start(pso_ft_t *s)
{
    int i;
    s->bpidx = 0; s->bss_head = 0;
    for (i = 0; i < s->n_w; ++i) s->word_lat_idx[i] = NO_BP;
    s->n_acl[0] = s->n_acl[1] = 0; s->n_awl[0] = s->n_awl[1] = 0;
    s->best_score = 0;
    for (i = 0; i < s->n_w; ++i) s->ltrans[i].sf = -1;
    s->n_frame = 0;
    s->n_root_eval = s->n_nonroot_eval = 0;
    for (i = 0; i < s->N; ++i) s->pos[i] = -1;               /* (prune_tree_list keeps it so between frames) */
    for (i = 0; i < s->n1; ++i) h_clear(&s->w1[i]);
    i = w1_index(s, s->startwid);
    h_clear(&s->w1[i]);
    h_enter(&s->w1[i], 0, NO_BP, 0);
}

/* A decoder session's carry-over into a NEW oracle object: the per-state ssids the permanent multiplexed channels (roots,
 * then single-phone words; [R + n1][n_emit]) hold when the utterance starts -- hmm_clear (hmm.c:181-196) does not reset
 * them.  (An oracle object that is re-started for the next utterance carries them by itself, as the reference does.)
 * Call after pso_ft_start. */
void pso_ft_set_mpx_ssids(pso_ft_t *s, const int32_t *ssid)
{
    int i, k;
    for (i = 0; i < s->R; ++i)
        for (k = 0; k < s->n_emit; ++k) s->node[i].senid[k] = (uint16_t)ssid[(size_t)i * s->n_emit + k];
    for (i = 0; i < s->n1; ++i)
        if (s->w1[i].mpx)
            for (k = 0; k < s->n_emit; ++k) s->w1[i].senid[k] = (uint16_t)ssid[(size_t)(s->R + i) * s->n_emit + k];
}
void pso_ft_get_mpx_ssids(const pso_ft_t *s, int32_t *ssid)
{
    int i, k;
    for (i = 0; i < s->R; ++i)
        for (k = 0; k < s->n_emit; ++k) ssid[(size_t)i * s->n_emit + k] = s->node[i].senid[k];
    for (i = 0; i < s->n1; ++i)
        for (k = 0; k < s->n_emit; ++k) ssid[(size_t)(s->R + i) * s->n_emit + k] = s->w1[i].senid[k];
}

/* acmod_activate_hmm, acmod.c:1179-1221 */
static void activate(pso_ft_t *s, const pso_hmm_t *h)
{
    int i;
    if (h->mpx) {
        for (i = 0; i < s->n_emit; ++i)
            if (h->senid[i] != BAD_SSID) s->sen_active[s->t.sseq[(size_t)h->senid[i] * s->n_emit + i]] = 1;
    }
    else
        for (i = 0; i < s->n_emit; ++i) s->sen_active[h->senid[i]] = 1;
}

/* compute_sen_active, ngram_search_fwdtree.c:526-564, + acmod_flags2list's bridging entries
 * (acmod.c:1223-1275): the ids of the list the scorer is given.  Returns its length. */
int pso_ft_active_list(pso_ft_t *s, int frame, int32_t *out)
{
    int i, k, w, n = 0, last = 0;
    memset(s->sen_active, 0, s->n_sen);
    for (i = 0; i < s->R; ++i) if (s->node[i].frame == frame) activate(s, &s->node[i]);
    for (i = 0; i < s->n_acl[frame & 1]; ++i) activate(s, &s->node[s->acl[frame & 1][i]]);
    for (i = 0; i < s->n_awl[frame & 1]; ++i) {
        w = s->awl[frame & 1][i];
        for (k = s->wc_off[w]; k < s->wc_off[w + 1]; ++k) if (s->wc_present[k]) activate(s, &s->wc[k]);
    }
    for (i = 0; i < s->n1; ++i) if (s->w1[i].frame == frame) activate(s, &s->w1[i]);
    for (i = 0; i < s->n_sen; ++i) {
        if (!s->sen_active[i]) continue;
        while (i - last > 255) { last += 255; out[n++] = last; }
        out[n++] = i; last = i;
    }
    return n;
}

/* renormalize_scores, :566-603 */
static void renormalize(pso_ft_t *s, int frame, int32_t norm)
{
    int i, k, w;
    for (i = 0; i < s->R; ++i) if (s->node[i].frame == frame) h_normalize(&s->node[i], norm);
    for (i = 0; i < s->n_acl[frame & 1]; ++i) h_normalize(&s->node[s->acl[frame & 1][i]], norm);
    for (i = 0; i < s->n_awl[frame & 1]; ++i) {
        w = s->awl[frame & 1][i];
        for (k = s->wc_off[w]; k < s->wc_off[w + 1]; ++k) if (s->wc_present[k]) h_normalize(&s->wc[k], norm);
    }
    for (i = 0; i < s->n1; ++i) if (s->w1[i].frame == frame) h_normalize(&s->w1[i], norm);
}

/* evaluate_channels = eval_root_chan + eval_nonroot_chan + eval_word_chan, :605-715 */
static void evaluate(pso_ft_t *s, int frame)
{
    int i, k, w, kk = 0, j = 0;
    int32_t bs = WORST, sc;
    for (i = 0; i < s->R; ++i)
        if (s->node[i].frame == frame) {
            sc = pso_hmm_vit_eval(&s->ctx, &s->node[i]);
            if (sc > bs) bs = sc;
            ++s->n_root_eval;
        }
    s->best_score = bs;
    bs = WORST;
    s->n_nonroot_eval += s->n_acl[frame & 1];
    for (i = 0; i < s->n_acl[frame & 1]; ++i) {
        sc = pso_hmm_vit_eval(&s->ctx, &s->node[s->acl[frame & 1][i]]);
        if (sc > bs) bs = sc;
    }
    if (bs > s->best_score) s->best_score = bs;
    bs = WORST;
    for (i = 0; i < s->n_awl[frame & 1]; ++i) {
        w = s->awl[frame & 1][i];
        s->word_active[w] = 0;
        for (k = s->wc_off[w]; k < s->wc_off[w + 1]; ++k)
            if (s->wc_present[k]) {
                sc = pso_hmm_vit_eval(&s->ctx, &s->wc[k]);
                if (sc > bs) bs = sc;
                ++kk;
            }
    }
    for (i = 0; i < s->n1; ++i) {
        if (s->w1[i].frame < frame) continue;
        sc = pso_hmm_vit_eval(&s->ctx, &s->w1[i]);
        if (sc > bs && s->t.w1_wid[i] != s->finishwid) bs = sc;
        ++j;
    }
    s->n_nonroot_eval += kk + j;
    if (bs > s->best_score) s->best_score = bs;
    s->last_phone_best_score = bs;
}

/* the part prune_root_chan and prune_nonroot_chan share: transitions out of tree node `src`
 * (:747-785, :826-870).  `root` selects the small differences between the two. */
static void node_transitions(pso_ft_t *s, int frame, const int32_t *pp, pso_hmm_t *h, int first_child, int penult_wid,
                             int root, int32_t **nacl)
{
    const int nf = frame + 1;
    const int32_t newphone_thresh = s->best_score + s->pbeam, lastphn_thresh = s->best_score + s->lpbeam;
    const int32_t newphone_score = h->out_score + s->pip;
    int c, w;
    if (s->has_pl || newphone_score > newphone_thresh) {
        for (c = first_child; c >= 0; c = s->t.node_sib[c]) {
            pso_hmm_t *nh = &s->node[c];
            const int32_t pl = newphone_score + pen(s, pp, s->t.node_ci[c]);
            if (pl > newphone_thresh && (nh->frame < frame || newphone_score > nh->score[0])) {
                if (root) {                                   /* :756-759: entered, then always listed */
                    h_enter(nh, newphone_score, h->out_history, nf);
                    *((*nacl)++) = c;
                }
                else {                                        /* :840-846: listed only if not already there */
                    if (nh->frame != nf) *((*nacl)++) = c;
                    h_enter(nh, newphone_score, h->out_history, nf);
                }
            }
        }
    }
    if (s->has_pl || newphone_score > lastphn_thresh) {
        for (w = penult_wid; w >= 0; w = s->t.homophone_set[w]) {
            const int32_t pl = newphone_score + pen(s, pp, s->t.dict_last[w]);
            if (pl > lastphn_thresh) {
                cand_t *cp = &s->cand[s->n_cand++];
                cp->wid = w; cp->score = newphone_score - s->nwpen; cp->bp = h->out_history;
            }
        }
    }
}

/* prune_root_chan, :722-790 */
static void prune_root(pso_ft_t *s, int frame, const int32_t *pp)
{
    const int nf = frame + 1;
    const int32_t thresh = s->best_score + s->dynamic_beam;
    int32_t *nacl = s->acl[nf & 1];
    int i;
    for (i = 0; i < s->R; ++i) {
        pso_hmm_t *h = &s->node[i];
        if (h->frame < frame) continue;
        if (h->bestscore > thresh) {
            h->frame = nf;
            node_transitions(s, frame, pp, h, s->t.node_child[i], s->t.node_penult_wid[i], 1, &nacl);
        }
    }
    s->n_acl[nf & 1] = (int32_t)(nacl - s->acl[nf & 1]);
}

/* prune_nonroot_chan, :796-877 */
static void prune_nonroot(pso_ft_t *s, int frame, const int32_t *pp)
{
    const int nf = frame + 1;
    const int32_t thresh = s->best_score + s->dynamic_beam;
    int32_t *acl = s->acl[frame & 1], *nacl = s->acl[nf & 1] + s->n_acl[nf & 1];
    int i;
    for (i = 0; i < s->n_acl[frame & 1]; ++i) {
        const int c = acl[i];
        pso_hmm_t *h = &s->node[c];
        if (h->bestscore > thresh) {
            if (h->frame != nf) { h->frame = nf; *(nacl++) = c; }
            node_transitions(s, frame, pp, h, s->t.node_child[c], s->t.node_penult_wid[c], 0, &nacl);
        }
        else if (h->frame != nf)
            h_clear(h);
    }
    s->n_acl[nf & 1] = (int32_t)(nacl - s->acl[nf & 1]);
}


/* ---------------------------------------------------------------------------------------
 * The same tree pruning (prune_root_chan + prune_nonroot_chan) written as independent
 * per-node computations on a snapshot of the state after evaluation, plus prefix sums for
 * the list positions -- the form a device kernel can run with one lane per node.  The
 * reference walks the active list sequentially; what that order decides is enumerable,
 * because a tree node has exactly one parent:
 *   - a node goes on the next active list either by its own retention or by its parent's
 *     transition into it, whichever comes first in list order (roots come before everything);
 *   - a node that fails the beam is cleared at its own turn unless its parent entered it
 *     earlier; if the parent comes later it enters a cleared node unconditionally.
 * Equivalence with the sequential walk is what tests/test_oracle_search.py checks
 * (pso_ft_set_parallel(1): identical back-pointer tables on every golden decode).
 * --------------------------------------------------------------------------------------- */
static void prune_tree_parallel(pso_ft_t *s, int frame, const int32_t *pp)
{
    const int nf = frame + 1, N = s->N, R = s->R;
    const int32_t thresh = s->best_score + s->dynamic_beam;
    const int32_t npt = s->best_score + s->pbeam, lpt = s->best_score + s->lpbeam;
    const int32_t *acl = s->acl[frame & 1];
    const int n_acl = s->n_acl[frame & 1];
    int32_t *nacl = s->acl[nf & 1];
    int i, p, c, w, total;

    /* snapshot + positions */
    for (i = 0; i < N; ++i) {
        const pso_hmm_t *h = &s->node[i];
        s->pos[i] = -1; s->o_frame[i] = h->frame; s->o_s0[i] = h->score[0]; s->o_best[i] = h->bestscore;
        s->o_out[i] = h->out_score; s->o_outh[i] = h->out_history;
    }
    for (p = 0; p < n_acl; ++p) s->pos[acl[p]] = p;
    /* retention of every node (roots: active this frame; others: on the list) */
    for (i = 0; i < N; ++i) {
        const int active = i < R ? s->o_frame[i] >= frame : s->pos[i] >= 0;
        s->ret[i] = (uint8_t)(active && s->o_best[i] > thresh);
    }
    /* every non-root node decides its own fate from its own and its parent's snapshot */
    for (c = R; c < N; ++c) {
        pso_hmm_t *h = &s->node[c];
        const int P = s->parent[c], pc = s->pos[c], in_acl = pc >= 0;
        const int32_t news = s->o_out[P] + s->pip;
        const int par_active = P < R ? 1 : s->pos[P] >= 0;
        const int parent_can = par_active && s->ret[P] && (s->has_pl || news > npt) && (news + pen(s, pp, s->t.node_ci[c]) > npt);
        const int parent_first = P < R || (in_acl && s->pos[P] < pc) || !in_acl;
        int fire, entered_first = 0, cleared = 0;
        if (!in_acl)                       fire = parent_can && (s->o_frame[c] < frame || news > s->o_s0[c]);
        else if (parent_first)             fire = parent_can && (s->o_frame[c] < frame || news > s->o_s0[c]);
        else if (s->ret[c])                fire = parent_can && news > s->o_s0[c];
        else                               fire = parent_can;          /* cleared at its own, earlier turn */
        if (fire && parent_first) entered_first = 1;
        s->selfapp[c] = (uint8_t)(in_acl && s->ret[c] && !entered_first);
        /* listed by the parent unless it put itself on the list earlier; a root's transition lists always */
        s->fire[c] = (uint8_t)(fire ? ((P < R || !(in_acl && !parent_first && s->ret[c])) ? 1 : 2) : 0);
        if (in_acl && !s->ret[c] && !entered_first) cleared = 1;
        /* new state */
        if (cleared) h_clear(h);
        if (in_acl && s->ret[c]) h->frame = nf;
        if (fire) h_enter(h, news, s->o_outh[P], nf);
    }
    for (i = 0; i < R; ++i) if (s->ret[i]) s->node[i].frame = nf;
    /* list positions: the root phase, then one segment per list position */
    total = 0;
    for (i = 0; i < R; ++i)
        for (c = s->t.node_child[i]; c >= 0; c = s->t.node_sib[c])
            if (s->fire[c]) nacl[total++] = c;              /* roots list every child they enter (:756-759) */
    for (p = 0; p < n_acl; ++p) {
        int k = s->selfapp[acl[p]];
        for (c = s->t.node_child[acl[p]]; c >= 0; c = s->t.node_sib[c]) k += (s->fire[c] == 1);
        s->cnt[p] = k;
    }
    s->offs[0] = total;
    for (p = 0; p < n_acl; ++p) s->offs[p + 1] = s->offs[p] + s->cnt[p];     /* exclusive prefix sum */
    for (p = 0; p < n_acl; ++p) {
        int o = s->offs[p];
        if (s->selfapp[acl[p]]) nacl[o++] = acl[p];
        for (c = s->t.node_child[acl[p]]; c >= 0; c = s->t.node_sib[c]) if (s->fire[c] == 1) nacl[o++] = c;
    }
    s->n_acl[nf & 1] = s->offs[n_acl];
    /* last-phone candidates: list order, homophone chain inside (order-free given the snapshot) */
    for (i = 0; i < R + n_acl; ++i) {
        const int node = i < R ? i : acl[i - R];
        const int32_t news = s->o_out[node] + s->pip;
        if (!s->ret[node] || !(s->has_pl || news > lpt)) continue;
        for (w = s->t.node_penult_wid[node]; w >= 0; w = s->t.homophone_set[w])
            if (news + pen(s, pp, s->t.dict_last[w]) > lpt) {
                cand_t *cp = &s->cand[s->n_cand++];
                cp->wid = w; cp->score = news - s->nwpen; cp->bp = s->o_outh[node];
            }
    }
}

/* The same decisions with work proportional to the ACTIVE part of the tree (pso_ft_set_parallel(2)): the form a
 * large-vocabulary kernel needs (248 k channels, ~9 k active per frame).  Work items are the nodes on the active
 * list (own retention / clearing, in_acl = 1) and the children of roots and of listed nodes that are not listed
 * themselves (entered by the parent or left alone, in_acl = 0; a node has one parent, so no item is visited
 * twice).  The per-node arrays (pos, snapshot, ret, fire, selfapp) are sized for the whole tree but touched only
 * at these items; pos is reset at the end.  Every read of another node's state goes to the snapshot of an
 * active node or root, every write to the item's own node, so the items are independent. */
static void decide_node(pso_ft_t *s, int c, int frame, const int32_t *pp, int32_t npt)
{
    const int nf = frame + 1, R = s->R;
    pso_hmm_t *h = &s->node[c];
    const int P = s->parent[c], pc = s->pos[c], in_acl = pc >= 0;
    const int par_active = P < R ? 1 : s->pos[P] >= 0;
    const int32_t news = (par_active ? s->o_out[P] : WORST) + s->pip;
    const int parent_can = par_active && s->ret[P] && (s->has_pl || news > npt) && (news + pen(s, pp, s->t.node_ci[c]) > npt);
    const int parent_first = P < R || (in_acl && s->pos[P] < pc) || !in_acl;
    const int retc = in_acl && s->ret[c];
    int fire, entered_first;
    if (!in_acl || parent_first) fire = parent_can && (h->frame < frame || news > h->score[0]);   /* own state: live */
    else if (retc)               fire = parent_can && news > h->score[0];
    else                         fire = parent_can;
    entered_first = fire && parent_first;
    s->selfapp[c] = (uint8_t)(retc && !entered_first);
    s->fire[c] = (uint8_t)(fire ? ((P < R || !(in_acl && !parent_first && retc)) ? 1 : 2) : 0);
    if (in_acl && !retc && !entered_first) h_clear(h);
    if (retc) h->frame = nf;
    if (fire) h_enter(h, news, s->o_outh[P], nf);
}

static void prune_tree_list(pso_ft_t *s, int frame, const int32_t *pp)
{
    const int nf = frame + 1, R = s->R;
    const int32_t thresh = s->best_score + s->dynamic_beam;
    const int32_t npt = s->best_score + s->pbeam, lpt = s->best_score + s->lpbeam;
    const int32_t *acl = s->acl[frame & 1];
    const int n_acl = s->n_acl[frame & 1];
    int32_t *nacl = s->acl[nf & 1];
    int i, p, c, w, total;

    /* positions, snapshot and retention of roots and listed nodes */
    for (p = 0; p < n_acl; ++p) s->pos[acl[p]] = p;
    for (i = 0; i < R + n_acl; ++i) {
        const int node = i < R ? i : acl[i - R];
        const pso_hmm_t *h = &s->node[node];
        const int active = node < R ? h->frame >= frame : 1;
        s->o_out[node] = h->out_score; s->o_outh[node] = h->out_history;
        s->ret[node] = (uint8_t)(active && h->bestscore > thresh);
    }
    /* decisions: listed nodes, then the unlisted children of roots and listed nodes */
    for (p = 0; p < n_acl; ++p) decide_node(s, acl[p], frame, pp, npt);
    for (i = 0; i < R + n_acl; ++i) {
        const int node = i < R ? i : acl[i - R];
        for (c = s->t.node_child[node]; c >= 0; c = s->t.node_sib[c])
            if (s->pos[c] < 0) decide_node(s, c, frame, pp, npt);
    }
    for (i = 0; i < R; ++i) if (s->ret[i]) s->node[i].frame = nf;
    /* list positions, as in prune_tree_parallel */
    total = 0;
    for (i = 0; i < R; ++i)
        for (c = s->t.node_child[i]; c >= 0; c = s->t.node_sib[c])
            if (s->fire[c]) nacl[total++] = c;
    for (p = 0; p < n_acl; ++p) {
        int k = s->selfapp[acl[p]];
        for (c = s->t.node_child[acl[p]]; c >= 0; c = s->t.node_sib[c]) k += (s->fire[c] == 1);
        s->cnt[p] = k;
    }
    s->offs[0] = total;
    for (p = 0; p < n_acl; ++p) s->offs[p + 1] = s->offs[p] + s->cnt[p];
    for (p = 0; p < n_acl; ++p) {
        int o = s->offs[p];
        if (s->selfapp[acl[p]]) nacl[o++] = acl[p];
        for (c = s->t.node_child[acl[p]]; c >= 0; c = s->t.node_sib[c]) if (s->fire[c] == 1) nacl[o++] = c;
    }
    s->n_acl[nf & 1] = s->offs[n_acl];
    for (i = 0; i < R + n_acl; ++i) {
        const int node = i < R ? i : acl[i - R];
        const int32_t news = s->o_out[node] + s->pip;
        if (!s->ret[node] || !(s->has_pl || news > lpt)) continue;
        for (w = s->t.node_penult_wid[node]; w >= 0; w = s->t.homophone_set[w])
            if (news + pen(s, pp, s->t.dict_last[w]) > lpt) {
                cand_t *cp = &s->cand[s->n_cand++];
                cp->wid = w; cp->score = news - s->nwpen; cp->bp = s->o_outh[node];
            }
    }
    for (p = 0; p < n_acl; ++p) s->pos[acl[p]] = -1;
}

void pso_ft_set_lm(pso_ft_t *s, const pso_lm_t *lm) { s->trie = lm; }

void pso_ft_set_parallel(pso_ft_t *s, int on) { s->par_mode = on; }

/* last_phone_transition, :884-1032 */
static void last_phone_transition(pso_ft_t *s, int frame)
{
    const int nf = frame + 1;
    int32_t *nawl = s->awl[nf & 1];
    int i, j, k, n_cand_sf = 0, bp, bpend, w;
    int32_t bestscore, thresh;
    for (i = 0; i < s->n_cand; ++i) {
        cand_t *cp = &s->cand[i];
        const pso_bp_t *e;
        if (cp->bp == -1) continue;
        e = &s->bp[cp->bp];
        cp->score -= exit_score(s, e, s->t.dict_first[cp->wid]);
        if (s->ltrans[cp->wid].sf != e->frame + 1) {
            for (j = 0; j < n_cand_sf; ++j) if (s->candsf[j].bp_ef == e->frame) break;
            if (j < n_cand_sf) cp->next = s->candsf[j].cand;
            else { j = n_cand_sf++; cp->next = -1; s->candsf[j].bp_ef = e->frame; }
            s->candsf[j].cand = i;
            s->ltrans[cp->wid].dscr = WORST;
            s->ltrans[cp->wid].sf = e->frame + 1;
        }
    }
    for (i = 0; i < n_cand_sf; ++i) {
        bp = s->bp_table_idx[s->candsf[i].bp_ef];
        bpend = s->bp_table_idx[s->candsf[i].bp_ef + 1];
        for (; bp < bpend; ++bp) {
            const pso_bp_t *e = &s->bp[bp];
            if (!e->valid) continue;
            for (j = s->candsf[i].cand; j >= 0; j = s->cand[j].next) {
                cand_t *cp = &s->cand[j];
                int32_t dscr = exit_score(s, e, s->t.dict_first[cp->wid]);
                if (dscr > WORST)
                    dscr += lm_score(s, s->t.dict_basewid[cp->wid], e->real_wid, e->prev_real_wid);
                if (dscr > s->ltrans[cp->wid].dscr) { s->ltrans[cp->wid].dscr = dscr; s->ltrans[cp->wid].bp = bp; }
            }
        }
    }
    bestscore = s->last_phone_best_score;
    for (i = 0; i < s->n_cand; ++i) {
        cand_t *cp = &s->cand[i];
        cp->score += s->ltrans[cp->wid].dscr;
        cp->bp = s->ltrans[cp->wid].bp;
        if (cp->score > bestscore) bestscore = cp->score;
    }
    s->last_phone_best_score = bestscore;
    thresh = bestscore + s->lponlybeam;
    for (i = 0; i < s->n_cand; ++i) {
        cand_t *cp = &s->cand[i];
        if (cp->score > thresh) {
            w = cp->wid;
            alloc_all_rc(s, w);
            k = 0;
            for (j = s->wc_off[w]; j < s->wc_off[w + 1]; ++j) {
                pso_hmm_t *h = &s->wc[j];
                if (!s->wc_present[j]) continue;
                if (h->frame < frame || cp->score > h->score[0]) { h_enter(h, cp->score, cp->bp, nf); ++k; }
            }
            if (k > 0) { *(nawl++) = w; s->word_active[w] = 1; }
        }
    }
    s->n_awl[nf & 1] = (int32_t)(nawl - s->awl[nf & 1]);
}

/* prune_word_chan, :1038-1128 */
static void prune_word(pso_ft_t *s, int frame)
{
    const int nf = frame + 1;
    const int32_t newword_thresh = s->last_phone_best_score + s->wbeam;
    const int32_t lastphn_thresh = s->last_phone_best_score + s->lponlybeam;
    int32_t *awl = s->awl[frame & 1], *nawl = s->awl[nf & 1] + s->n_awl[nf & 1];
    int i, j, k, w;
    for (i = 0; i < s->n_awl[frame & 1]; ++i) {
        w = awl[i];
        k = 0;
        for (j = s->wc_off[w]; j < s->wc_off[w + 1]; ++j) {
            pso_hmm_t *h = &s->wc[j];
            if (!s->wc_present[j]) continue;
            if (h->bestscore > lastphn_thresh) {
                h->frame = nf;
                ++k;
                if (h->out_score > newword_thresh)
                    save_bp(s, frame, w, h->out_score, h->out_history, j - s->wc_off[w]);
            }
            else if (h->frame == nf) { /* entered this frame by last_phone_transition: stays */ }
            else s->wc_present[j] = 0;                      /* hmm_deinit + listelem_free */
        }
        if (k > 0 && !s->word_active[w]) { *(nawl++) = w; s->word_active[w] = 1; }
    }
    s->n_awl[nf & 1] = (int32_t)(nawl - s->awl[nf & 1]);
    for (i = 0; i < s->n1; ++i) {
        pso_hmm_t *h = &s->w1[i];
        if (h->frame < frame) continue;
        if (h->bestscore > lastphn_thresh) {
            h->frame = nf;
            if (h->out_score > newword_thresh)
                save_bp(s, frame, s->t.w1_wid[i], h->out_score, h->out_history, 0);
        }
    }
}

/* prune_channels, :1130-1187 */
static void prune_channels(pso_ft_t *s, int frame, const int32_t *pp)
{
    s->n_cand = 0;
    s->dynamic_beam = s->beam;
    if (s->maxhmmpf != -1 && s->n_root_eval + s->n_nonroot_eval > s->maxhmmpf) {
        int32_t bins[256], bw = -s->beam / 256, nh = 0, b;
        int i;
        memset(bins, 0, sizeof bins);
        for (i = 0; i < s->R; ++i) {                          /* every root channel, active or not (:1146-1154) */
            b = (s->best_score - s->node[i].bestscore) / bw;
            if (b >= 256) b = 255;
            ++bins[b];
        }
        for (i = 0; i < s->n_acl[frame & 1]; ++i) {
            b = (s->best_score - s->node[s->acl[frame & 1][i]].bestscore) / bw;
            if (b >= 256) b = 255;
            ++bins[b];
        }
        for (i = 0; i < 256; ++i) { nh += bins[i]; if (nh > s->maxhmmpf) break; }
        s->dynamic_beam = -(i * bw);
    }
    if (s->par_mode == 2) prune_tree_list(s, frame, pp);
    else if (s->par_mode) prune_tree_parallel(s, frame, pp);
    else { prune_root(s, frame, pp); prune_nonroot(s, frame, pp); }
    last_phone_transition(s, frame);
    prune_word(s, frame);
}

/* bptable_maxwpf, :1193-1241 */
static void bptable_maxwpf(pso_ft_t *s, int frame)
{
    int bp, n = 0;
    int32_t bestscr = INT_MIN, worstscr;
    pso_bp_t *best = NULL, *worst;
    if (s->maxwpf == -1 || s->maxwpf == s->n_w) return;
    for (bp = s->bp_table_idx[frame]; bp < s->bpidx; ++bp) {
        pso_bp_t *e = &s->bp[bp];
        if (s->t.dict_filler[e->wid]) {
            if (e->score > bestscr) { bestscr = e->score; best = e; }
            e->valid = 0;
            ++n;
        }
    }
    if (best) { best->valid = 1; --n; }
    n = (s->bpidx - s->bp_table_idx[frame]) - n;
    for (; n > s->maxwpf; --n) {
        worstscr = INT_MAX; worst = NULL;
        for (bp = s->bp_table_idx[frame]; bp < s->bpidx; ++bp) {
            pso_bp_t *e = &s->bp[bp];
            if (e->valid && e->score < worstscr) { worstscr = e->score; worst = e; }
        }
        if (!worst) break;
        worst->valid = 0;
    }
}

/* word_transition, :1243-1427 */
static void word_transition(pso_ft_t *s, int frame, const int32_t *pp)
{
    const int nf = frame + 1;
    int i, k = 0, bp, rc, w;
    int32_t thresh, newscore;
    for (i = s->n_ci - 1; i >= 0; --i) s->bestrc[i].score = WORST;
    for (bp = s->bp_table_idx[frame]; bp < s->bpidx; ++bp) {
        const pso_bp_t *e = &s->bp[bp];
        s->word_lat_idx[e->wid] = NO_BP;
        if (e->wid == s->finishwid) continue;
        ++k;
        if (e->last2_phone == -1) {
            for (rc = 0; rc < s->n_ci; ++rc)
                if (e->score > s->bestrc[rc].score) {
                    s->bestrc[rc].score = e->score; s->bestrc[rc].path = bp; s->bestrc[rc].lc = e->last_phone;
                }
        }
        else {
            const int32_t *cimap = rs_cimap(s, e->last_phone, e->last2_phone);
            const int32_t *rcss = &s->bss[e->s_idx];
            for (rc = 0; rc < s->n_ci; ++rc)
                if (rcss[cimap[rc]] > s->bestrc[rc].score) {
                    s->bestrc[rc].score = rcss[cimap[rc]]; s->bestrc[rc].path = bp; s->bestrc[rc].lc = e->last_phone;
                }
        }
    }
    if (k == 0) return;
    thresh = s->best_score + s->dynamic_beam;
    for (i = 0; i < s->R; ++i) {                                 /* tree roots (:1306-1325) */
        pso_hmm_t *h = &s->node[i];
        const bestrc_t *b = &s->bestrc[s->t.node_ci[i]];
        newscore = b->score + s->nwpen + s->pip;
        if (newscore + pen(s, pp, s->t.node_ci[i]) > thresh && (h->frame < frame || newscore > h->score[0])) {
            h_enter(h, newscore, b->path, nf);
            h->senid[0] = (uint16_t)s->t.ldiph_lc[((size_t)s->t.node_ci[i] * s->n_ci + s->t.node_ci2[i]) * s->n_ci + b->lc];
        }
    }
    for (i = 0; i < s->n1lm; ++i) s->ltrans[s->t.w1_wid[i]].dscr = INT_MIN;      /* MAX_NEG_INT32 (:1331-1334) */
    for (bp = s->bp_table_idx[frame]; bp < s->bpidx; ++bp) {
        const pso_bp_t *e = &s->bp[bp];
        if (!e->valid) continue;
        for (i = 0; i < s->n1lm; ++i) {
            w = s->t.w1_wid[i];
            newscore = exit_score(s, e, s->t.dict_first[w]);
            if (newscore != WORST)
                newscore += lm_score(s, s->t.dict_basewid[w], e->real_wid, e->prev_real_wid);
            if (newscore > s->ltrans[w].dscr) { s->ltrans[w].dscr = newscore; s->ltrans[w].bp = bp; }
        }
    }
    for (i = 0; i < s->n1lm; ++i) {                              /* in-LM single-phone words (:1363-1388) */
        pso_hmm_t *h = &s->w1[i];
        w = s->t.w1_wid[i];
        if (w == s->startwid) continue;
        newscore = s->ltrans[w].dscr + s->pip;
        if (newscore + pen(s, pp, s->t.w1_ci[i]) > thresh) {
            const pso_bp_t *e = &s->bp[s->ltrans[w].bp];
            if (h->frame < frame || newscore > h->score[0]) {
                h_enter(h, newscore, s->ltrans[w].bp, nf);
                h->senid[0] = (uint16_t)s->t.ldiph_lc[((size_t)s->t.w1_ci[i] * s->n_ci + s->t.w1_ci2[i]) * s->n_ci
                                                       + s->t.dict_last[e->wid]];
            }
        }
    }
    {                                                            /* <sil> and the noise words (:1390-1426) */
        const bestrc_t *b = &s->bestrc[s->sil_ci];
        i = w1_index(s, s->silwid);
        newscore = b->score + s->silpen + s->pip;
        if (newscore + pen(s, pp, s->t.w1_ci[i]) > thresh && (s->w1[i].frame < frame || newscore > s->w1[i].score[0]))
            h_enter(&s->w1[i], newscore, b->path, nf);
        for (w = s->filler_start; w <= s->filler_end; ++w) {
            if (w == s->silwid || w == s->startwid) continue;
            i = w1_index(s, w);
            if (i < 0) continue;
            newscore = b->score + s->fillpen + s->pip;
            if (newscore + pen(s, pp, s->t.w1_ci[i]) > thresh && (s->w1[i].frame < frame || newscore > s->w1[i].score[0]))
                h_enter(&s->w1[i], newscore, b->path, nf);
        }
    }
}

/* deactivate_channels, :1429-1450 */
static void deactivate(pso_ft_t *s, int frame)
{
    int i;
    for (i = 0; i < s->R; ++i) if (s->node[i].frame == frame) h_clear(&s->node[i]);
    for (i = 0; i < s->n1; ++i) if (s->w1[i].frame == frame) h_clear(&s->w1[i]);
}

/* ngram_fwdtree_search, :1452-1495, with the frame's scores given as (listed ids, their scores,
 * the value of every other entry) and the penalties vector the phone loop holds at this point */
int pso_ft_step(pso_ft_t *s, int frame, const int32_t *ids, const int16_t *scr, int n, int16_t rest,
                const int32_t *penalties)
{
    int i;
    for (i = 0; i < s->n_sen; ++i) s->senscr[i] = rest;
    for (i = 0; i < n; ++i) s->senscr[ids[i]] = scr[i];
    mark_bptable(s, frame);
    if (s->best_score == WORST || s->best_score < WORST) return 0;
    if (s->best_score + 2 * s->beam < WORST) renormalize(s, frame, s->best_score);
    evaluate(s, frame);
    prune_channels(s, frame, penalties);
    bptable_maxwpf(s, frame);
    word_transition(s, frame, penalties);
    deactivate(s, frame);
    ++s->n_frame;
    return 1;
}

/* ngram_fwdtree_finish, :1497-1533 (the parts that touch the result) */
void pso_ft_finish(pso_ft_t *s, int n_frames)
{
    mark_bptable(s, n_frames);
}

int32_t pso_ft_best_score(const pso_ft_t *s) { return s->best_score; }
int32_t pso_ft_last_phone_best_score(const pso_ft_t *s) { return s->last_phone_best_score; }
int32_t pso_ft_bpidx(const pso_ft_t *s) { return s->bpidx; }
int32_t pso_ft_bss_head(const pso_ft_t *s) { return s->bss_head; }
const pso_bp_t *pso_ft_bp(const pso_ft_t *s) { return s->bp; }
const int32_t *pso_ft_bss(const pso_ft_t *s) { return s->bss; }
const int32_t *pso_ft_bp_table_idx(const pso_ft_t *s) { return s->bp_table_idx; }
