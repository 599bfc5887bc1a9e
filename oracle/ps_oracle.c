/* oracle/ps_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the PocketSphinx acoustic-scoring / Viterbi-step
 * arithmetic, written from the behaviour specified in SURVEY.md section 8(a)
 * and checked against the compiled reference (see ps_oracle.h for the parity
 * statement).  Compile with -ffp-contract=off: the Gaussian distance is a
 * strictly sequential fp32 sub/mul/mul/sub chain with no fused multiply-add
 * (SURVEY F5).
 *
 * Citations are to files under /root/reference/src.
 */
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <math.h>
#include "ps_oracle.h"

/* ================================================================== */
/* PTM scorer                                                          */
/* ================================================================== */

struct pso_ptm_s {
    int n_mgau, n_feat, n_density, n_sen, topn, ds_ratio, n_hist;
    int32_t *featlen;
    int32_t *featoff;        /* offset of stream f inside a frame vector */
    int64_t *cboff;          /* float offset of (mgau, feat) block in mean/var */
    int veclen;
    const float *mean, *var, *det;
    const uint8_t *mixw, *mixw_cb, *sen2cb, *logadd8;
    int logadd8_size;
    /* history ring (ptm_mgau.h:68-71, ptm_mgau.c:884-890) */
    pso_topn_t *hist;        /* [n_hist][n_mgau][n_feat][topn] */
    uint8_t *hist_active;    /* [n_hist][n_mgau] */
    int cur;                 /* current slot (s->f) */
    int frame_idx;           /* ps_mgau_t.frame_idx (acmod.h:113-116) */
};

static size_t
slot_len(const pso_ptm_t *s)
{
    return (size_t)s->n_mgau * s->n_feat * s->topn;
}

pso_ptm_t *
pso_ptm_new(int n_mgau, int n_feat, int n_density, const int32_t *featlen,
            int n_sen, int topn, int ds_ratio, int n_fast_hist,
            const float *mean, const float *var, const float *det,
            const uint8_t *mixw, const uint8_t *mixw_cb, const uint8_t *sen2cb,
            const uint8_t *logadd8, int logadd8_size)
{
    pso_ptm_t *s = calloc(1, sizeof(*s));
    int m, f;
    int64_t o = 0;
    s->n_mgau = n_mgau; s->n_feat = n_feat; s->n_density = n_density;
    s->n_sen = n_sen; s->topn = topn; s->ds_ratio = ds_ratio; s->n_hist = n_fast_hist;
    s->featlen = malloc(sizeof(int32_t) * n_feat);
    s->featoff = malloc(sizeof(int32_t) * n_feat);
    s->cboff = malloc(sizeof(int64_t) * n_mgau * n_feat);
    for (f = 0; f < n_feat; ++f) {
        s->featlen[f] = featlen[f];
        s->featoff[f] = s->veclen;
        s->veclen += featlen[f];
    }
    for (m = 0; m < n_mgau; ++m)
        for (f = 0; f < n_feat; ++f) {
            s->cboff[m * n_feat + f] = o;
            o += (int64_t)n_density * featlen[f];
        }
    s->mean = mean; s->var = var; s->det = det;
    s->mixw = mixw; s->mixw_cb = mixw_cb; s->sen2cb = sen2cb;
    s->logadd8 = logadd8; s->logadd8_size = logadd8_size;
    s->hist = malloc(sizeof(pso_topn_t) * slot_len(s) * n_fast_hist);
    s->hist_active = malloc((size_t)n_fast_hist * n_mgau);
    pso_ptm_reset_hist(s);
    return s;
}

void
pso_ptm_free(pso_ptm_t *s)
{
    if (!s) return;
    free(s->featlen); free(s->featoff); free(s->cboff);
    free(s->hist); free(s->hist_active);
    free(s);
}

/* ptm_mgau.c:777-802: every list starts as cw = 0..N-1, score = WORST_DIST,
 * every codebook flagged active. */
void
pso_ptm_reset_hist(pso_ptm_t *s)
{
    size_t i, n = slot_len(s) * s->n_hist;
    for (i = 0; i < n; ++i) {
        s->hist[i].cw = (int32_t)(i % s->topn);
        s->hist[i].score = PSO_WORST_DIST;
    }
    memset(s->hist_active, 1, (size_t)s->n_hist * s->n_mgau);
    s->cur = 0;
}

void pso_ptm_set_frame_idx(pso_ptm_t *s, int v) { s->frame_idx = v; }
int pso_ptm_get_frame_idx(const pso_ptm_t *s) { return s->frame_idx; }

const pso_topn_t *
pso_ptm_cur_topn(const pso_ptm_t *s)
{
    return s->hist + slot_len(s) * s->cur;
}

/* The Gaussian "distance": d = det - sum_j ((x_j - m_j)^2 * v_j), evaluated
 * left to right in fp32 (ptm_mgau.c:64-69 macros, :102-128, :182-209; the
 * reference's 1-then-4x grouping is the same left-to-right order). */
static float
gau_dist(float d, const float *x, const float *m, const float *v, int len)
{
    int j;
    for (j = 0; j < len; ++j) {
        float diff = x[j] - m[j];
        float sq = diff * diff;
        float c = sq * v[j];
        d = d - c;
    }
    return d;
}

/* float -> int32 as the reference does it (ptm_mgau.c:129-132, :220-223) */
static int32_t
dist_to_int(float d)
{
    if (d < (float)PSO_MAX_NEG_INT32)
        return PSO_MAX_NEG_INT32;
    return (int32_t)d;
}

/* eval_topn (ptm_mgau.c:87-136) with insertion_sort_topn (:71-85) */
static void
rescore_seed_list(const pso_ptm_t *s, pso_topn_t *tl, int cb, int f, const float *x)
{
    int len = s->featlen[f], i;
    int64_t base = s->cboff[cb * s->n_feat + f];
    const float *det = s->det + ((size_t)cb * s->n_feat + f) * s->n_density;
    for (i = 0; i < s->topn; ++i) {
        int cw = tl[i].cw, j;
        float d = gau_dist(det[cw], x, s->mean + base + (int64_t)cw * len,
                           s->var + base + (int64_t)cw * len, len);
        int32_t sc = dist_to_int(d);
        pso_topn_t e;
        e.cw = cw; e.score = sc;
        /* bubble entry i upward past strictly worse entries */
        for (j = i - 1; j >= 0 && sc > tl[j].score; --j)
            tl[j + 1] = tl[j];
        tl[j + 1] = e;
    }
}

/* eval_cb (ptm_mgau.c:151-226) with insertion_sort_cb (:140-149).  The
 * reference abandons a codeword as soon as the partial distance drops below
 * the threshold; since every subtracted term is >= 0 that is equivalent to
 * testing the completed distance, which is what is done here. */
static void
scan_codebook(const pso_ptm_t *s, pso_topn_t *tl, int cb, int f, const float *x)
{
    int len = s->featlen[f], N = s->topn, cw;
    int64_t base = s->cboff[cb * s->n_feat + f];
    const float *det = s->det + ((size_t)cb * s->n_feat + f) * s->n_density;
    for (cw = 0; cw < s->n_density; ++cw) {
        float thresh = (float)tl[N - 1].score;
        float d = gau_dist(det[cw], x, s->mean + base + (int64_t)cw * len,
                           s->var + base + (int64_t)cw * len, len);
        int i, p;
        int32_t sc;
        if (d < thresh)
            continue;
        for (i = 0; i < N; ++i)
            if (tl[i].cw == cw)
                break;
        if (i < N)
            continue;
        sc = dist_to_int(d);
        /* goes ahead of equal scores; the old worst entry falls off */
        for (p = N - 1; p > 0 && sc >= tl[p - 1].score; --p)
            tl[p] = tl[p - 1];
        tl[p].cw = cw;
        tl[p].score = sc;
    }
}

/* fast_logmath_add (tied_mgau_common.h:106-125); operands are negated logs.
 * The reference indexes the table without a bound check; the table is
 * "never smaller than 256 entries" and its tail is zero, so an index beyond
 * it is treated as zero here. */
static int
logadd8(const pso_ptm_t *s, int x, int y)
{
    int d, r;
    if (x > y) { d = x - y; r = y; }
    else       { d = y - x; r = x; }
    return r - (d < s->logadd8_size ? s->logadd8[d] : 0);
}

int
pso_ptm_frame_eval(pso_ptm_t *s, int16_t *senscr,
                   const uint8_t *senone_active, int32_t n_senone_active,
                   const float *feat, int32_t frame, int32_t compallsen,
                   pso_topn_t *raw_topn)
{
    int N = s->topn, evaluated = 0;
    int slot = frame % s->n_hist;               /* ptm_mgau.c:425-426 */
    pso_topn_t *cur = s->hist + slot_len(s) * slot;
    uint8_t *active = s->hist_active + (size_t)slot * s->n_mgau;
    int cb, f, k, i, lastsen, best;

    s->cur = slot;
    if (frame >= s->frame_idx) {                /* ptm_mgau.c:430 */
        int prev = (slot == 0) ? s->n_hist - 1 : slot - 1;
        evaluated = 1;
        memcpy(cur, s->hist + slot_len(s) * prev, sizeof(pso_topn_t) * slot_len(s));
        /* ptm_mgau_calc_cb_active (:297-321) */
        if (compallsen)
            memset(active, 1, s->n_mgau);
        else {
            memset(active, 0, s->n_mgau);
            for (lastsen = i = 0; i < n_senone_active; ++i) {
                int sen = senone_active[i] + lastsen;
                active[s->sen2cb[sen]] = 1;
                lastsen = sen;
            }
        }
        /* ptm_mgau_codebook_eval (:231-254): seeds of EVERY codebook are
         * re-scored, only active ones are scanned, and only on frames that
         * are multiples of the downsampling ratio. */
        for (cb = 0; cb < s->n_mgau; ++cb)
            for (f = 0; f < s->n_feat; ++f)
                rescore_seed_list(s, cur + ((size_t)cb * s->n_feat + f) * N, cb, f,
                                  feat + s->featoff[f]);
        if (frame % s->ds_ratio == 0) {
            for (cb = 0; cb < s->n_mgau; ++cb) {
                if (!active[cb]) continue;
                for (f = 0; f < s->n_feat; ++f)
                    scan_codebook(s, cur + ((size_t)cb * s->n_feat + f) * N, cb, f,
                                  feat + s->featoff[f]);
            }
        }
        if (raw_topn)
            memcpy(raw_topn, cur, sizeof(pso_topn_t) * slot_len(s));
        /* ptm_mgau_codebook_norm (:265-295) */
        for (f = 0; f < s->n_feat; ++f) {
            int32_t norm = PSO_WORST_SCORE;
            for (cb = 0; cb < s->n_mgau; ++cb) {
                int32_t top;
                if (!active[cb]) continue;
                top = cur[((size_t)cb * s->n_feat + f) * N].score >> PSO_SENSCR_SHIFT;
                if (norm < top) norm = top;
            }
            for (cb = 0; cb < s->n_mgau; ++cb) {
                pso_topn_t *tl = cur + ((size_t)cb * s->n_feat + f) * N;
                if (!active[cb]) continue;
                for (k = 0; k < N; ++k) {
                    int32_t v = tl[k].score >> PSO_SENSCR_SHIFT;
                    v = v - norm;   /* wraps like the reference's int arithmetic */
                    v = -v;
                    if (v > PSO_MAX_NEG_ASCR) v = PSO_MAX_NEG_ASCR;
                    tl[k].score = v;
                }
            }
        }
    }

    /* ptm_mgau_senone_eval (:326-403) */
    memset(senscr, 0, sizeof(int16_t) * s->n_sen);
    if (compallsen)
        n_senone_active = s->n_sen;
    best = 0x7fffffff;
    for (lastsen = i = 0; i < n_senone_active; ++i) {
        int sen = compallsen ? i : senone_active[i] + lastsen;
        int ascore = 0;
        lastsen = sen;
        cb = s->sen2cb[sen];
        if (!active[cb]) {
            /* persistent overwrite of the slot (:353-364) */
            for (f = 0; f < s->n_feat; ++f)
                for (k = 0; k < N; ++k)
                    cur[((size_t)cb * s->n_feat + f) * N + k].score = PSO_MAX_NEG_ASCR;
        }
        for (f = 0; f < s->n_feat; ++f) {
            const pso_topn_t *tl = cur + ((size_t)cb * s->n_feat + f) * N;
            int fden = 0;
            for (k = 0; k < N; ++k) {
                int w;
                if (s->mixw_cb) {
                    /* 4-bit clustered weights; the nibble is selected by the
                     * low bit of the packed byte itself, as in the reference
                     * (:375-379), not by the senone's parity. */
                    size_t row = (size_t)(s->n_sen + 1) / 2;
                    int dcw = s->mixw[((size_t)f * s->n_density + tl[k].cw) * row + sen / 2];
                    dcw = (dcw & 1) ? dcw >> 4 : dcw & 0x0f;
                    w = s->mixw_cb[dcw];
                }
                else
                    w = s->mixw[((size_t)f * s->n_density + tl[k].cw) * s->n_sen + sen];
                if (k == 0)
                    fden = w + tl[k].score;
                else
                    fden = logadd8(s, fden, w + tl[k].score);
            }
            ascore += fden;
        }
        if (ascore < best) best = ascore;
        senscr[sen] = (int16_t)ascore;
    }
    for (i = 0; i < s->n_sen; ++i)              /* :398-400, int16 store */
        senscr[i] = (int16_t)(senscr[i] - best);
    return evaluated;
}

void
pso_ptm_score_utt(pso_ptm_t *s, const float *feats, int T, int reset_hist,
                  int16_t *senscr, uint8_t *topn_cw, int32_t *topn_raw)
{
    size_t L = slot_len(s), i;
    pso_topn_t *raw = malloc(sizeof(pso_topn_t) * L);
    int16_t *row = malloc(sizeof(int16_t) * s->n_sen);
    int t;
    if (reset_hist) pso_ptm_reset_hist(s);
    s->frame_idx = 0;                            /* acmod_start_utt, acmod.c:419 */
    for (t = 0; t < T; ++t) {
        pso_ptm_frame_eval(s, senscr ? senscr + (size_t)t * s->n_sen : row, NULL, 0,
                           feats + (size_t)t * s->veclen, t, 1, raw);
        for (i = 0; i < L; ++i) {
            if (topn_cw) topn_cw[(size_t)t * L + i] = (uint8_t)raw[i].cw;
            if (topn_raw) topn_raw[(size_t)t * L + i] = raw[i].score;
        }
        s->frame_idx++;                          /* acmod_advance, acmod.c:874 */
    }
    free(raw); free(row);
}

/* ================================================================== */
/* semi-continuous scorer (s2_semi_mgau.c)                             */
/* ================================================================== */

struct pso_semi_s {
    int n_feat, n_density, n_sen, topn, ds_ratio, n_hist, veclen;
    int32_t *featlen, *featoff;
    int64_t *foff;                 /* float offset of stream f in mean/var */
    uint8_t *beam;
    const float *mean, *var, *det;
    const uint8_t *mixw, *mixw_cb, *logadd8;
    int logadd8_size;
    pso_topn_t *hist;              /* [n_hist][n_feat][topn]  (topn_hist, s2_semi_mgau.h:83) */
    uint8_t *hist_n;               /* [n_hist][n_feat]        (topn_hist_n, :84) */
    int cur, frame_idx;
};

pso_semi_t *
pso_semi_new(int n_feat, int n_density, const int32_t *featlen, int n_sen,
             int topn, int ds_ratio, int n_hist, const uint8_t *topn_beam,
             const float *mean, const float *var, const float *det,
             const uint8_t *mixw, const uint8_t *mixw_cb,
             const uint8_t *logadd8, int logadd8_size)
{
    pso_semi_t *s = calloc(1, sizeof(*s));
    int f; int64_t o = 0;
    s->n_feat = n_feat; s->n_density = n_density; s->n_sen = n_sen; s->topn = topn;
    s->ds_ratio = ds_ratio; s->n_hist = n_hist;
    s->featlen = malloc(sizeof(int32_t) * n_feat);
    s->featoff = malloc(sizeof(int32_t) * n_feat);
    s->foff = malloc(sizeof(int64_t) * n_feat);
    s->beam = malloc(n_feat);
    for (f = 0; f < n_feat; ++f) {
        s->featlen[f] = featlen[f]; s->featoff[f] = s->veclen; s->veclen += featlen[f];
        s->foff[f] = o; o += (int64_t)n_density * featlen[f];
        s->beam[f] = topn_beam ? topn_beam[f] : 0;
    }
    s->mean = mean; s->var = var; s->det = det; s->mixw = mixw; s->mixw_cb = mixw_cb;
    s->logadd8 = logadd8; s->logadd8_size = logadd8_size;
    s->hist = malloc(sizeof(pso_topn_t) * (size_t)n_hist * n_feat * topn);
    s->hist_n = malloc((size_t)n_hist * n_feat);
    pso_semi_reset_hist(s);
    return s;
}

void
pso_semi_free(pso_semi_t *s)
{
    if (!s) return;
    free(s->featlen); free(s->featoff); free(s->foff); free(s->beam);
    free(s->hist); free(s->hist_n); free(s);
}

/* s2_semi_mgau.c:1305-1322: topn_hist_n comes from ckd_calloc_2d (zeros),
 * every list is codeword k / WORST_DIST */
void
pso_semi_reset_hist(pso_semi_t *s)
{
    size_t i, n = (size_t)s->n_hist * s->n_feat * s->topn;
    for (i = 0; i < n; ++i) {
        s->hist[i].cw = (int32_t)(i % s->topn);
        s->hist[i].score = PSO_WORST_DIST;
    }
    memset(s->hist_n, 0, (size_t)s->n_hist * s->n_feat);
    s->cur = 0;
}

void pso_semi_set_frame_idx(pso_semi_t *s, int v) { s->frame_idx = v; }

const pso_topn_t *
pso_semi_cur_topn(const pso_semi_t *s, uint8_t *n_used)
{
    if (n_used) memcpy(n_used, s->hist_n + (size_t)s->cur * s->n_feat, s->n_feat);
    return s->hist + (size_t)s->cur * s->n_feat * s->topn;
}

static int
semi_logadd8(const pso_semi_t *s, int x, int y)
{
    int d, r;
    if (x > y) { d = x - y; r = y; } else { d = y - x; r = x; }
    return r - ((d >= 0 && d < s->logadd8_size) ? s->logadd8[d] : 0);
}

/* eval_topn (s2_semi_mgau.c:69-109): same as the PTM one */
static void
semi_rescore(const pso_semi_t *s, pso_topn_t *tl, int f, const float *x)
{
    int len = s->featlen[f], i;
    const float *mean = s->mean + s->foff[f], *var = s->var + s->foff[f];
    const float *det = s->det + (size_t)f * s->n_density;
    for (i = 0; i < s->topn; ++i) {
        int cw = tl[i].cw, j;
        float d = gau_dist(det[cw], x, mean + (int64_t)cw * len, var + (int64_t)cw * len, len);
        pso_topn_t e;
        e.cw = cw; e.score = dist_to_int(d);
        for (j = i - 1; j >= 0 && e.score > tl[j].score; --j)
            tl[j + 1] = tl[j];
        tl[j + 1] = e;
    }
}

/* eval_cb (s2_semi_mgau.c:111-170).  Unlike the PTM scan, the acceptance test
 * is NOT a function of the finished distance alone: the float test
 * `d >= worst->score` guards every dimension but is not applied to the
 * finished sum (the loop ends on j == ceplen), which is then compared as a
 * TRUNCATED int (`d_int < worst->score`).  Since every term is >= 0 the
 * partial sums fall monotonically, so "all guards passed" is "the partial sum
 * before the last dimension passed". */
static void
semi_scan(const pso_semi_t *s, pso_topn_t *tl, int f, const float *x)
{
    int len = s->featlen[f], N = s->topn, cw;
    const float *mean = s->mean + s->foff[f], *var = s->var + s->foff[f];
    const float *det = s->det + (size_t)f * s->n_density;
    for (cw = 0; cw < s->n_density; ++cw) {
        const float *m = mean + (int64_t)cw * len, *v = var + (int64_t)cw * len;
        float worst = (float)tl[N - 1].score;   /* int -> float for the comparison */
        float d = det[cw];
        int j, i, p;
        int32_t di;
        for (j = 0; j < len && d >= worst; ++j) {
            float diff = x[j] - m[j];
            float sq = diff * diff;
            float c = sq * v[j];
            d = d - c;
        }
        if (j < len)
            continue;
        di = dist_to_int(d);
        if (di < tl[N - 1].score)
            continue;
        for (i = 0; i < N; ++i)
            if (tl[i].cw == cw) break;
        if (i < N)
            continue;
        for (p = N - 1; p > 0 && di >= tl[p - 1].score; --p)
            tl[p] = tl[p - 1];
        tl[p].cw = cw; tl[p].score = di;
    }
}

/* mgau_norm (s2_semi_mgau.c:185-203): entries past the beam cut stay raw */
static int
semi_norm(const pso_semi_t *s, pso_topn_t *tl, int f)
{
    int32_t norm = tl[0].score >> PSO_SENSCR_SHIFT;
    int j;
    for (j = 0; j < s->topn; ++j) {
        tl[j].score = -((tl[j].score >> PSO_SENSCR_SHIFT) - norm);
        if (tl[j].score > PSO_MAX_NEG_ASCR) tl[j].score = PSO_MAX_NEG_ASCR;
        if (s->beam[f] && tl[j].score > s->beam[f]) break;
    }
    return j;
}

int
pso_semi_frame_eval(pso_semi_t *s, int16_t *senscr,
                    const uint8_t *senone_active, int32_t n_senone_active,
                    const float *feat, int32_t frame, int32_t compallsen)
{
    int slot = frame % s->n_hist, f, evaluated = 0;   /* :849-850 */
    pso_topn_t *cur = s->hist + (size_t)slot * s->n_feat * s->topn;
    uint8_t *cur_n = s->hist_n + (size_t)slot * s->n_feat;
    size_t row4 = (size_t)(s->n_sen + 1) / 2;

    memset(senscr, 0, sizeof(int16_t) * s->n_sen);     /* :846 */
    s->cur = slot;
    for (f = 0; f < s->n_feat; ++f) {
        pso_topn_t *tl = cur + (size_t)f * s->topn;
        int n, i, k, l;
        if (frame >= s->frame_idx) {                   /* :853-862 */
            int prev = (slot == 0) ? s->n_hist - 1 : slot - 1;
            evaluated = 1;
            memcpy(tl, s->hist + ((size_t)prev * s->n_feat + f) * s->topn, sizeof(pso_topn_t) * s->topn);
            semi_rescore(s, tl, f, feat + s->featoff[f]);
            if (frame % s->ds_ratio == 0)              /* mgau_dist :172-183 */
                semi_scan(s, tl, f, feat + s->featoff[f]);
            cur_n[f] = (uint8_t)semi_norm(s, tl, f);
        }
        n = cur_n[f];
        if (compallsen) {
            /* get_scores_{8b,4b}_feat_all (:431-444, :792-831): int arithmetic */
            int last = s->mixw_cb ? (s->n_sen & ~1) : s->n_sen;
            for (i = 0; i < last; ++i) {
                int tmp = 0;
                for (k = 0; k < n || k == 0; ++k) {
                    int w;
                    if (s->mixw_cb) {
                        int b = s->mixw[((size_t)f * s->n_density + tl[k].cw) * row4 + i / 2];
                        w = s->mixw_cb[(i & 1) ? (b >> 4) : (b & 0x0f)];
                    }
                    else
                        w = s->mixw[((size_t)f * s->n_density + tl[k].cw) * s->n_sen + i];
                    tmp = (k == 0) ? w + tl[k].score : semi_logadd8(s, tmp, w + tl[k].score);
                    if (n == 0) break;
                }
                senscr[i] = (int16_t)(senscr[i] + tmp);
            }
        }
        else {
            /* get_scores_8b_feat_N / _any (:205-394), get_scores_4b_feat_N / _any
             * (:446-790).  The unrolled 4-bit kernels (N = 1..6) precompute
             * w_den as uint8, i.e. mixw_cb + score wraps mod 256; _any and the
             * 8-bit kernels add in int.  N = 0 cannot happen after mgau_norm;
             * a never-evaluated slot (count 0) scores like the `_any` loop
             * with zero iterations: first codeword only. */
            int wrap = (s->mixw_cb != NULL) && n >= 1 && n <= 6;
            for (l = i = 0; i < n_senone_active; ++i) {
                int sen = senone_active[i] + l, tmp = 0;
                for (k = 0; k < n || k == 0; ++k) {
                    int w, y;
                    if (s->mixw_cb) {
                        int b = s->mixw[((size_t)f * s->n_density + tl[k].cw) * row4 + sen / 2];
                        w = s->mixw_cb[(sen & 1) ? (b >> 4) : (b & 0x0f)];
                    }
                    else
                        w = s->mixw[((size_t)f * s->n_density + tl[k].cw) * s->n_sen + sen];
                    y = w + tl[k].score;
                    if (wrap) y &= 0xff;
                    tmp = (k == 0) ? y : semi_logadd8(s, tmp, y);
                    if (n == 0) break;
                }
                senscr[sen] = (int16_t)(senscr[sen] + tmp);
                l = sen;
            }
        }
    }
    return evaluated;
}

/* ================================================================== */
/* multi-stream / continuous scorer (ms_mgau.c, ms_gauden.c, ms_senone.c) */
/* ================================================================== */

typedef struct { int32_t id; float dist; } pso_gdist_t;      /* gauden_dist_t, ms_gauden.h:70-75 */

struct pso_ms_s {
    int n_mgau, n_feat, n_density, n_sen, topn, aw, veclen;
    int32_t *featlen, *featoff;
    int64_t *cboff;
    const float *mean, *var, *det;
    const uint8_t *pdf;
    const uint32_t *sen2mgau;
    const void *logadd; int logadd_size, logadd_width; int32_t log_zero;
    pso_gdist_t *dist;             /* [n_mgau][n_feat][topn], persistent like msg->dist (ms_mgau.c:150-152) */
    uint8_t *active;
};

pso_ms_t *
pso_ms_new(int n_mgau, int n_feat, int n_density, const int32_t *featlen,
           int n_sen, int topn, int aw,
           const float *mean, const float *var, const float *det,
           const uint8_t *pdf, const uint32_t *sen2mgau,
           const void *logadd, int logadd_size, int logadd_width, int32_t log_zero)
{
    pso_ms_t *s = calloc(1, sizeof(*s));
    int m, f; int64_t o = 0;
    s->n_mgau = n_mgau; s->n_feat = n_feat; s->n_density = n_density; s->n_sen = n_sen;
    s->topn = topn; s->aw = aw;
    s->featlen = malloc(sizeof(int32_t) * n_feat);
    s->featoff = malloc(sizeof(int32_t) * n_feat);
    s->cboff = malloc(sizeof(int64_t) * n_mgau * n_feat);
    for (f = 0; f < n_feat; ++f) { s->featlen[f] = featlen[f]; s->featoff[f] = s->veclen; s->veclen += featlen[f]; }
    for (m = 0; m < n_mgau; ++m)
        for (f = 0; f < n_feat; ++f) { s->cboff[m * n_feat + f] = o; o += (int64_t)n_density * featlen[f]; }
    s->mean = mean; s->var = var; s->det = det; s->pdf = pdf; s->sen2mgau = sen2mgau;
    s->logadd = logadd; s->logadd_size = logadd_size; s->logadd_width = logadd_width; s->log_zero = log_zero;
    s->dist = calloc((size_t)n_mgau * n_feat * topn, sizeof(pso_gdist_t));   /* ckd_calloc_3d: ids start at 0 */
    s->active = calloc(n_mgau, 1);
    return s;
}

void
pso_ms_free(pso_ms_t *s)
{
    if (!s) return;
    free(s->featlen); free(s->featoff); free(s->cboff); free(s->dist); free(s->active); free(s);
}

/* compute_dist / compute_dist_all (ms_gauden.c:377-483).  The early exit on
 * `dval >= worst->dist` is result-neutral here (the finished value is tested
 * in float as well, :461), so the finished distance decides. */
static void
ms_topn(const pso_ms_t *s, int m, int f, const float *x, pso_gdist_t *out)
{
    int len = s->featlen[f], N = s->topn, d, i, j;
    int64_t base = s->cboff[m * s->n_feat + f];
    const float *det = s->det + ((size_t)m * s->n_feat + f) * s->n_density;
    if (N >= s->n_density) {                       /* compute_dist_all: unsorted, every density */
        for (d = 0; d < s->n_density; ++d) {
            out[d].dist = gau_dist(det[d], x, s->mean + base + (int64_t)d * len, s->var + base + (int64_t)d * len, len);
            out[d].id = d;
        }
        return;
    }
    for (i = 0; i < N; ++i)
        out[i].dist = (float)PSO_WORST_DIST;       /* ids keep their previous contents */
    for (d = 0; d < s->n_density; ++d) {
        float dval = gau_dist(det[d], x, s->mean + base + (int64_t)d * len, s->var + base + (int64_t)d * len, len);
        if (!(dval >= out[N - 1].dist))
            continue;
        for (i = 0; i < N && dval < out[i].dist; ++i) ;
        for (j = N - 1; j > i; --j) out[j] = out[j - 1];
        out[i].dist = dval; out[i].id = d;
    }
}

/* logmath_add (util/logmath.c:401-446) on the senone log-math object */
static int
ms_logadd(const pso_ms_t *s, int x, int y)
{
    int d, r;
    if (x <= s->log_zero) return y;
    if (y <= s->log_zero) return x;
    if (x > y) { d = x - y; r = x; } else { d = y - x; r = y; }
    if (d < 0 || d >= s->logadd_size) return r;
    switch (s->logadd_width) {
    case 1: return r + ((const uint8_t *)s->logadd)[d];
    case 2: return r + ((const uint16_t *)s->logadd)[d];
    case 4: return r + (int)((const uint32_t *)s->logadd)[d];
    }
    return r;
}

/* senone_eval (ms_senone.c:357-407) */
static int32_t
ms_senone(const pso_ms_t *s, int id, const pso_gdist_t *dist)
{
    int f, t, scr = 0;
    for (f = 0; f < s->n_feat; ++f) {
        const pso_gdist_t *fd = dist + (size_t)f * s->topn;
        const uint8_t *pdf = s->pdf + ((size_t)id * s->n_feat + f) * s->n_density;
        int fscr = 0;
        for (t = 0; t < s->topn; ++t) {
            int fden, fw;
            if (fd[t].dist < (float)PSO_MAX_NEG_INT32) fden = PSO_MAX_NEG_INT32 >> PSO_SENSCR_SHIFT;
            else fden = ((int32_t)fd[t].dist + ((1 << PSO_SENSCR_SHIFT) - 1)) >> PSO_SENSCR_SHIFT;
            fw = fden - pdf[fd[t].id];
            fscr = (t == 0) ? fw : ms_logadd(s, fscr, fw);
        }
        scr -= fscr;
    }
    scr /= s->aw;
    if (scr > 32767) scr = 32767;
    if (scr < -32768) scr = -32768;
    return scr;
}

int
pso_ms_frame_eval(pso_ms_t *s, int16_t *senscr,
                  const uint8_t *senone_active, int32_t n_senone_active,
                  const float *feat, int32_t compallsen)
{
    int m, f, i, n, best = 0x7fffffff;
    size_t L = (size_t)s->n_feat * s->topn;
    if (compallsen)
        memset(s->active, 1, s->n_mgau);
    else {
        memset(s->active, 0, s->n_mgau);
        for (n = 0, i = 0; i < n_senone_active; ++i) { n += senone_active[i]; s->active[s->sen2mgau[n]] = 1; }
    }
    for (m = 0; m < s->n_mgau; ++m)
        if (s->active[m])
            for (f = 0; f < s->n_feat; ++f)
                ms_topn(s, m, f, feat + s->featoff[f], s->dist + m * L + (size_t)f * s->topn);
    if (compallsen) n_senone_active = s->n_sen;
    for (n = 0, i = 0; i < n_senone_active; ++i) {
        int sen = compallsen ? i : (n += senone_active[i]);
        senscr[sen] = (int16_t)ms_senone(s, sen, s->dist + s->sen2mgau[sen] * L);
        if (best > senscr[sen]) best = senscr[sen];
    }
    for (n = 0, i = 0; i < n_senone_active; ++i) {
        int sen = compallsen ? i : (n += senone_active[i]);
        int bs = senscr[sen] - best;
        if (bs > 32767) bs = 32767;
        if (bs < -32768) bs = -32768;
        senscr[sen] = (int16_t)bs;
    }
    return 0;
}

/* ================================================================== */
/* dynamic features (feat/feat.c, feat/cmn.c)                          */
/* ================================================================== */
void
pso_dynfeat_1s_c_d_dd(const float *cep, int T, int cepsize, float *out)
{
    float *n = malloc(sizeof(float) * (size_t)(T + 6) * cepsize);   /* normalised + padded by 3 on each side */
    float *sum = calloc(cepsize, sizeof(float));
    int t, i, nframe = 0;
    if (T <= 0) { free(n); free(sum); return; }
    for (t = 0; t < T; ++t) {                       /* cmn.c:182-194 */
        if (cep[(size_t)t * cepsize] < 0) continue;
        for (i = 0; i < cepsize; ++i) sum[i] += cep[(size_t)t * cepsize + i];
        ++nframe;
    }
    for (i = 0; i < cepsize; ++i) sum[i] = sum[i] / nframe;           /* :196-198 */
    for (t = 0; t < T; ++t)
        for (i = 0; i < cepsize; ++i)
            n[(size_t)(t + 3) * cepsize + i] = cep[(size_t)t * cepsize + i] - sum[i];
    for (t = 0; t < 3; ++t) {                       /* feat.c:1295-1302 */
        memcpy(n + (size_t)t * cepsize, n + (size_t)3 * cepsize, sizeof(float) * cepsize);
        memcpy(n + (size_t)(T + 3 + t) * cepsize, n + (size_t)(T + 2) * cepsize, sizeof(float) * cepsize);
    }
    for (t = 0; t < T; ++t) {                       /* feat.c:579-622, FEAT_DCEP_WIN = 2 */
        const float *m = n + (size_t)(t + 3) * cepsize;
        float *f = out + (size_t)t * 3 * cepsize;
        for (i = 0; i < cepsize; ++i) {
            float d1, d2;
            f[i] = m[i];
            f[cepsize + i] = m[2 * cepsize + i] - m[-2 * cepsize + i];
            d1 = m[3 * cepsize + i] - m[-1 * cepsize + i];
            d2 = m[1 * cepsize + i] - m[-3 * cepsize + i];
            f[2 * cepsize + i] = d1 - d2;
        }
    }
    free(n); free(sum);
}

/* ================================================================== */
/* acmod_flags2list (acmod.c:1223-1275)                                */
/* ================================================================== */
int
pso_flags2list(const uint8_t *flags, int n_sen, uint8_t *deltas)
{
    int n = 0, last = 0, sen;
    for (sen = 0; sen < n_sen; ++sen) {
        int delta;
        if (!flags[sen]) continue;
        delta = sen - last;
        while (delta > 255) {       /* lossy bridge: the bridging entries get scored too */
            deltas[n++] = 255;
            delta -= 255;
        }
        deltas[n++] = (uint8_t)delta;
        last = sen;
    }
    return n;
}

/* ================================================================== */
/* HMM Viterbi step (hmm.c:222-805)                                    */
/* ================================================================== */

#define W PSO_WORST_SCORE

static inline int32_t clampw(int32_t v) { return v < W ? W : v; }

/* hmm_vit_eval_3st_lr (hmm.c:529-607) */
static int32_t
vit3(const pso_hmm_ctx_t *ctx, pso_hmm_t *h)
{
    const uint8_t *tp = ctx->tp + (size_t)h->tmatid * 3 * 4;
    const int16_t *ss = ctx->senscore;
#define TP(i, j) (-(int32_t)tp[(i) * 4 + (j)])
    int32_t s2 = h->score[2] - ss[h->senid[2]];
    int32_t s1 = h->score[1] - ss[h->senid[1]];
    int32_t s0 = h->score[0] - ss[h->senid[0]];
    int32_t best = W, t0, t1, t2 = INT_MIN, s3;

    if (s1 > W) {
        t1 = s2 + TP(2, 3);
        if (TP(1, 3) > -PSO_TMAT_WORST)
            t2 = s1 + TP(1, 3);
        if (t1 > t2) { s3 = t1; h->out_history = h->history[2]; }
        else         { s3 = t2; h->out_history = h->history[1]; }
        s3 = clampw(s3);
        h->out_score = s3;
        best = s3;
    }
    t0 = s2 + TP(2, 2);
    t1 = s1 + TP(1, 2);
    if (TP(0, 2) > -PSO_TMAT_WORST)
        t2 = s0 + TP(0, 2);         /* otherwise t2 keeps its previous value */
    if (t0 > t1) {
        if (t2 > t0) { s2 = t2; h->history[2] = h->history[0]; }
        else s2 = t0;
    }
    else {
        if (t2 > t1) { s2 = t2; h->history[2] = h->history[0]; }
        else { s2 = t1; h->history[2] = h->history[1]; }
    }
    s2 = clampw(s2);
    if (s2 > best) best = s2;
    h->score[2] = s2;

    t0 = s1 + TP(1, 1);
    t1 = s0 + TP(0, 1);
    if (t0 > t1) s1 = t0;
    else { s1 = t1; h->history[1] = h->history[0]; }
    s1 = clampw(s1);
    if (s1 > best) best = s1;
    h->score[1] = s1;

    s0 = clampw(s0 + TP(0, 0));
    if (s0 > best) best = s0;
    h->score[0] = s0;
    h->bestscore = best;
    return best;
#undef TP
}

/* hmm_vit_eval_3st_lr_mpx (hmm.c:609-707) */
static int32_t
vit3_mpx(const pso_hmm_ctx_t *ctx, pso_hmm_t *h)
{
    const uint8_t *tp = ctx->tp + (size_t)h->tmatid * 3 * 4;
    const int16_t *ss = ctx->senscore;
    uint16_t *ssid = h->senid;
#define TP(i, j) (-(int32_t)tp[(i) * 4 + (j)])
#define SEN(st) (-(int32_t)ss[ctx->sseq[(size_t)ssid[st] * 3 + (st)]])
    int32_t s3, s2, s1, s0, t0, t1, t2 = INT_MIN, best;

    if (ssid[2] == PSO_BAD_SSID) s2 = t1 = W;
    else { s2 = h->score[2] + SEN(2); t1 = s2 + TP(2, 3); }
    if (ssid[1] == PSO_BAD_SSID) s1 = t2 = W;
    else {
        s1 = h->score[1] + SEN(1);
        if (TP(1, 3) > -PSO_TMAT_WORST) t2 = s1 + TP(1, 3);
    }
    if (t1 > t2) { s3 = t1; h->out_history = h->history[2]; }
    else         { s3 = t2; h->out_history = h->history[1]; }
    s3 = clampw(s3);
    h->out_score = s3;
    best = s3;

    s0 = h->score[0] + SEN(0);
    t0 = t1 = W;
    if (s2 != W) t0 = s2 + TP(2, 2);
    if (s1 != W) t1 = s1 + TP(1, 2);
    if (TP(0, 2) > -PSO_TMAT_WORST) t2 = s0 + TP(0, 2);
    if (t0 > t1) {
        if (t2 > t0) { s2 = t2; h->history[2] = h->history[0]; ssid[2] = ssid[0]; }
        else s2 = t0;
    }
    else {
        if (t2 > t1) { s2 = t2; h->history[2] = h->history[0]; ssid[2] = ssid[0]; }
        else { s2 = t1; h->history[2] = h->history[1]; ssid[2] = ssid[1]; }
    }
    s2 = clampw(s2);
    if (s2 > best) best = s2;
    h->score[2] = s2;

    t0 = W;
    if (s1 != W) t0 = s1 + TP(1, 1);
    t1 = s0 + TP(0, 1);
    if (t0 > t1) s1 = t0;
    else { s1 = t1; h->history[1] = h->history[0]; ssid[1] = ssid[0]; }
    s1 = clampw(s1);
    if (s1 > best) best = s1;
    h->score[1] = s1;

    s0 = clampw(s0 + TP(0, 0));
    if (s0 > best) best = s0;
    h->score[0] = s0;
    h->bestscore = best;
    return best;
#undef TP
#undef SEN
}

/* three-way arg-max used for states 2..4 of the 5-state topologies: the
 * self loop t0, the neighbour t1 (from `nb`), the skip t2 (from `sk`). */
#define PICK3(dst, T0, T1, T2, nb, sk, MPX)                                  \
    do {                                                                     \
        if ((T0) > (T1)) {                                                   \
            if ((T2) > (T0)) { dst = (T2); h->history[(nb) + 1] = h->history[sk]; \
                               if (MPX) ssid[(nb) + 1] = ssid[sk]; }         \
            else dst = (T0);                                                 \
        } else {                                                             \
            if ((T2) > (T1)) { dst = (T2); h->history[(nb) + 1] = h->history[sk]; \
                               if (MPX) ssid[(nb) + 1] = ssid[sk]; }         \
            else { dst = (T1); h->history[(nb) + 1] = h->history[nb];        \
                   if (MPX) ssid[(nb) + 1] = ssid[nb]; }                     \
        }                                                                    \
    } while (0)

/* hmm_vit_eval_5st_lr (hmm.c:222-350) */
static int32_t
vit5(const pso_hmm_ctx_t *ctx, pso_hmm_t *h)
{
    const uint8_t *tp = ctx->tp + (size_t)h->tmatid * 5 * 6;
    const int16_t *ss = ctx->senscore;
    uint16_t *ssid = h->senid; /* unused by the non-mpx PICK3 expansion */
#define TP(i, j) (-(int32_t)tp[(i) * 6 + (j)])
#define SEN(st) (-(int32_t)ss[h->senid[st]])
    int32_t s5, s4, s3, s2, s1, s0, t0, t1, t2, best = W;

    s4 = h->score[4] + SEN(4);
    s3 = h->score[3] + SEN(3);
    if (s3 > W) {
        t1 = s4 + TP(4, 5);
        t2 = s3 + TP(3, 5);
        if (t1 > t2) { s5 = t1; h->out_history = h->history[4]; }
        else         { s5 = t2; h->out_history = h->history[3]; }
        s5 = clampw(s5);
        h->out_score = s5;
        best = s5;
    }
    s2 = h->score[2] + SEN(2);
    if (s2 > W) {
        t0 = s4 + TP(4, 4); t1 = s3 + TP(3, 4); t2 = s2 + TP(2, 4);
        PICK3(s4, t0, t1, t2, 3, 2, 0);
        s4 = clampw(s4);
        if (s4 > best) best = s4;
        h->score[4] = s4;
    }
    s1 = h->score[1] + SEN(1);
    if (s1 > W) {
        t0 = s3 + TP(3, 3); t1 = s2 + TP(2, 3); t2 = s1 + TP(1, 3);
        PICK3(s3, t0, t1, t2, 2, 1, 0);
        s3 = clampw(s3);
        if (s3 > best) best = s3;
        h->score[3] = s3;
    }
    s0 = h->score[0] + SEN(0);
    t0 = s2 + TP(2, 2); t1 = s1 + TP(1, 2); t2 = s0 + TP(0, 2);
    PICK3(s2, t0, t1, t2, 1, 0, 0);
    s2 = clampw(s2);
    if (s2 > best) best = s2;
    h->score[2] = s2;

    t0 = s1 + TP(1, 1); t1 = s0 + TP(0, 1);
    if (t0 > t1) s1 = t0;
    else { s1 = t1; h->history[1] = h->history[0]; }
    s1 = clampw(s1);
    if (s1 > best) best = s1;
    h->score[1] = s1;

    s0 = clampw(s0 + TP(0, 0));
    if (s0 > best) best = s0;
    h->score[0] = s0;
    h->bestscore = best;
    (void)ssid;
    return best;
#undef TP
#undef SEN
}

/* hmm_vit_eval_5st_lr_mpx (hmm.c:355-525) */
static int32_t
vit5_mpx(const pso_hmm_ctx_t *ctx, pso_hmm_t *h)
{
    const uint8_t *tp = ctx->tp + (size_t)h->tmatid * 5 * 6;
    const int16_t *ss = ctx->senscore;
    uint16_t *ssid = h->senid;
#define TP(i, j) (-(int32_t)tp[(i) * 6 + (j)])
#define SEN(st) (-(int32_t)ss[ctx->sseq[(size_t)ssid[st] * 5 + (st)]])
    int32_t s5, s4, s3, s2, s1, s0, t0, t1, t2, best;

    if (ssid[4] == PSO_BAD_SSID) s4 = t1 = W;
    else { s4 = h->score[4] + SEN(4); t1 = s4 + TP(4, 5); }
    if (ssid[3] == PSO_BAD_SSID) s3 = t2 = W;
    else { s3 = h->score[3] + SEN(3); t2 = s3 + TP(3, 5); }
    if (t1 > t2) { s5 = t1; h->out_history = h->history[4]; }
    else         { s5 = t2; h->out_history = h->history[3]; }
    s5 = clampw(s5);
    h->out_score = s5;
    best = s5;

    if (ssid[2] == PSO_BAD_SSID) s2 = t2 = W;
    else { s2 = h->score[2] + SEN(2); t2 = s2 + TP(2, 4); }
    t0 = t1 = W;
    if (s4 != W) t0 = s4 + TP(4, 4);
    if (s3 != W) t1 = s3 + TP(3, 4);
    PICK3(s4, t0, t1, t2, 3, 2, 1);
    s4 = clampw(s4);
    if (s4 > best) best = s4;
    h->score[4] = s4;

    if (ssid[1] == PSO_BAD_SSID) s1 = t2 = W;
    else { s1 = h->score[1] + SEN(1); t2 = s1 + TP(1, 3); }
    t0 = t1 = W;
    if (s3 != W) t0 = s3 + TP(3, 3);
    if (s2 != W) t1 = s2 + TP(2, 3);
    PICK3(s3, t0, t1, t2, 2, 1, 1);
    s3 = clampw(s3);
    if (s3 > best) best = s3;
    h->score[3] = s3;

    s0 = h->score[0] + SEN(0);
    t0 = t1 = W;
    if (s2 != W) t0 = s2 + TP(2, 2);
    if (s1 != W) t1 = s1 + TP(1, 2);
    t2 = s0 + TP(0, 2);
    PICK3(s2, t0, t1, t2, 1, 0, 1);
    s2 = clampw(s2);
    if (s2 > best) best = s2;
    h->score[2] = s2;

    t0 = W;
    if (s1 != W) t0 = s1 + TP(1, 1);
    t1 = s0 + TP(0, 1);
    if (t0 > t1) s1 = t0;
    else { s1 = t1; h->history[1] = h->history[0]; ssid[1] = ssid[0]; }
    s1 = clampw(s1);
    if (s1 > best) best = s1;
    h->score[1] = s1;

    s0 = clampw(s0 + TP(0, 0));
    if (s0 > best) best = s0;
    h->score[0] = s0;
    h->bestscore = best;
    return best;
#undef TP
#undef SEN
}

/* hmm_vit_eval_anytopo (hmm.c:710-784): any number of emitting states up to
 * HMM_MAX_NSTATE (5), any upper-triangular transition matrix.  The reference
 * reaches it for 1, 2 and 4 emitting states.  Unlike the hard-wired forms it
 * (a) clamps only the INCOMING state + senone sums of states 1.. (state 0's is
 * left as is), (b) never clamps the new scores, (c) leaves a state's history
 * (and, multiplexed, its ssid) alone when its self loop wins -- `bestfrom`
 * stays -1 -- or when nothing reaches it. */
static int32_t
vit_any(const pso_hmm_ctx_t *ctx, pso_hmm_t *h)
{
    const int ne = h->n_emit_state;
    const uint8_t *tp = ctx->tp + (size_t)h->tmatid * ne * (ne + 1);
    int32_t st[5], scr, newscr, bestscr;
    int to, from, bestfrom;
#define TP(i, j) (-(int32_t)tp[(i) * (ne + 1) + (j)])
    /* hmm_senscr (hmm.h:207-209): WORST_SCORE for a state without a senone */
    for (from = 0; from < ne; ++from) {
        int32_t sen;
        if (h->mpx) {
            const uint16_t ssid = h->senid[from];
            sen = ssid == PSO_BAD_SSID ? W : -(int32_t)ctx->senscore[ctx->sseq[(size_t)ssid * ne + from]];
        }
        else
            sen = h->senid[from] == PSO_BAD_SSID ? W : -(int32_t)ctx->senscore[h->senid[from]];
        st[from] = h->score[from] + sen;
        if (from > 0 && st[from] < W) st[from] = W;
    }
    /* the non-emitting final state: no self transition */
    to = ne;
    scr = W;
    bestfrom = -1;
    for (from = to - 1; from >= 0; --from)
        if (TP(from, to) > -PSO_TMAT_WORST && (newscr = st[from] + TP(from, to)) > scr) { scr = newscr; bestfrom = from; }
    h->out_score = scr;
    if (bestfrom >= 0) h->out_history = h->history[bestfrom];
    bestscr = scr;
    for (to = ne - 1; to >= 0; --to) {
        scr = TP(to, to) > -PSO_TMAT_WORST ? st[to] + TP(to, to) : W;
        bestfrom = -1;
        for (from = to - 1; from >= 0; --from)
            if (TP(from, to) > -PSO_TMAT_WORST && (newscr = st[from] + TP(from, to)) > scr) { scr = newscr; bestfrom = from; }
        h->score[to] = scr;
        if (bestfrom >= 0) {
            h->history[to] = h->history[bestfrom];
            if (h->mpx) h->senid[to] = h->senid[bestfrom];
        }
        if (bestscr < scr) bestscr = scr;
    }
    h->bestscore = bestscr;
    return bestscr;
#undef TP
}

/* hmm_vit_eval (hmm.c:786-805) */
int32_t
pso_hmm_vit_eval(const pso_hmm_ctx_t *ctx, pso_hmm_t *h)
{
    if (h->n_emit_state == 3)
        return h->mpx ? vit3_mpx(ctx, h) : vit3(ctx, h);
    if (h->n_emit_state == 5)
        return h->mpx ? vit5_mpx(ctx, h) : vit5(ctx, h);
    return vit_any(ctx, h);
}

/* the host's libm log over an array: what the reference's fe_mel_cep calls per mel channel (fe_sigproc.c:1215-1228).
 * (numpy's log is its own vector routine, not libm: tests that pin the device's log call this.) */
void
pso_libm_log(const double *x, int64_t n, double *out)
{
    int64_t i;
    for (i = 0; i < n; ++i) out[i] = log(x[i]);
}

/* ====================================================================== */
/* MFCC front end                                                         */
/* ====================================================================== */

int
pso_fe_n_frames(const pso_fe_t *fe, long n)
{
    /* fe_interface.c:398-403 (full frames), :526-541 (fe_end_utt: one more frame
     * from the overflow samples; after the full frames the overflow buffer holds
     * frame_size - frame_shift + (left-over < frame_shift) > 0 samples, :452-466;
     * with fewer than frame_size samples it holds all of them, :375-381) */
    if (n <= 0) return 0;
    if (n < fe->frame_size) return 1;
    return 1 + (int)((n - fe->frame_size) / fe->frame_shift) + 1;
}

/* in-place real FFT of fe_fft_real (fe_sigproc.c:1052-1149): bit reversal, one
 * stage of 2-point butterflies, then stages k = 1..m-1 over blocks of 2^(k+1)
 * points.  Output: x[j] real part, x[n-j] imaginary part. */
static void
fe_rfft(const pso_fe_t *fe, double *x)
{
    int n = fe->fft_size, m = fe->fft_order, i, j, k;
    for (i = 0; i < n; ++i) {                       /* :1062-1075 as a permutation */
        int r = 0, b;
        for (b = 0; b < m; ++b) r |= ((i >> b) & 1) << (m - 1 - b);
        if (i < r) { double t = x[i]; x[i] = x[r]; x[r] = t; }
    }
    for (i = 0; i < n; i += 2) {                    /* :1081-1085 */
        double a = x[i], b = x[i + 1];
        x[i] = a + b;
        x[i + 1] = a - b;
    }
    for (k = 1; k < m; ++k) {                       /* :1088-1144 */
        int half = 1 << k, quarter = half >> 1, blk = half << 1, base;
        for (base = 0; base < n; base += blk) {
            double a = x[base], b = x[base + half];
            x[base] = a + b;
            x[base + half] = a - b;
            x[base + half + quarter] = -x[base + half + quarter];
            for (j = 1; j < quarter; ++j) {
                int i1 = base + j, i2 = base + half - j, i3 = base + half + j, i4 = base + blk - j;
                double cc = fe->ccc[j << (m - k - 1)], ss = fe->sss[j << (m - k - 1)];
                double p = x[i3] * cc, q = x[i4] * ss, r = x[i3] * ss, s = x[i4] * cc;
                double t1 = p + q, t2 = r - s;
                double x1 = x[i1], x2 = x[i2];
                x[i4] = x2 - t2;
                x[i3] = -x2 - t2;
                x[i2] = x1 - t1;
                x[i1] = x1 + t1;
            }
        }
    }
}

/* fe_remove_noise (fe_noise.c:268-364), floating-point branches. */
static void
fe_denoise(const pso_fe_t *fe, double *mf, double *st, int32_t *undefined)
{
    const double l_pow = 0.7, c_pow = 1 - 0.7, l_a = 0.995, c_a = 1 - 0.995, l_b = 0.5, c_b = 1 - 0.5;
    const double l_t = 0.85, mu_t = 0.2, max_gain = 20, inv_max_gain = 1.0 / 20;   /* :59-68, :203-213 */
    int nf = fe->n_filt, i, j;
    double *power = st, *noise = st + nf, *floor_ = st + 2 * nf, *peak = st + 3 * nf;
    double signal[256], gain[256];
    if (*undefined) {                               /* :282-298 */
        for (i = 0; i < nf; ++i) {
            power[i] = mf[i];
            noise[i] = mf[i] / max_gain;
            floor_[i] = mf[i] / max_gain;
            peak[i] = 0.0;
        }
        *undefined = 0;
    }
    for (i = 0; i < nf; ++i) {
        double in;
        power[i] = l_pow * power[i] + c_pow * mf[i];                      /* :301-309 */
        if (power[i] >= noise[i]) noise[i] = l_a * noise[i] + c_a * power[i];      /* fe_lower_envelope :109-127 */
        else                      noise[i] = l_b * noise[i] + c_b * power[i];
        signal[i] = power[i] - noise[i];                                   /* :315-323 */
        if (signal[i] < 1.0) signal[i] = 1.0;
        if (signal[i] >= floor_[i]) floor_[i] = l_a * floor_[i] + c_a * signal[i]; /* :327 */
        else                        floor_[i] = l_b * floor_[i] + c_b * signal[i];
        in = signal[i];                                                    /* fe_temp_masking :131-152 */
        peak[i] *= l_t;
        if (signal[i] < l_t * peak[i]) signal[i] = peak[i] * mu_t;
        if (in > peak[i]) peak[i] = in;
        if (signal[i] < floor_[i]) signal[i] = floor_[i];                  /* :331-334 */
        if (signal[i] < max_gain * power[i]) gain[i] = signal[i] / power[i];       /* :337-345 */
        else                                 gain[i] = max_gain;
        if (gain[i] < inv_max_gain) gain[i] = inv_max_gain;
    }
    for (i = 0; i < nf; ++i) {                      /* fe_weight_smooth :155-184, window 4 */
        int l1 = i - 4 > 0 ? i - 4 : 0, l2 = i + 4 < nf - 1 ? i + 4 : nf - 1;
        double c = 0;
        for (j = l1; j <= l2; ++j) c += gain[j];
        mf[i] = mf[i] * (c / (l2 - l1 + 1));
    }
}

/* fe_mel_cep + fe_lifter (fe_sigproc.c:1217-1342): mfcc_t (float32) accumulators,
 * float64 log spectrum. */
static void
fe_cepstrum(const pso_fe_t *fe, double *mf, float *cep)
{
    int nf = fe->n_filt, nc = fe->num_cepstra, i, j;
    for (i = 0; i < nf; ++i)
        mf[i] = log(mf[i] + 1e-4);                  /* LOG_FLOOR :1215, :1228 */
    if (fe->log_spec == 1) {                        /* RAW_LOG_SPEC :1233-1237 */
        for (i = 0; i < fe->out_dim; ++i) cep[i] = (float)mf[i];
        return;
    }
    if (fe->log_spec == 2 || fe->transform == 1 || fe->transform == 2) {
        int htk = (fe->log_spec != 2 && fe->transform == 2);
        float c[256];
        float *o = fe->log_spec == 2 ? c : cep;
        o[0] = (float)mf[0];                        /* fe_dct2 :1288-1310 */
        for (j = 1; j < nf; ++j) o[0] = (float)(o[0] + mf[j]);
        o[0] = o[0] * (htk ? fe->sqrt_inv_2n : fe->sqrt_inv_n);
        for (i = 1; i < nc; ++i) {
            o[i] = 0;
            for (j = 0; j < nf; ++j)
                o[i] = (float)(o[i] + mf[j] * fe->mel_cosine[i * nf + j]);
            o[i] = o[i] * fe->sqrt_inv_2n;
        }
        if (fe->log_spec == 2) {                    /* SMOOTH_LOG_SPEC :1240-1248: fe_dct3 :1326-1338 */
            for (i = 0; i < nf; ++i) {
                mf[i] = c[0] * 0.707106781186548;   /* SQRT_HALF is a double constant, fe_internal.h:106 */
                for (j = 1; j < nc; ++j)
                    mf[i] += c[j] * fe->mel_cosine[j * nf + i];
                mf[i] = mf[i] * fe->sqrt_inv_2n;
            }
            for (i = 0; i < fe->out_dim; ++i) cep[i] = (float)mf[i];
            return;                                 /* fe_lifter is still applied by the caller */
        }
    }
    else {                                          /* fe_spec2cep :1257-1285 */
        cep[0] = (float)(mf[0] / 2);
        for (j = 1; j < nf; ++j) cep[0] = (float)(cep[0] + mf[j]);
        cep[0] = (float)(cep[0] / (double)nf);
        for (i = 1; i < nc; ++i) {
            cep[i] = 0;
            for (j = 0; j < nf; ++j) {
                int beta = j == 0 ? 1 : 2;
                cep[i] = (float)(cep[i] + mf[j] * fe->mel_cosine[i * nf + j] * beta);
            }
            cep[i] = (float)(cep[i] / ((double)nf * 2));
        }
    }
}

int
pso_fe_process_utt(const pso_fe_t *fe, const int16_t *pcm, long n, float *cep,
                   double *noise, int32_t *undefined)
{
    int nfr = pso_fe_n_frames(fe, n), t, i, j;
    int fs = fe->frame_size, N = fe->fft_size;
    double *x = malloc(sizeof(double) * N), *spec = malloc(sizeof(double) * (N / 2 + 1));
    double mf[256];
    for (t = 0; t < nfr; ++t) {
        long start = (long)t * fe->frame_shift;
        int len = (int)(n - start < fs ? n - start : fs);
        float *out = cep + (size_t)t * fe->out_dim;
        /* fe_spch_to_frame (fe_sigproc.c:839-860): pre-emphasis with the sample
         * before the frame (fe_pre_emphasis_int16 :745-749; prior = 0 at the
         * utterance start, fe_interface.c:325), zero padding, window */
        if (fe->alpha != 0.0f) {
            double prior = start > 0 ? (double)pcm[start - 1] : 0.0;
            x[0] = (double)pcm[start] - prior * fe->alpha;
            for (i = 1; i < len; ++i)
                x[i] = (double)pcm[start + i] - (double)pcm[start + i - 1] * fe->alpha;
        }
        else
            for (i = 0; i < len; ++i) x[i] = (double)pcm[start + i];
        for (i = len; i < N; ++i) x[i] = 0;
        if (fe->remove_dc) {                        /* fe_hamming_window :802-815 */
            double mean = 0;
            for (i = 0; i < fs; ++i) mean += x[i];
            mean /= fs;
            for (i = 0; i < fs; ++i) x[i] -= mean;
        }
        for (i = 0; i < fs / 2; ++i) {              /* :826-829 */
            x[i] = x[i] * fe->hamming[i];
            x[fs - 1 - i] = x[fs - 1 - i] * fe->hamming[i];
        }
        fe_rfft(fe, x);
        spec[0] = x[0] * x[0];                      /* fe_spec_magnitude :1171-1191 */
        for (j = 1; j <= N / 2; ++j)
            spec[j] = x[j] * x[j] + x[N - j] * x[N - j];
        for (i = 0; i < fe->n_filt; ++i) {          /* fe_mel_spec :1194-1213 */
            const float *co = fe->filt_coeffs + fe->filt_start[i];
            const double *sp = spec + fe->spec_start[i];
            double a = 0;
            for (j = 0; j < fe->filt_width[i]; ++j) a += sp[j] * co[j];
            mf[i] = a;
        }
        if (fe->remove_noise)
            fe_denoise(fe, mf, noise, undefined);
        fe_cepstrum(fe, mf, out);
        if (fe->has_lifter)                         /* fe_lifter :1313-1323 */
            for (i = 0; i < fe->num_cepstra; ++i) out[i] = out[i] * fe->lifter[i];
    }
    free(x); free(spec);
    return nfr;
}
