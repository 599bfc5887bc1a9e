/* oracle/dropin_decode.c -- TEST INFRASTRUCTURE (checker), not product.
 *
 * Drop-in proof for the GMM boundary: decode the same PCM with two decoders
 * built from the UNMODIFIED reference library (oracle/_ref/libpocketsphinx.so)
 *   A: untouched (CPU ptm_mgau_t)
 *   B: after psgpu_mgau_attach() (integration/psgpu_mgau_shim.c -> libpsgpu.so)
 * through the public ps_start_utt / ps_process_raw / ps_end_utt path
 * (ps_decode_raw does exactly this, pocketsphinx.c:1030-1070) and compare
 *   - every frame_eval call's full int16 senone score vector (FNV-1a hash per
 *     call; both passes, phone-loop and search calls alike)
 *   - hypothesis string and path score
 *   - segmentation (word, start frame, end frame, ascr, lscr, lback)
 * Prints one JSON line; exit status 0 iff everything is identical.
 *
 * Built twice by oracle/Makefile:
 *   dropin_decode       against the plain reference library + the GMM shim
 *   dropin_decode_full  (-DPSGPU_SEARCH_HOOKS) against libpocketsphinx_psgpu.so,
 *                       the reference with its three hmm_vit_eval loops routed
 *                       through integration/psgpu_search_hooks.h; decoder B
 *                       then also runs every Viterbi step on the device
 *
 * usage: dropin_decode MODELDIR LM DICT RAW NREP [key val ...]
 *   pseudo keys: mllr_after FILE | psgpu_mgau yes|no | psgpu_search yes|no | align_text "WORDS"
 *                | psgpu_fe yes (decoder B: PCM -> cepstra on the device, integration/psgpu_fe_shim.c)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "bin_mdef.h"
#include "psgpu_mgau_shim.h"
#include "psgpu_fe_shim.h"
#include "psgpu_phone_loop_shim.h"
#include "ngram_search.h"
#include "psgpu_device_decode.h"
#include "phone_loop_search.h"
#ifdef PSGPU_SEARCH_HOOKS
#include "psgpu_search_hooks.h"
#endif

typedef struct rec_s {
    ps_mgaufuncs_t funcs;          /* wrapper vtable */
    ps_mgaufuncs_t *orig;
    uint64_t *hash;
    int32 *frame;
    int n, cap, n_sen;
} rec_t;

static rec_t *g_rec;               /* the recorder of the decoder being run */

static uint64_t
fnv1a(const void *p, size_t n)
{
    const unsigned char *b = p;
    uint64_t h = 0xcbf29ce484222325ull;
    size_t i;
    for (i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

static int
rec_frame_eval(ps_mgau_t *mg, int16 *senscr, uint8 *act, int32 nact,
               mfcc_t **feat, int32 frame, int32 compallsen)
{
    rec_t *r = g_rec;
    int rc = r->orig->frame_eval(mg, senscr, act, nact, feat, frame, compallsen);
    if (r->n == r->cap) {
        r->cap = r->cap ? r->cap * 2 : 4096;
        r->hash = realloc(r->hash, sizeof(uint64_t) * r->cap);
        r->frame = realloc(r->frame, sizeof(int32) * r->cap);
    }
    if (r->n == 0 && getenv("PSGPU_DEBUG_DUMP")) {
        int i;
        fprintf(stderr, "call0 frame %d nact %d compall %d:", frame, nact, compallsen);
        for (i = 0; i < 24 && i < r->n_sen; ++i) fprintf(stderr, " %d", senscr[i]);
        fprintf(stderr, "\n");
    }
    r->hash[r->n] = fnv1a(senscr, sizeof(int16) * r->n_sen);
    r->frame[r->n] = frame;
    ++r->n;
    return rc;
}

static void
rec_install(rec_t *r, ps_decoder_t *ps)
{
    memset(r, 0, sizeof *r);
    r->orig = ps->acmod->mgau->vt;
    r->funcs = *r->orig;
    r->funcs.frame_eval = rec_frame_eval;
    r->n_sen = bin_mdef_n_sen(ps->acmod->mdef);
    ps->acmod->mgau->vt = &r->funcs;
}

static ps_decoder_t *
make_decoder(const char *modeldir, const char *lm, const char *dict, int argc, char **argv)
{
    ps_config_t *config = ps_config_init(NULL);
    ps_decoder_t *ps;
    int i;
    ps_config_set_str(config, "hmm", modeldir);
    if (strcmp(lm, "-")) ps_config_set_str(config, "lm", lm);
    if (strcmp(dict, "-")) ps_config_set_str(config, "dict", dict);
    ps_config_set_str(config, "loglevel", "ERROR");
    for (i = 0; i + 1 < argc; i += 2) {
        const char *k = argv[i];
        if (k[0] == '-') ++k;
        if (!strcmp(k, "mllr_after") || !strcmp(k, "psgpu_mgau") || !strcmp(k, "psgpu_search")
            || !strcmp(k, "align_text") || !strcmp(k, "psgpu_fe") || !strcmp(k, "psgpu_phone_loop") || !strcmp(k, "psgpu_device_search") || !strcmp(k, "psgpu_device_vtable") || !strcmp(k, "chunked"))
            continue;                                  /* handled by main() */
        if (ps_config_set_str(config, k, argv[i + 1]) == NULL) {
            fprintf(stderr, "bad config %s=%s\n", k, argv[i + 1]); exit(2);
        }
    }
    ps = ps_init(config);
    if (!ps) { fprintf(stderr, "ps_init failed\n"); exit(2); }
    return ps;
}

typedef struct result_s {
    char hyp[4096];
    int32 score;
    char seg[65536];               /* "word sf ef ascr lscr lback\n" ... */
    int n_frames;
    char partial[65536];           /* "chunked N": "hyp|score;" after every ps_process_raw call of the utterance */
    int n_partial;
} result_t;

static int g_chunk;                /* "chunked N": the utterance arrives N samples at a time, ps_get_hyp after each piece */

static double
now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* Sphinx cepstra file, as pocketsphinx_batch reads its -cepdir inputs
 * (programs/pocketsphinx_batch.c:195-250) */
static float32 **
read_mfc(const char *path, int ceplen, int *out_nfr)
{
    FILE *fp = fopen(path, "rb");
    long flen; int32 nmfc; int nfr, i, swap = 0;
    float32 **mfcs;
    if (!fp) { perror(path); exit(2); }
    fseek(fp, 0, SEEK_END); flen = ftell(fp); fseek(fp, 0, SEEK_SET);
    if (fread(&nmfc, 4, 1, fp) != 1) { perror(path); exit(2); }
    if (nmfc != flen / 4 - 1) {
        nmfc = (int32)__builtin_bswap32((uint32_t)nmfc); swap = 1;
        if (nmfc != flen / 4 - 1) { fprintf(stderr, "%s: not an MFCC file\n", path); exit(2); }
    }
    nfr = nmfc / ceplen;
    mfcs = (float32 **)ckd_calloc_2d(nfr, ceplen, sizeof(float32));
    if (fread(mfcs[0], 4, (size_t)nfr * ceplen, fp) != (size_t)nfr * ceplen) { perror(path); exit(2); }
    fclose(fp);
    if (swap)
        for (i = 0; i < nfr * ceplen; ++i) {
            uint32_t *u = (uint32_t *)&mfcs[0][i];
            *u = __builtin_bswap32(*u);
        }
    *out_nfr = nfr;
    return mfcs;
}

/* "psgpu_phone_loop yes": decoder B's phone-loop steps come from one device launch per utterance
 * (integration/psgpu_phone_loop_shim.c).  Both decoders' steps are wrapped to record a hash of
 * pls->penalties after every step: the only thing the phone loop hands to the n-gram search. */
enum { MAX_PL = 1 << 18 };
static ps_searchfuncs_t g_plvt[2], *g_plorig[2];
static uint64_t *g_plhash[2];
static int g_pln[2];
static int
pl_record(int which, ps_search_t *s, int frame_idx)
{
    phone_loop_search_t *pls = (phone_loop_search_t *)s;
    int rv = g_plorig[which]->step(s, frame_idx), i;
    uint64_t h = 1469598103934665603ULL;
    const unsigned char *b = (const unsigned char *)pls->penalties;
    for (i = 0; i < (int)(pls->n_phones * sizeof(int32)); ++i) { h ^= b[i]; h *= 1099511628211ULL; }
    if (g_pln[which] < MAX_PL) g_plhash[which][g_pln[which]++] = h;
    return rv;
}
static int pl_record_a(ps_search_t *s, int f) { return pl_record(0, s, f); }
static int pl_record_b(ps_search_t *s, int f) { return pl_record(1, s, f); }

static psgpu_device_decode_t *g_dd;   /* "psgpu_device_search yes": decoder B's whole first pass runs on the device */
static long g_dd_frames;
static long g_live_frames, g_live_steps, g_live_restarts, g_live_utt_frames;   /* psgpu_device_search_live_stats, summed over the utterances read out in mid-utterance */
static psgpu_fe_shim_t *g_fe;      /* "psgpu_fe yes": decoder B's cepstra come from the device */

static void
decode(ps_decoder_t *ps, const int16 *pcm, size_t n, float32 **mfcs, int nfr, result_t *res, int dev_fe)
{
    const char *hyp;
    ps_seg_t *seg;
    size_t o = 0;
    if (dev_fe == 2) {
        int nf = psgpu_device_decode_utt(g_dd, pcm, n);
        if (nf < 0) { fprintf(stderr, "device decode failed\n"); exit(3); }
        g_dd_frames += nf;
        goto results;
    }
    ps_start_utt(ps);
    res->partial[0] = 0; res->n_partial = 0;
    if (g_chunk > 0 && pcm && !dev_fe) {
        /* live decoding (ps_process_raw without full_utt, pocketsphinx.c:1220-1257): results in mid-utterance */
        size_t at = 0, po = 0;
        while (at < n) {
            size_t k = n - at < (size_t)g_chunk ? n - at : (size_t)g_chunk;
            int32 sc = 0;
            const char *h;
            ps_process_raw(ps, pcm + at, k, FALSE, FALSE);
            at += k;
            h = ps_get_hyp(ps, &sc);
            if (po + 512 < sizeof res->partial) po += snprintf(res->partial + po, sizeof res->partial - po, "%s|%d;", h ? h : "", sc);
            ++res->n_partial;
        }
    }
    else if (mfcs && g_chunk > 0 && !dev_fe) {
        /* ... from cepstra: g_chunk FRAMES a piece (ps_process_cep without full_utt), a read-out after each */
        int at = 0; size_t po = 0;
        while (at < nfr) {
            int k = nfr - at < g_chunk ? nfr - at : g_chunk;
            int32 sc = 0;
            const char *h;
            ps_process_cep(ps, mfcs + at, k, FALSE, FALSE);
            at += k;
            h = ps_get_hyp(ps, &sc);
            if (po + 512 < sizeof res->partial) po += snprintf(res->partial + po, sizeof res->partial - po, "%s|%d;", h ? h : "", sc);
            ++res->n_partial;
        }
    }
    else if (mfcs)
        ps_process_cep(ps, mfcs, nfr, FALSE, TRUE);
    else if (dev_fe) {
        if (psgpu_process_raw_full(ps, g_fe, pcm, n) < 0) { fprintf(stderr, "device front end failed\n"); exit(3); }
    }
    else
        ps_process_raw(ps, pcm, n, FALSE, TRUE);
    ps_end_utt(ps);
results:
    hyp = ps_get_hyp(ps, &res->score);
    snprintf(res->hyp, sizeof res->hyp, "%s", hyp ? hyp : "");
    res->seg[0] = 0;
    for (seg = ps_seg_iter(ps); seg; seg = ps_seg_next(seg)) {
        int sf, ef; int32 ascr, lscr, lback;
        ps_seg_frames(seg, &sf, &ef);
        ps_seg_prob(seg, &ascr, &lscr, &lback);
        o += snprintf(res->seg + o, sizeof res->seg - o, "%s %d %d %d %d %d\n",
                      ps_seg_word(seg), sf, ef, ascr, lscr, lback);
    }
    res->n_frames = ps_get_n_frames(ps);
}

int
main(int argc, char **argv)
{
    ps_decoder_t *cpu, *gpu;
    rec_t rc_cpu, rc_gpu;
    result_t *ra, *rb;
    FILE *fp; long sz; int16 *pcm; size_t n;
    enum { MAX_IN = 512 };
    char *in_id[MAX_IN], *in_path[MAX_IN];
    int n_in = 0, n_res, u, total_frames = 0;
    int nrep, r, i, ok = 1, bad_calls = 0, first_bad = -1, hyp_equal = 1, seg_equal = 1, partial_equal = 1, n_partial = 0;
    double t_cpu = 0, t_gpu = 0, t0;
    int use_mgau = 1;
#ifdef PSGPU_SEARCH_HOOKS
    int use_search = 1;
#else
    int use_search = 0;
#endif
    long hmm_batches = 0, hmm_evals = 0, pl_dev = 0, pl_host = 0;
    int use_pl = 0, pl_bad = 0, use_dd = 0, use_dv = 0;

    if (argc < 6) {
        fprintf(stderr, "usage: dropin_decode MODELDIR LM|- DICT|- RAW NREP [key val ...]\n");
        return 2;
    }
    nrep = atoi(argv[5]);
    /* inputs: one raw/.mfc file, or "@CTLFILE:DIR" = every id of CTLFILE as DIR/id.mfc
     * (pocketsphinx_batch -ctl/-cepdir, test/regression/test-tidigits-simple.sh) */
    if (argv[4][0] == '@') {
        char *spec = strdup(argv[4] + 1), *dir = strchr(spec, ':'), *ext, line[512];
        if (!dir) { fprintf(stderr, "bad ctl spec\n"); return 2; }
        *dir++ = 0;
        ext = strchr(dir, ':');                       /* optional third field: file extension */
        if (ext) *ext++ = 0; else ext = "mfc";
        fp = fopen(spec, "r");
        if (!fp) { perror(spec); return 2; }
        while (fgets(line, sizeof line, fp)) {
            line[strcspn(line, "\r\n")] = 0;
            if (!line[0]) continue;
            in_id[n_in] = strdup(line);
            in_path[n_in] = malloc(strlen(dir) + strlen(line) + strlen(ext) + 8);
            sprintf(in_path[n_in], "%s/%s.%s", dir, line, ext);
            if (++n_in == MAX_IN) break;
        }
        fclose(fp);
    }
    else {
        in_id[0] = "utt"; in_path[0] = argv[4]; n_in = 1;
    }
    err_set_loglevel(ERR_ERROR);

    cpu = make_decoder(argv[1], argv[2], argv[3], argc - 6, argv + 6);
    gpu = make_decoder(argv[1], argv[2], argv[3], argc - 6, argv + 6);
    for (i = 6; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "align_text")) {        /* forced alignment: state_align_search */
            if (ps_set_align_text(cpu, argv[i + 1]) < 0 || ps_set_align_text(gpu, argv[i + 1]) < 0) {
                fprintf(stderr, "ps_set_align_text failed\n"); return 2;
            }
        }
        if (!strcmp(argv[i], "psgpu_mgau")) use_mgau = !strcmp(argv[i + 1], "yes");
        if (!strcmp(argv[i], "psgpu_search")) use_search = !strcmp(argv[i + 1], "yes");
        if (!strcmp(argv[i], "psgpu_phone_loop")) use_pl = !strcmp(argv[i + 1], "yes");
        if (!strcmp(argv[i], "psgpu_device_search")) use_dd = !strcmp(argv[i + 1], "yes");
        /* "psgpu_device_vtable yes": decoder B's n-gram search gets the device ps_searchfuncs_t (psgpu_device_search_attach)
         * and is then driven by the UNMODIFIED public calls below (ps_start_utt / ps_process_raw / ps_end_utt) */
        if (!strcmp(argv[i], "psgpu_device_vtable")) use_dv = use_dd = !strcmp(argv[i + 1], "yes");
        if (!strcmp(argv[i], "chunked")) g_chunk = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "psgpu_fe") && !strcmp(argv[i + 1], "yes")) {
            g_fe = psgpu_fe_wrap(gpu->acmod->fe);
            if (!g_fe) { fprintf(stderr, "psgpu_fe_wrap failed\n"); return 3; }
        }
    }
    if (use_mgau && psgpu_mgau_attach(gpu) < 0) {
        fprintf(stderr, "psgpu_mgau_attach failed\n");
        return 3;
    }
#ifdef PSGPU_SEARCH_HOOKS
    if (use_search && psgpu_search_attach(gpu) < 0) {
        fprintf(stderr, "psgpu_search_attach failed\n");
        return 3;
    }
#else
    if (use_search) { fprintf(stderr, "built without the search hooks\n"); return 2; }
#endif
    if (use_dd) {
        g_dd = psgpu_device_decode_attach(gpu);
        if (!g_dd) { fprintf(stderr, "psgpu_device_decode_attach failed\n"); return 3; }
        if (use_dv && psgpu_device_search_attach(g_dd) < 0) { fprintf(stderr, "psgpu_device_search_attach failed\n"); return 3; }
    }
    if (use_pl) {
        if (psgpu_phone_loop_attach(gpu) < 0) { fprintf(stderr, "psgpu_phone_loop_attach failed\n"); return 3; }
        if (cpu->phone_loop && gpu->phone_loop) {
            g_plhash[0] = malloc(sizeof(uint64_t) * MAX_PL); g_plhash[1] = malloc(sizeof(uint64_t) * MAX_PL);
            g_plorig[0] = cpu->phone_loop->vt; g_plvt[0] = *g_plorig[0]; g_plvt[0].step = pl_record_a;
            cpu->phone_loop->vt = &g_plvt[0];
            g_plorig[1] = gpu->phone_loop->vt; g_plvt[1] = *g_plorig[1]; g_plvt[1].step = pl_record_b;
            gpu->phone_loop->vt = &g_plvt[1];
        }
    }
    /* "mllr_after FILE": apply an MLLR transform AFTER attaching, so that the
     * shim's vt->transform (acmod_update_mllr, acmod.c:329) is what runs */
    for (i = 6; i + 1 < argc; i += 2)
        if (!strcmp(argv[i], "mllr_after") || !strcmp(argv[i], "-mllr_after")) {
            ps_mllr_t *ma = ps_mllr_read(argv[i + 1]), *mb = ps_mllr_read(argv[i + 1]);
            if (!ma || !mb) { fprintf(stderr, "cannot read MLLR %s\n", argv[i + 1]); return 2; }
            ps_update_mllr(cpu, ma);
            ps_update_mllr(gpu, mb);
        }
    if (!use_pl && !use_dd) {       /* (the recorder replaces vt, which hides the psgpu scorer from the other shims) */
        rec_install(&rc_cpu, cpu);
        rec_install(&rc_gpu, gpu);
    }
    n_res = nrep * n_in;
    ra = calloc(n_res, sizeof *ra);
    rb = calloc(n_res, sizeof *rb);
    for (r = 0; r < nrep; ++r) {
        for (u = 0; u < n_in; ++u) {
            const char *path = in_path[u];
            size_t len = strlen(path);
            int k = r * n_in + u, nfr = 0;
            float32 **mfcs = NULL;
            pcm = NULL; n = 0;
            if (len > 4 && !strcmp(path + len - 4, ".mfc"))
                mfcs = read_mfc(path, ps_config_int(ps_get_config(cpu), "ceplen"), &nfr);
            else {
                fp = fopen(path, "rb");
                if (!fp) { perror(path); return 2; }
                fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
                pcm = malloc(sz);
                if (fread(pcm, 1, sz, fp) != (size_t)sz) { perror("read"); return 2; }
                fclose(fp);
                n = sz / 2;
            }
            g_rec = &rc_cpu; t0 = now_s(); decode(cpu, pcm, n, mfcs, nfr, &ra[k], 0); t_cpu += now_s() - t0;
            if (mfcs) {     /* ps_process_cep normalises its input in place (CMN): reload for B */
                ckd_free_2d(mfcs);
                mfcs = read_mfc(path, ps_config_int(ps_get_config(cpu), "ceplen"), &nfr);
            }
            g_rec = &rc_gpu; t0 = now_s(); decode(gpu, pcm, n, mfcs, nfr, &rb[k], (g_dd && !use_dv) ? 2 : (g_fe != NULL)); t_gpu += now_s() - t0;
            if (use_dv) {
                long fs = 0, st = 0, rs = 0;
                g_dd_frames += ((ngram_search_t *)gpu->search)->n_frame;
                psgpu_device_search_live_stats(g_dd, &fs, &st, &rs);
                if (st > 0) { g_live_frames += fs; g_live_steps += st; g_live_restarts += rs; g_live_utt_frames += ((ngram_search_t *)gpu->search)->n_frame; }
            }
            if (strcmp(ra[k].hyp, rb[k].hyp) || ra[k].score != rb[k].score) hyp_equal = 0;
            if (strcmp(ra[k].seg, rb[k].seg)) seg_equal = 0;
            if (strcmp(ra[k].partial, rb[k].partial) || ra[k].n_partial != rb[k].n_partial) partial_equal = 0;
            n_partial += ra[k].n_partial;
            total_frames += ra[k].n_frames;
            if (mfcs) ckd_free_2d(mfcs);
            free(pcm);
        }
    }
    if (use_pl) {
        /* the device phone loop makes no frame_eval calls of its own: the call sequences differ by
         * construction; what must agree is every step's penalties vector and the decode results */
        if (g_pln[0] != g_pln[1]) pl_bad = -1;
        else for (i = 0; i < g_pln[0]; ++i) pl_bad += g_plhash[0][i] != g_plhash[1][i];
        if (pl_bad) ok = 0;
        cpu->phone_loop->vt = g_plorig[0]; gpu->phone_loop->vt = g_plorig[1];
        psgpu_phone_loop_stats(gpu, &pl_dev, &pl_host);
        psgpu_phone_loop_detach(gpu);
    }
    else if (use_dd) { /* the device search makes no frame_eval calls at all: results only */ }
    else if (rc_cpu.n != rc_gpu.n) { ok = 0; bad_calls = -1; }
    else
        for (i = 0; i < rc_cpu.n; ++i)
            if (rc_cpu.hash[i] != rc_gpu.hash[i] || rc_cpu.frame[i] != rc_gpu.frame[i]) {
                if (first_bad < 0) first_bad = i;
                ++bad_calls;
            }
    if (bad_calls || !hyp_equal || !seg_equal || !partial_equal) ok = 0;
#ifdef PSGPU_SEARCH_HOOKS
    psgpu_search_stats(gpu, &hmm_batches, &hmm_evals);
    psgpu_search_detach(gpu);
#endif
    /* detach recorders before the decoders free their scorers */
    if (!use_pl && !use_dd) {
        cpu->acmod->mgau->vt = rc_cpu.orig;
        gpu->acmod->mgau->vt = rc_gpu.orig;
    }
    {
        int n_seg = 0; const char *p;
        for (p = ra[0].seg; *p; ++p) n_seg += (*p == '\n');
        printf("{\"ok\": %s, \"nrep\": %d, \"n_frames\": %d, \"calls_cpu\": %d, \"calls_gpu\": %d, "
               "\"device_calls\": %d, \"mismatching_calls\": %d, \"first_bad_call\": %d, "
               "\"hyp_equal\": %s, \"seg_equal\": %s, \"hyp_cpu\": \"%s\", \"hyp_gpu\": \"%s\", "
               "\"score_cpu\": %d, \"score_gpu\": %d, \"n_seg\": %d, "
               "\"decode_s_cpu\": %.4f, \"decode_s_gpu\": %.4f, \"mgau\": \"%s\", "
               "\"cache_served\": %ld, \"search_hooks\": %s, \"hmm_batches\": %ld, \"hmm_evals\": %ld, \"n_utts\": %d, "
               "\"total_frames\": %d, \"device_fe\": %s, \"pl_steps\": %d, \"pl_mismatch\": %d, \"pl_device_steps\": %ld, "
               "\"pl_host_steps\": %ld, \"device_search_frames\": %ld, \"partial_results\": %d, \"partial_equal\": %s, \"live_frames_searched\": %ld, \"live_utt_frames\": %ld, \"live_steps\": %ld, \"live_restarts\": %ld, "
               "\"last_partial_cpu\": \"%.200s\", \"utts\": [",
               ok ? "true" : "false", nrep, ra[0].n_frames, rc_cpu.n, rc_gpu.n,
               use_mgau ? (int)psgpu_mgau_n_calls(gpu->acmod->mgau) : 0, bad_calls, first_bad,
               hyp_equal ? "true" : "false", seg_equal ? "true" : "false",
               ra[n_res - 1].hyp, rb[n_res - 1].hyp, ra[n_res - 1].score, rb[n_res - 1].score,
               n_seg, t_cpu, t_gpu, gpu->acmod->mgau->vt->name,
               use_mgau ? psgpu_mgau_n_cache_served(gpu->acmod->mgau) : 0L,
               use_search ? "true" : "false", hmm_batches, hmm_evals, n_res, total_frames, g_fe ? "true" : "false",
               g_pln[0], pl_bad, pl_dev, pl_host, g_dd_frames, n_partial, partial_equal ? "true" : "false", g_live_frames, g_live_utt_frames, g_live_steps, g_live_restarts,
               ra[n_res - 1].n_partial ? (strrchr(ra[n_res - 1].partial, ';') ? ra[n_res - 1].partial + (strlen(ra[n_res - 1].partial) > 180 ? strlen(ra[n_res - 1].partial) - 180 : 0) : "") : "");
        for (u = 0; u < n_res; ++u)
            printf("%s{\"id\": \"%s\", \"hyp\": \"%s\", \"score\": %d}", u ? ", " : "",
                   in_id[u % n_in], rb[u].hyp, rb[u].score);
        printf("]}\n");
    }
    psgpu_fe_shim_free(g_fe);
    psgpu_device_decode_detach(g_dd);
    ps_free(cpu);
    ps_free(gpu);
    return ok ? 0 : 1;
}
