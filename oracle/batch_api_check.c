/* oracle/batch_api_check.c -- TEST INFRASTRUCTURE for integration/psgpu_decode_batch.c.
 *
 * usage: batch_api_check MODELDIR LM|- DICT|- N_WORKERS FLAGS RAW [RAW ...] [-- key val ...]
 *
 * 1. decodes every RAW on a FRESH unmodified CPU decoder (ps_init, first utterance)
 *    -- the definition psgpu_decode_batch's results are held to;
 * 2. psgpu_decode_batch() on the whole list (B = number of files, in the given order
 *    and reversed), then on each file alone (B = 1);
 * 3. compares hypothesis, path score, frame count and every segment
 *    (word, sf, ef, ascr, lscr, lback).
 * Prints one JSON line; exit code 0 iff everything is identical. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "psgpu_decode_batch.h"

static double
now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static ps_config_t *
make_config(char **argv, int nx, char **extra)
{
    ps_config_t *config = ps_config_init(NULL);
    int i;
    ps_config_set_str(config, "hmm", argv[1]);
    if (strcmp(argv[2], "-")) ps_config_set_str(config, "lm", argv[2]);
    if (strcmp(argv[3], "-")) ps_config_set_str(config, "dict", argv[3]);
    ps_config_set_str(config, "loglevel", "ERROR");
    for (i = 0; i + 1 < nx; i += 2)
        if (ps_config_set_str(config, extra[i], extra[i + 1]) == NULL) {
            fprintf(stderr, "bad config %s\n", extra[i]); exit(2);
        }
    return config;
}

static int
same(const psgpu_batch_result_t *a, const psgpu_batch_result_t *b)
{
    int i;
    if (strcmp(a->hyp, b->hyp) || a->score != b->score || a->n_frames != b->n_frames || a->n_seg != b->n_seg)
        return 0;
    for (i = 0; i < a->n_seg; ++i)
        if (strcmp(a->seg[i].word, b->seg[i].word) || a->seg[i].sf != b->seg[i].sf || a->seg[i].ef != b->seg[i].ef
            || a->seg[i].ascr != b->seg[i].ascr || a->seg[i].lscr != b->seg[i].lscr || a->seg[i].lback != b->seg[i].lback)
            return 0;
    return 1;
}

static psgpu_multi_t *g_multi;
static int
run_batch(psgpu_batch_t *b, const int16 *const pcm[], const size_t n[], int B, psgpu_batch_result_t out[])
{
    return g_multi ? psgpu_decode_batch_multi(g_multi, pcm, n, B, out) : psgpu_decode_batch(b, pcm, n, B, out);
}

int
main(int argc, char **argv)
{
    enum { MAXU = 2048 };
    int16 *pcm[MAXU]; size_t n[MAXU];
    const int16 *cp[MAXU], *rp[MAXU]; size_t rn[MAXU];
    psgpu_batch_result_t ref[MAXU], got[MAXU], rev[MAXU], one[MAXU];
    int B = 0, i, nx = 0, n_workers, bad_batch = 0, bad_rev = 0, bad_one = 0, frames = 0;
    unsigned flags;
    char **extra = NULL;
    psgpu_batch_t *cpu, *dev;
    double t0, t_ref, t_batch;

    if (argc < 7) {
        fprintf(stderr, "usage: batch_api_check MODELDIR LM|- DICT|- N_WORKERS FLAGS RAW [RAW ...] [-- key val ...]\n");
        return 2;
    }
    n_workers = atoi(argv[4]); flags = (unsigned)atoi(argv[5]);
    for (i = 6; i < argc; ++i) {
        FILE *fp; long sz;
        if (!strcmp(argv[i], "--")) { extra = argv + i + 1; nx = argc - i - 1; break; }
        if (B == MAXU) break;
        fp = fopen(argv[i], "rb");
        if (!fp) { perror(argv[i]); return 2; }
        fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
        pcm[B] = malloc(sz);
        if (fread(pcm[B], 1, sz, fp) != (size_t)sz) { perror("read"); return 2; }
        fclose(fp);
        n[B] = sz / 2; cp[B] = pcm[B];
        ++B;
    }
    err_set_loglevel(ERR_ERROR);

    /* 1. the definition: every utterance on its own fresh CPU decoder
     *    (BATCH_CHECK_TIMING_ONLY=1 skips it and every comparison: throughput runs on big batches) */
    t0 = now_s();
    for (i = 0; i < B && !getenv("BATCH_CHECK_TIMING_ONLY"); ++i) {
        cpu = psgpu_batch_init(make_config(argv, nx, extra), 1, PSGPU_BATCH_CPU_ONLY);
        if (!cpu || psgpu_decode_batch(cpu, &cp[i], &n[i], 1, &ref[i]) < 0) { fprintf(stderr, "cpu decode failed\n"); return 2; }
        psgpu_batch_free(cpu);
        frames += ref[i].n_frames;
    }
    t_ref = now_s() - t0;

    /* 2. the batch call (BATCH_CHECK_DEVICES="0,0": through the multi-device dispatcher, one batch object per entry) */
    if (getenv("BATCH_CHECK_DEVICES")) {
        int devs[16], nd = 0;
        char *spec = strdup(getenv("BATCH_CHECK_DEVICES")), *tok;
        for (tok = strtok(spec, ","); tok && nd < 16; tok = strtok(NULL, ",")) devs[nd++] = atoi(tok);
        g_multi = psgpu_multi_init(make_config(argv, nx, extra), devs, nd, n_workers, flags);
        if (!g_multi) { fprintf(stderr, "psgpu_multi_init failed\n"); return 3; }
        dev = NULL;
    }
    else
    dev = psgpu_batch_init(make_config(argv, nx, extra), n_workers, flags);
    if (!dev && !g_multi) { fprintf(stderr, "psgpu_batch_init failed\n"); return 3; }
    if (run_batch(dev, cp, n, B, got) < 0) { fprintf(stderr, "batch decode failed\n"); return 3; }   /* warm */
    for (i = 0; i < B; ++i) psgpu_batch_result_clear(&got[i]);
    t0 = now_s();
    if (run_batch(dev, cp, n, B, got) < 0) { fprintf(stderr, "batch decode failed\n"); return 3; }
    t_batch = now_s() - t0;
    if (getenv("BATCH_CHECK_TIMING_ONLY")) {
        for (i = 0; i < B; ++i) frames += got[i].n_frames;
        printf("{\"timing_only\": true, \"B\": %d, \"workers\": %d, \"flags\": %u, \"frames\": %d, \"batch_s\": %.4f, "
               "\"frames_per_s\": %.1f, \"hyp0\": \"%s\"}\n", B, n_workers, flags, frames, t_batch, frames / t_batch, got[0].hyp);
        if (g_multi) psgpu_multi_free(g_multi); else psgpu_batch_free(dev);
        return 0;
    }
    for (i = 0; i < B; ++i) { rp[i] = cp[B - 1 - i]; rn[i] = n[B - 1 - i]; }
    if (run_batch(dev, rp, rn, B, rev) < 0) { fprintf(stderr, "batch decode failed\n"); return 3; }
    for (i = 0; i < B; ++i)
        if (run_batch(dev, &cp[i], &n[i], 1, &one[i]) < 0) { fprintf(stderr, "B=1 decode failed\n"); return 3; }
    for (i = 0; i < B; ++i) {
        if (!same(&ref[i], &got[i]) && getenv("BATCH_CHECK_VERBOSE")) {
            int k;
            fprintf(stderr, "utt %d: ref \"%s\" %d (%d fr, %d seg) vs \"%s\" %d (%d fr, %d seg)\n", i, ref[i].hyp,
                    ref[i].score, ref[i].n_frames, ref[i].n_seg, got[i].hyp, got[i].score, got[i].n_frames, got[i].n_seg);
            for (k = 0; k < ref[i].n_seg && k < got[i].n_seg; ++k)
                fprintf(stderr, "  %s %d %d %d %d %d | %s %d %d %d %d %d\n", ref[i].seg[k].word, ref[i].seg[k].sf,
                        ref[i].seg[k].ef, ref[i].seg[k].ascr, ref[i].seg[k].lscr, ref[i].seg[k].lback, got[i].seg[k].word,
                        got[i].seg[k].sf, got[i].seg[k].ef, got[i].seg[k].ascr, got[i].seg[k].lscr, got[i].seg[k].lback);
        }
        bad_batch += !same(&ref[i], &got[i]);
        bad_rev += !same(&ref[i], &rev[B - 1 - i]);
        bad_one += !same(&ref[i], &one[i]);
    }
    printf("{\"ok\": %s, \"B\": %d, \"workers\": %d, \"flags\": %u, \"frames\": %d, \"mismatch_batch\": %d, "
           "\"mismatch_reversed\": %d, \"mismatch_single\": %d, \"cpu_s\": %.4f, \"batch_s\": %.4f, \"hyps\": [",
           (bad_batch || bad_rev || bad_one) ? "false" : "true", B, n_workers, flags, frames, bad_batch, bad_rev,
           bad_one, t_ref, t_batch);
    for (i = 0; i < B; ++i)
        printf("%s\"%s\"", i ? ", " : "", got[i].hyp);
    printf("]}\n");
    if (g_multi) psgpu_multi_free(g_multi); else psgpu_batch_free(dev);
    return (bad_batch || bad_rev || bad_one) ? 1 : 0;
}
