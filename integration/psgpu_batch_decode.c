/* integration/psgpu_batch_decode.c -- REFERENCE-SIDE example program.
 *
 * Throughput use of the bindings: what `pocketsphinx_batch`
 * (programs/pocketsphinx_batch.c) does -- decode a list of utterances and report
 * CPU/wall time per second of speech -- with one decoder per host thread, every
 * decoder's GMM scorer swapped for the MI355X one (psgpu_mgau_attach).  The
 * reference decodes on one thread; its decoder objects are independent
 * (one ps_decoder_t per thread is the supported pattern), and so
 * are the psgpu objects (model + state + HIP stream per decoder), so utterances
 * shard over host threads the same way they shard over GPUs: no shared state,
 * no collective.
 *
 * usage: psgpu_batch_decode MODELDIR LM|- DICT|- RAW N_UTT N_THREADS gpu|cpu [key val ...]
 * Prints one JSON line: wall time, xRT, frames/s, and whether every hypothesis
 * equals the first one.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "psgpu_mgau_shim.h"

typedef struct worker_s {
    pthread_t tid;
    ps_decoder_t *ps;
    const int16 *pcm;
    size_t n_samples;
    int n_utt;
    int n_frames;                 /* frames decoded by this worker */
    int mismatches;               /* hypotheses different from the reference string */
    char hyp0[1024];
    pthread_barrier_t *start;
} worker_t;

static double
now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void *
work(void *arg)
{
    worker_t *w = arg;
    int u;
    pthread_barrier_wait(w->start);
    for (u = 0; u < w->n_utt; ++u) {
        const char *hyp;
        int32 score;
        ps_start_utt(w->ps);
        ps_process_raw(w->ps, w->pcm, w->n_samples, FALSE, TRUE);
        ps_end_utt(w->ps);
        hyp = ps_get_hyp(w->ps, &score);
        if (hyp == NULL) hyp = "";
        if (u == 0)
            snprintf(w->hyp0, sizeof w->hyp0, "%s", hyp);
        else if (strcmp(hyp, w->hyp0) != 0 && u > 1)
            ++w->mismatches;      /* utterance 0 starts from the initial CMN state, skip it */
        w->n_frames += ps_get_n_frames(w->ps);
    }
    return NULL;
}

int
main(int argc, char **argv)
{
    int n_utt, n_thr, use_gpu, t, i, bad = 0, frames = 0;
    worker_t *w;
    pthread_barrier_t start;
    FILE *fp; long sz; int16 *pcm;
    double t0, wall;

    if (argc < 8) {
        fprintf(stderr, "usage: psgpu_batch_decode MODELDIR LM|- DICT|- RAW N_UTT N_THREADS gpu|cpu [key val ...]\n");
        return 2;
    }
    n_utt = atoi(argv[5]); n_thr = atoi(argv[6]); use_gpu = !strcmp(argv[7], "gpu");
    if (n_thr < 1 || n_utt < n_thr) { fprintf(stderr, "need N_UTT >= N_THREADS >= 1\n"); return 2; }
    fp = fopen(argv[4], "rb");
    if (!fp) { perror(argv[4]); return 2; }
    fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    pcm = malloc(sz);
    if (fread(pcm, 1, sz, fp) != (size_t)sz) { perror("read"); return 2; }
    fclose(fp);
    err_set_loglevel(ERR_ERROR);

    w = calloc(n_thr, sizeof *w);
    pthread_barrier_init(&start, NULL, n_thr + 1);
    for (t = 0; t < n_thr; ++t) {
        ps_config_t *config = ps_config_init(NULL);
        ps_config_set_str(config, "hmm", argv[1]);
        if (strcmp(argv[2], "-")) ps_config_set_str(config, "lm", argv[2]);
        if (strcmp(argv[3], "-")) ps_config_set_str(config, "dict", argv[3]);
        ps_config_set_str(config, "loglevel", "ERROR");
        for (i = 8; i + 1 < argc; i += 2)
            if (ps_config_set_str(config, argv[i][0] == '-' ? argv[i] + 1 : argv[i], argv[i + 1]) == NULL) {
                fprintf(stderr, "bad config %s\n", argv[i]); return 2;
            }
        w[t].ps = ps_init(config);
        if (!w[t].ps) { fprintf(stderr, "ps_init failed\n"); return 2; }
        if (use_gpu && psgpu_mgau_attach(w[t].ps) < 0) {
            fprintf(stderr, "psgpu_mgau_attach failed\n");
            return 3;
        }
        w[t].pcm = pcm; w[t].n_samples = sz / 2;
        w[t].n_utt = n_utt / n_thr + (t < n_utt % n_thr);
        w[t].start = &start;
        pthread_create(&w[t].tid, NULL, work, &w[t]);
    }
    pthread_barrier_wait(&start);
    t0 = now_s();
    for (t = 0; t < n_thr; ++t)
        pthread_join(w[t].tid, NULL);
    wall = now_s() - t0;
    for (t = 0; t < n_thr; ++t) {
        frames += w[t].n_frames;
        bad += w[t].mismatches;
    }
    /* steady-state hypotheses must agree across workers too (each worker's utterance 1) */
    printf("{\"mode\": \"%s\", \"threads\": %d, \"utterances\": %d, \"frames\": %d, \"wall_s\": %.4f, "
           "\"xrt\": %.6f, \"frames_per_s\": %.1f, \"hyp\": \"%s\", \"hyp_mismatches\": %d, \"mgau\": \"%s\"}\n",
           use_gpu ? "gpu" : "cpu", n_thr, n_utt, frames, wall, wall / (frames / 100.0), frames / wall,
           w[0].hyp0, bad, w[0].ps->acmod->mgau->vt->name);
    for (t = 0; t < n_thr; ++t)
        ps_free(w[t].ps);
    return bad ? 1 : 0;
}
