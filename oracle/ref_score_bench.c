/* oracle/ref_score_bench.c -- TEST INFRASTRUCTURE: CPU baseline runner.
 *
 * Times the UNMODIFIED reference's ptm_mgau_frame_eval(compallsen=TRUE)
 * (src/ptm_mgau.c:408-454), single thread, over a feature matrix split into
 * utterances, with a fresh top-N history per utterance -- the same workload
 * bench.py gives the GPU.  Prints one JSON line.  Optionally writes the int16
 * scores so the caller can check them against the GPU output.
 *
 * usage: ref_score_bench MODELDIR FEATS.f32 UTT_LEN [SCORES_OUT.i16]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "ptm_mgau.h"

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv)
{
    ps_config_t *config;
    ps_decoder_t *ps;
    ptm_mgau_t *s;
    FILE *fp;
    long sz;
    float *feat;
    int dim = 0, f, nfr, t, seglen, n_sen;
    int16 *scr;
    mfcc_t *ptr[16];
    double t0, t1;
    FILE *out = NULL;

    if (argc < 4) { fprintf(stderr, "usage: %s MODELDIR FEATS.f32 UTT_LEN [OUT.i16]\n", argv[0]); return 2; }
    err_set_loglevel(ERR_ERROR);
    config = ps_config_init(NULL);
    ps_config_set_str(config, "hmm", argv[1]);
    ps_config_set_str(config, "loglevel", "ERROR");
    ps_config_set_bool(config, "compallsen", TRUE);
    /* acoustic model only: no LM / dictionary needed for scoring */
    ps_config_set_str(config, "allphone", NULL);
    ps = ps_init(config);
    if (!ps) {
        /* ps_init without a search is fine in 5.x; if it is not, bail out */
        fprintf(stderr, "ps_init failed\n"); return 2;
    }
    s = (ptm_mgau_t *)ps->acmod->mgau;
    if (strcmp(ps->acmod->mgau->vt->name, "ptm")) { fprintf(stderr, "not PTM\n"); return 2; }
    n_sen = s->n_sen;
    for (f = 0; f < s->g->n_feat; ++f) dim += s->g->featlen[f];

    fp = fopen(argv[2], "rb");
    if (!fp) { perror(argv[2]); return 2; }
    fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    feat = malloc(sz);
    if (fread(feat, 1, sz, fp) != (size_t)sz) { perror("read"); return 2; }
    fclose(fp);
    nfr = (int)(sz / 4 / dim);
    seglen = atoi(argv[3]);
    if (seglen <= 0) seglen = nfr;
    if (argc > 4) out = fopen(argv[4], "wb");
    scr = malloc(sizeof(int16) * n_sen);

    t0 = now();
    for (t = 0; t < nfr; ++t) {
        int frame = t % seglen, o = 0;
        for (f = 0; f < s->g->n_feat; ++f) { ptr[f] = feat + (size_t)t * dim + o; o += s->g->featlen[f]; }
        if (frame == 0) {
            ps_mgau_base(s)->frame_idx = 0;             /* acmod_start_utt */
            ptm_mgau_reset_fast_hist(ps_mgau_base(s));  /* fresh decoder per utterance */
        }
        ptm_mgau_frame_eval(ps_mgau_base(s), scr, NULL, 0, ptr, frame, TRUE);
        ps_mgau_base(s)->frame_idx++;                   /* acmod_advance */
        if (out) fwrite(scr, sizeof(int16), n_sen, out);
    }
    t1 = now();
    if (out) fclose(out);
    printf("{\"frames\": %d, \"seconds\": %.6f, \"frames_per_s\": %.2f, \"threads\": 1}\n",
           nfr, t1 - t0, nfr / (t1 - t0));
    return 0;
}
