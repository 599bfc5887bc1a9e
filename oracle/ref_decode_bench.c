/* oracle/ref_decode_bench.c -- TEST / BASELINE INFRASTRUCTURE: the UNMODIFIED reference (oracle/_ref/libpocketsphinx.so)
 * decoding utterances on one host thread, timed.  bench.py's `cpu_baseline` and its per-utterance parity check.
 *
 * usage: ref_decode_bench MODELDIR LM DICT PCMFILE SAMPLES_PER_UTT [-- key val ...]
 *
 * PCMFILE: 16-bit samples, utterances of SAMPLES_PER_UTT back to back.  Each is decoded as
 * `ps_start_utt; ps_process_raw(full_utt); ps_end_utt` (= ps_decode_raw, pocketsphinx.c:1030-1070) from the state a
 * decoder has after ps_start_stream() on its first utterance (noise tracker reset, top-N history reset, multiplex HMMs'
 * senone-sequence ids as hmm_init leaves them): what a fresh decoder gives, and what the device pipeline reproduces.
 * Defaults: -fwdflat no -bestpath no (the fwdtree-only configuration); override after `--`.
 * One JSON line per utterance {"utt", "hyp", "score", "frames", "cpu_s", "seg": [[word, wid, sf, ef, ascr, lscr], ...]}
 * and a last line {"total": ...} with the summed CPU time of the decodes (the quantity pocketsphinx_batch prints as
 * "TOTAL ... seconds CPU", programs/pocketsphinx_batch.c:897). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "ptm_mgau.h"
#include "ngram_search.h"
#include "hmm.h"

static double
cpu_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void
fresh_mpx(hmm_t *h)
{
    int i;
    if (hmm_is_mpx(h))
        for (i = 1; i < hmm_n_emit_state(h); ++i) h->senid[i] = BAD_SSID;
}

static void
reset_decoder(ps_decoder_t *ps)
{
    ps_search_t *search = ps->search;
    ps_start_stream(ps);
    if (search && !strcmp(ps_search_type(search), PS_SEARCH_TYPE_NGRAM)) {
        ngram_search_t *ngs = (ngram_search_t *)search;
        int i;
        if (ngs->fwdtree && ngs->root_chan)
            for (i = 0; i < ngs->n_root_chan; ++i) fresh_mpx(&ngs->root_chan[i].hmm);
        if (ngs->word_chan && ngs->single_phone_wid)
            for (i = 0; i < ngs->n_1ph_words; ++i)
                if (ngs->word_chan[ngs->single_phone_wid[i]])
                    fresh_mpx(&((root_chan_t *)ngs->word_chan[ngs->single_phone_wid[i]])->hmm);
    }
    if (!strcmp(ps->acmod->mgau->vt->name, "ptm")) ptm_mgau_reset_fast_hist(ps->acmod->mgau);
}

int
main(int argc, char **argv)
{
    ps_config_t *config;
    ps_decoder_t *ps;
    FILE *fp;
    long sz;
    int16 *pcm;
    size_t per, n_utt, u;
    int i, frames_total = 0;
    double total = 0.0;

    if (argc < 6) { fprintf(stderr, "usage: ref_decode_bench MODELDIR LM DICT PCMFILE SAMPLES_PER_UTT [-- key val ...]\n"); return 2; }
    config = ps_config_init(NULL);
    ps_config_set_str(config, "hmm", argv[1]);
    ps_config_set_str(config, "lm", argv[2]);
    ps_config_set_str(config, "dict", argv[3]);
    ps_config_set_str(config, "loglevel", "ERROR");
    ps_config_set_bool(config, "fwdflat", 0);
    ps_config_set_bool(config, "bestpath", 0);
    for (i = 6; i < argc && strcmp(argv[i], "--"); ++i) ;
    for (++i; i + 1 < argc; i += 2)
        if (ps_config_set_str(config, argv[i], argv[i + 1]) == NULL) { fprintf(stderr, "bad config %s\n", argv[i]); return 2; }
    err_set_loglevel(ERR_ERROR);
    ps = ps_init(config);
    if (ps == NULL) { fprintf(stderr, "ps_init failed\n"); return 2; }
    fp = fopen(argv[4], "rb");
    if (!fp) { perror(argv[4]); return 2; }
    fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    pcm = malloc(sz ? sz : 2);
    if (fread(pcm, 1, sz, fp) != (size_t)sz) { perror("read"); return 2; }
    fclose(fp);
    per = (size_t)atol(argv[5]);
    if (per == 0) { fprintf(stderr, "SAMPLES_PER_UTT must be positive\n"); return 2; }
    n_utt = (size_t)sz / 2 / per;
    for (u = 0; u < n_utt; ++u) {
        const char *hyp;
        int32 score = 0;
        ps_seg_t *seg;
        double t0, dt;
        int first = 1;
        /* REFDEC_CHUNKS="1600,317,...": the utterance through ps_process_raw(full_utt = FALSE) in pieces of these sizes (the list
         * repeats) -- a live decoder: running cepstral mean, feature frames three cepstra behind (pocketsphinx.c:1210-1246).
         * REFDEC_SESSION=1: the decoder is NOT put back between the utterances (its noise tracker, cepstral mean and feature window go on). */
        const char *chunks = getenv("REFDEC_CHUNKS");
        if (!(getenv("REFDEC_SESSION") && u > 0)) reset_decoder(ps);
        t0 = cpu_s();
        if (chunks && *chunks) {
            int32 csz[64]; int nc = 0, k = 0; size_t at = 0;
            char *cp = strdup(chunks), *tok;
            for (tok = strtok(cp, ","); tok && nc < 64; tok = strtok(NULL, ",")) csz[nc++] = atoi(tok);
            free(cp);
            if (nc == 0 || ps_start_utt(ps) < 0) return 3;
            while (at < per) {
                size_t take = per - at < (size_t)csz[k % nc] ? per - at : (size_t)csz[k % nc];
                ++k;
                if (ps_process_raw(ps, pcm + u * per + at, take, FALSE, FALSE) < 0) return 3;
                at += take;
            }
            if (ps_end_utt(ps) < 0) return 3;
        }
        else
        if (ps_start_utt(ps) < 0 || ps_process_raw(ps, pcm + u * per, per, FALSE, TRUE) < 0 || ps_end_utt(ps) < 0) {
            fprintf(stderr, "decode of utterance %zu failed\n", u);
            return 3;
        }
        hyp = ps_get_hyp(ps, &score);
        dt = cpu_s() - t0;
        total += dt; frames_total += ps_get_n_frames(ps) - 1;
        printf("{\"utt\": %zu, \"hyp\": \"%s\", \"score\": %d, \"frames\": %d, \"cpu_s\": %.6f, \"n_bp\": %d, \"n_bss\": %d, \"seg\": [", u,
               hyp ? hyp : "", score, ps_get_n_frames(ps) - 1, dt, ((ngram_search_t *)ps->search)->bpidx,
               ((ngram_search_t *)ps->search)->bss_head);
        for (seg = ps_seg_iter(ps); seg; seg = ps_seg_next(seg)) {
            int sf, ef; int32 ascr, lscr, lback;
            ps_seg_frames(seg, &sf, &ef);
            ps_seg_prob(seg, &ascr, &lscr, &lback);
            printf("%s[\"%s\", %d, %d, %d, %d, %d]", first ? "" : ", ", ps_seg_word(seg), seg->wid, sf, ef, ascr, lscr);
            first = 0;
        }
        printf("]}\n");
    }
    printf("{\"total\": true, \"utterances\": %zu, \"frames\": %d, \"cpu_s\": %.6f, \"frames_per_s\": %.2f, \"xrt\": %.6f}\n", n_utt,
           frames_total, total, total > 0 ? frames_total / total : 0.0, frames_total ? total / (frames_total / 100.0) : 0.0);
    free(pcm);
    ps_free(ps);
    return 0;
}
