/* TEST INFRASTRUCTURE (checker): a group of live decoders against the same number of CPU decoders.
 *
 *   streams_decode MODELDIR LM DICT DATADIR N "id,id,...;id,...;..." CHUNK0,CHUNK1,... [key value ...]
 *
 * N reference decoders with the psgpu scorer and the device search bound (psgpu_mgau_attach, psgpu_device_decode_attach,
 * psgpu_device_search_attach) in ONE group (psgpu_live_group_create: one device pipeline in streams mode), and N unmodified
 * CPU decoders.  Stream s decodes its list of recordings (DATADIR/id.raw) one after another through ONE decoder, CHUNKs samples
 * a piece.  A round feeds every stream that has audio left one piece (ps_process_raw without full_utt on both decoders of the
 * stream), steps the group once, and asks every stream for ps_get_hyp: the device side must say what the CPU decoder says.  A
 * stream whose recording is used up gets ps_end_utt on both sides (final hypothesis, score, segmentation compared) and
 * ps_start_utt for its next one.  Prints one JSON line. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "ngram_search.h"
#include "psgpu_mgau_shim.h"
#include "psgpu_device_decode.h"

#define MAXS 16
#define MAXU 16

static ps_decoder_t *
make_decoder(const char *modeldir, const char *lm, const char *dict, int argc, char **argv)
{
    ps_config_t *config = ps_config_init(NULL);
    ps_decoder_t *ps;
    int i;
    ps_config_set_str(config, "hmm", modeldir);
    ps_config_set_str(config, "lm", lm);
    ps_config_set_str(config, "dict", dict);
    ps_config_set_str(config, "loglevel", "ERROR");
    for (i = 0; i + 1 < argc; i += 2)
        if (ps_config_set_str(config, argv[i][0] == '-' ? argv[i] + 1 : argv[i], argv[i + 1]) == NULL) { fprintf(stderr, "bad config %s\n", argv[i]); exit(2); }
    ps = ps_init(config);
    if (!ps) { fprintf(stderr, "ps_init failed\n"); exit(2); }
    return ps;
}

static int16 *
read_raw(const char *path, size_t *n)
{
    FILE *fp = fopen(path, "rb");
    long len;
    int16 *buf;
    if (!fp) { perror(path); exit(2); }
    fseek(fp, 0, SEEK_END); len = ftell(fp); fseek(fp, 0, SEEK_SET);
    buf = malloc(len + 2);
    if (fread(buf, 1, len, fp) != (size_t)len) { perror(path); exit(2); }
    fclose(fp);
    *n = len / 2;
    return buf;
}

static void
segs(ps_decoder_t *ps, char *out, size_t cap)
{
    ps_seg_t *seg;
    size_t o = 0;
    out[0] = 0;
    for (seg = ps_seg_iter(ps); seg; seg = ps_seg_next(seg)) {
        int sf, ef; int32 a, l, b;
        ps_seg_frames(seg, &sf, &ef); ps_seg_prob(seg, &a, &l, &b);
        if (o + 128 < cap) o += snprintf(out + o, cap - o, "%s %d %d %d %d %d\n", ps_seg_word(seg), sf, ef, a, l, b);
    }
}

int
main(int argc, char **argv)
{
    ps_decoder_t *cpu[MAXS], *gpu[MAXS];
    psgpu_device_decode_t *dd[MAXS];
    psgpu_live_group_t *grp;
    int16 *pcm[MAXS][MAXU]; size_t len[MAXS][MAXU]; int n_utt[MAXS], cur[MAXS], chunk[MAXS];
    size_t at[MAXS];
    int N, s, rounds = 0, n_partial = 0, partial_bad = 0, n_final = 0, final_bad = 0, live[MAXS];
    long steps = 0, searched, frames = 0;
    char *lists, *chunks, *tok, *save;
    char first_bad[512] = "";
    static char sa[65536], sb[65536];

    if (argc < 8) { fprintf(stderr, "usage: %s MODELDIR LM DICT DATADIR N LISTS CHUNKS [key value ...]\n", argv[0]); return 2; }
    N = atoi(argv[5]);
    if (N < 1 || N > MAXS) return 2;
    err_set_loglevel(ERR_ERROR);
    lists = strdup(argv[6]); chunks = strdup(argv[7]);
    for (s = 0, tok = strtok_r(lists, ";", &save); s < N; ++s, tok = strtok_r(NULL, ";", &save)) {
        char *id, *sv2, *l2;
        if (!tok) { fprintf(stderr, "need %d lists\n", N); return 2; }
        l2 = strdup(tok);
        n_utt[s] = 0;
        for (id = strtok_r(l2, ",", &sv2); id && n_utt[s] < MAXU; id = strtok_r(NULL, ",", &sv2)) {
            char path[1024];
            snprintf(path, sizeof path, "%s/%s.raw", argv[4], id);
            pcm[s][n_utt[s]] = read_raw(path, &len[s][n_utt[s]]);
            ++n_utt[s];
        }
    }
    for (s = 0, tok = strtok_r(chunks, ",", &save); s < N; ++s, tok = strtok_r(NULL, ",", &save))
        chunk[s] = tok ? atoi(tok) : 4096;
    for (s = 0; s < N; ++s) {
        cpu[s] = make_decoder(argv[1], argv[2], argv[3], argc - 8, argv + 8);
        gpu[s] = make_decoder(argv[1], argv[2], argv[3], argc - 8, argv + 8);
        if (psgpu_mgau_attach(gpu[s]) < 0) { fprintf(stderr, "psgpu_mgau_attach failed\n"); return 3; }
        dd[s] = psgpu_device_decode_attach(gpu[s]);
        if (!dd[s] || psgpu_device_search_attach(dd[s]) < 0) { fprintf(stderr, "device attach failed\n"); return 3; }
    }
    grp = psgpu_live_group_create(dd, N, 4000, 64);
    if (!grp) { fprintf(stderr, "psgpu_live_group_create failed\n"); return 3; }
    for (s = 0; s < N; ++s) {
        cur[s] = 0; at[s] = 0; live[s] = n_utt[s] > 0;
        if (live[s]) { ps_start_utt(cpu[s]); ps_start_utt(gpu[s]); }
    }
    for (;;) {
        int any = 0;
        for (s = 0; s < N; ++s) {
            size_t k;
            if (!live[s]) continue;
            any = 1;
            k = len[s][cur[s]] - at[s] < (size_t)chunk[s] ? len[s][cur[s]] - at[s] : (size_t)chunk[s];
            ps_process_raw(cpu[s], pcm[s][cur[s]] + at[s], k, FALSE, FALSE);
            ps_process_raw(gpu[s], pcm[s][cur[s]] + at[s], k, FALSE, FALSE);
            at[s] += k;
        }
        if (!any) break;
        ++rounds;
        if (psgpu_live_group_step(grp) < 0) { fprintf(stderr, "group step failed\n"); return 3; }
        for (s = 0; s < N; ++s) {
            int32 sa_ = 0, sb_ = 0;
            const char *ha, *hb;
            if (!live[s]) continue;
            ha = ps_get_hyp(cpu[s], &sa_); snprintf(sa, 4096, "%s", ha ? ha : "");
            hb = ps_get_hyp(gpu[s], &sb_);
            ++n_partial;
            if (strcmp(sa, hb ? hb : "") || sa_ != sb_) {
                if (!partial_bad++) snprintf(first_bad, sizeof first_bad, "round %d stream %d: cpu '%.150s' %d, device '%.150s' %d", rounds, s, sa, sa_, hb ? hb : "", sb_);
            }
            if (at[s] == len[s][cur[s]]) {                        /* the recording is used up: the utterance ends */
                int32 fa = 0, fb = 0;
                ps_end_utt(cpu[s]); ps_end_utt(gpu[s]);
                ha = ps_get_hyp(cpu[s], &fa); snprintf(sa, 4096, "%s", ha ? ha : "");
                hb = ps_get_hyp(gpu[s], &fb);
                ++n_final;
                frames += ((ngram_search_t *)gpu[s]->search)->n_frame;
                if (strcmp(sa, hb ? hb : "") || fa != fb) {
                    if (!final_bad++ && !first_bad[0]) snprintf(first_bad, sizeof first_bad, "final stream %d utt %d: cpu '%.150s' %d, device '%.150s' %d", s, cur[s], sa, fa, hb ? hb : "", fb);
                }
                else {
                    segs(cpu[s], sa, sizeof sa); segs(gpu[s], sb, sizeof sb);
                    if (strcmp(sa, sb)) { if (!final_bad++ && !first_bad[0]) snprintf(first_bad, sizeof first_bad, "segmentation stream %d utt %d", s, cur[s]); }
                }
                ++cur[s]; at[s] = 0;
                if (cur[s] < n_utt[s]) { ps_start_utt(cpu[s]); ps_start_utt(gpu[s]); }
                else live[s] = 0;
            }
        }
    }
    searched = psgpu_live_group_stats(grp, &steps);
    printf("{\"ok\": %s, \"streams\": %d, \"rounds\": %d, \"partial_results\": %d, \"partial_mismatches\": %d, \"final_results\": %d, "
           "\"final_mismatches\": %d, \"group_steps\": %ld, \"frames_searched\": %ld, \"frames\": %ld, \"first_mismatch\": \"%s\"}\n",
           (!partial_bad && !final_bad) ? "true" : "false", N, rounds, n_partial, partial_bad, n_final, final_bad, steps, searched, frames, first_bad);
    psgpu_live_group_free(grp);
    for (s = 0; s < N; ++s) { psgpu_device_decode_detach(dd[s]); ps_free(cpu[s]); ps_free(gpu[s]); }
    return (!partial_bad && !final_bad) ? 0 : 1;
}
