/* integration/psgpu_decode_batch.h -- the additive batch call of SURVEY 8(b):
 * decode B whole utterances with the reference's own decoder objects, GMM scoring
 * (and optionally the front end and the Viterbi step) on the MI355X.  Built with
 * the same objects as the drop-in path, so B = 1 IS the drop-in path.
 * See INTEGRATION.md section 3. */
#ifndef PSGPU_DECODE_BATCH_H
#define PSGPU_DECODE_BATCH_H

#include <pocketsphinx.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSGPU_BATCH_DEVICE_FE     1u   /* cepstra of the whole batch in one device call (psgpu_fe_shim) */
#define PSGPU_BATCH_DEVICE_SEARCH 2u   /* hmm_vit_eval loops on the device (needs the hooked library) */
#define PSGPU_BATCH_CPU_ONLY      4u   /* no device at all: the reference as it is (for A/B runs) */
#define PSGPU_BATCH_DEVICE_PHONE_LOOP 8u /* each utterance's phone-loop search in one device launch (psgpu_phone_loop_shim) */
#define PSGPU_BATCH_DEVICE_FIRST_PASS 16u /* the WHOLE first pass of the batch on the device in one launch set: front end,
                                          * features, scores, phone loop, lexicon-tree search (psgpu_device_decode.h); the
                                          * workers only read the results out (ps_get_hyp / ps_seg_iter on the injected
                                          * tables; with -bestpath yes the reference's lattice pass runs on them on the
                                          * worker's host thread).  With -fwdflat yes the environment must say
                                          * PSGPU_DEVICE_SECOND_PASS=1: the flat-lexicon pass of the batch then runs on the device
                                          * too (psgpu_decode_second_pass) and ITS tables are injected; otherwise
                                          * psgpu_batch_init refuses the flag (the reference's own second pass wants the
                                          * utterance's feature vectors in acmod; behind ps_decode_raw --
                                          * psgpu_device_search_attach -- it has them) */

typedef struct psgpu_batch_seg_s {
    char *word;
    int32 sf, ef, ascr, lscr, lback;   /* ps_seg_frames / ps_seg_prob */
} psgpu_batch_seg_t;

typedef struct psgpu_batch_result_s {
    char *hyp;                         /* ps_get_hyp ("" when there is none) */
    int32 score;
    int32 n_frames;
    int32 n_seg;
    psgpu_batch_seg_t *seg;
} psgpu_batch_result_t;

typedef struct psgpu_batch_s psgpu_batch_t;

/* n_workers decoders (ps_init(config) each, one host thread each), every one with its
 * own psgpu model/state/stream on the current device (psgpu_set_device).  NULL on failure
 * (also when the device bindings cannot be attached: no silent CPU fallback unless
 * PSGPU_BATCH_CPU_ONLY is asked for). */
psgpu_batch_t *psgpu_batch_init(ps_config_t *config, int n_workers, unsigned flags);
void psgpu_batch_free(psgpu_batch_t *b);

/* Decode utterances pcm[u][0..n[u]) (16-bit, the decoder's sample rate) with every
 * pass the configuration enables.  Each utterance is decoded from the state a decoder
 * has after ps_start_stream() on its first utterance (noise tracker and top-N history
 * reset), so results do not depend on B, on the worker count or on the order: out[u]
 * equals what `ps_start_stream; ps_start_utt; ps_process_raw(full_utt); ps_end_utt`
 * gives on a fresh decoder.  out[] entries are filled (free with
 * psgpu_batch_result_clear).  Returns 0, or -1 if any utterance failed. */
int psgpu_decode_batch(psgpu_batch_t *b, const int16 *const pcm[], const size_t n[], int B,
                       psgpu_batch_result_t out[]);
void psgpu_batch_result_clear(psgpu_batch_result_t *r);

/* ---- several devices.  One psgpu_batch_t per entry of devices[] (psgpu_set_device(devices[k]) while it is built: every
 * decoder's device objects live there; an index may appear more than once -- two batch objects then share that GPU),
 * n_workers decoders each.  psgpu_decode_batch_multi splits the B utterances into n_devices consecutive blocks of about
 * equal audio length, decodes every block with its device's psgpu_decode_batch on a host thread of its own, and returns
 * the results in the callers' order: the utterances are independent (see psgpu_decode_batch), so nothing crosses devices
 * but the PCM going out and the results coming back.  Returns 0, or -1 if any utterance failed. */
typedef struct psgpu_multi_s psgpu_multi_t;
psgpu_multi_t *psgpu_multi_init(ps_config_t *config, const int devices[], int n_devices, int n_workers, unsigned flags);
void psgpu_multi_free(psgpu_multi_t *m);
int psgpu_multi_n_devices(const psgpu_multi_t *m);
int psgpu_decode_batch_multi(psgpu_multi_t *m, const int16 *const pcm[], const size_t n[], int B,
                             psgpu_batch_result_t out[]);

#ifdef __cplusplus
}
#endif
#endif
