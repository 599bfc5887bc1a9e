/* integration/psgpu_device_decode.h -- the first pass of an utterance entirely on the MI355X behind
 * the reference's own result API (ps_get_hyp, ps_seg_iter, ...).  See INTEGRATION.md section 2d. */
#ifndef PSGPU_DEVICE_DECODE_H
#define PSGPU_DEVICE_DECODE_H

#include <pocketsphinx.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psgpu_device_decode_s psgpu_device_decode_t;

/* Reads the decoder's search structures (tree, dictionary, dict2pid, beams, phone loop, language model
 * as a dense table: small vocabularies) and uploads them.  Needs psgpu_mgau_attach(ps) first, the n-gram
 * search with -fwdflat no (-bestpath yes then runs on the host over the injected table), the 1s_c_d_dd feature type, pl_window > 0.  NULL on failure. */
psgpu_device_decode_t *psgpu_device_decode_attach(ps_decoder_t *ps);
void psgpu_device_decode_detach(psgpu_device_decode_t *d);

/* = ps_start_utt; ps_process_raw(pcm, n, FALSE, TRUE); ps_end_utt -- with front end, features, senone
 * scores, phone loop and lexicon-tree search on the device; afterwards the decoder's back-pointer
 * table, score stack and frame marks hold the result in the reference's layout, so ps_get_hyp(),
 * ps_seg_iter() etc. work as after a host decode.  Returns the number of frames searched, or -1. */
int psgpu_device_decode_utt(psgpu_device_decode_t *d, int16 const *pcm, size_t n_samples);

#ifdef __cplusplus
}
#endif
#endif
