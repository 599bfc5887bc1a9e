/* integration/psgpu_phone_loop_shim.h -- reference-side binding of the device phone-loop
 * search (psgpu_phone_loop_run_dev): the decoder's phone_loop_search_t keeps its object
 * and its `penalties` vector, its per-frame step is answered from one device launch per
 * utterance.  See INTEGRATION.md section 2c. */
#ifndef PSGPU_PHONE_LOOP_SHIM_H
#define PSGPU_PHONE_LOOP_SHIM_H

#include <pocketsphinx.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Needs psgpu_mgau_attach(ps) first (PTM scorer with look-ahead), full-utterance decoding
 * (ps_process_raw / ps_process_cep with full_utt = TRUE), at most 64 CI phones, non-multiplex
 * 3- or 5-state HMMs.  0, or -1 (decoder untouched).  Utterances that do not fit (streaming
 * input, no cache) run the reference's own step, frame by frame, as before. */
int psgpu_phone_loop_attach(ps_decoder_t *ps);
void psgpu_phone_loop_detach(ps_decoder_t *ps);
/* steps answered from the device / steps run by the reference code */
void psgpu_phone_loop_stats(ps_decoder_t *ps, long *n_device, long *n_host);

#ifdef __cplusplus
}
#endif
#endif
