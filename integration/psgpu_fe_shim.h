/* integration/psgpu_fe_shim.h -- reference-side binding of the psgpu MFCC front
 * end: fe_process_utt + fe_end_utt (fe/fe_interface.c:505-541) computed on the
 * MI355X from the tables of the decoder's own fe_t.  See INTEGRATION.md section 4. */
#ifndef PSGPU_FE_SHIM_H
#define PSGPU_FE_SHIM_H

#include <pocketsphinx.h>
#include "fe/fe.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psgpu_fe_shim_s psgpu_fe_shim_t;

/* Uploads the tables of `fe` (window, twiddle factors, mel filters, DCT matrix,
 * lifter; fe_internal.h:70-161).  NULL (message through E_ERROR) when no gfx950
 * device is usable or the configuration cannot be reproduced (dither). */
psgpu_fe_shim_t *psgpu_fe_wrap(fe_t *fe);
void psgpu_fe_shim_free(psgpu_fe_shim_t *s);
/* hand over the underlying device object (psgpu.h) and drop the wrapper */
struct psgpu_fe_s *psgpu_fe_shim_release(psgpu_fe_shim_t *s);

/* = fe_reset_noisestats(fe->noise_stats) (what ps_start_stream does, pocketsphinx.c:1081) */
void psgpu_fe_shim_reset_noise(psgpu_fe_shim_t *s);

/* = fe_start_utt + fe_process_utt + fe_end_utt: all frames of one utterance
 * including the zero-padded tail frame.  *cep_block is allocated with
 * ckd_calloc_2d (free with ckd_free_2d), like fe_process_utt's.  The noise
 * tracker is carried from call to call like fe->noise_stats.  0 or -1. */
int psgpu_fe_process_utt(psgpu_fe_shim_t *s, int16 const *spch, size_t nsamps,
                         mfcc_t ***cep_block, int32 *nframes);

/* = ps_process_raw(ps, data, n, FALSE, TRUE) between ps_start_utt and ps_end_utt,
 * with the cepstra computed on the device (-> ps_process_cep, full_utt). */
int psgpu_process_raw_full(ps_decoder_t *ps, psgpu_fe_shim_t *s, int16 const *data, size_t n_samples);

#ifdef __cplusplus
}
#endif
#endif
