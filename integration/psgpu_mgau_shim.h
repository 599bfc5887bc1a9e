/* integration/psgpu_mgau_shim.h -- reference-side binding of the psgpu scorer
 * behind PocketSphinx's ps_mgau_t vtable (acmod.h:98-116).  See INTEGRATION.md. */
#ifndef PSGPU_MGAU_SHIM_H
#define PSGPU_MGAU_SHIM_H

#include <stdint.h>
#include <pocketsphinx.h>
#include "acmod.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Wrap the decoder's CPU "ptm" scorer: returns a ps_mgau_t whose frame_eval
 * runs on the MI355X, or NULL (message through E_ERROR) if the model shape is
 * unsupported or no gfx950 device is usable.  The returned object owns
 * `cpu_mgau` and frees it in its vt->free. */
ps_mgau_t *psgpu_mgau_wrap(ps_mgau_t *cpu_mgau);

/* ps->acmod->mgau := psgpu_mgau_wrap(ps->acmod->mgau).  0 on success, -1 on
 * failure (the decoder is left untouched and keeps its CPU scorer). */
int psgpu_mgau_attach(ps_decoder_t *ps);

/* Back to the top-N history of a freshly initialised scorer (ptm_mgau_reset_fast_hist,
 * ptm_mgau.c:777-802).  0, or -1 if `mgau` is not a psgpu scorer. */
int psgpu_mgau_reset(ps_mgau_t *mgau);

/* history slot `slot` of a wrapped PTM scorer := the codeword lists cw [n_chain][topn] (see psgpu_mgau_shim.c) */
int psgpu_mgau_seed_history(ps_mgau_t *mgau, int slot, const int32 *cw);
/* ... and reads them (the lists that will seed the next utterance's first frame are slot n_fast_hist - 1's) */
int psgpu_mgau_get_history(ps_mgau_t *ps, int slot, int32 *cw);

/* Hooks for a device-side search component (integration/psgpu_phone_loop_shim.c): score what
 * lies ahead of `frame` now and return the device rows; account for a fresh frame_eval call
 * that component no longer makes.  0, or -1 when the scorer is not the psgpu PTM scorer with
 * an attached acmod or the request does not fit the cache. */
int psgpu_mgau_prefetch(ps_mgau_t *mgau, int frame, const int16_t **raw_dev, const int32_t **best_dev,
                        int *frame0, int *n_frames, int *n_sen);
int psgpu_mgau_mark_fresh(ps_mgau_t *mgau, int frame);
/* the device model behind a wrapped PTM scorer, or NULL */
struct psgpu_ptm_model_s *psgpu_mgau_ptm_model(ps_mgau_t *mgau);
/* the device model behind a wrapped semi-continuous ("s2_semi") scorer, or NULL */
struct psgpu_semi_model_s *psgpu_mgau_semi_model(ps_mgau_t *mgau);
/* the device model behind a wrapped multi-stream ("ms") scorer, or NULL */
struct psgpu_ms_model_s *psgpu_mgau_ms_model(ps_mgau_t *mgau);

/* number of frame_eval calls served by the device (-1 if not a psgpu scorer) */
int32 psgpu_mgau_n_calls(ps_mgau_t *mgau);
/* of which answered from the look-ahead cache (one batched pass per utterance pass) */
long psgpu_mgau_n_cache_served(ps_mgau_t *mgau);

#ifdef __cplusplus
}
#endif
#endif
