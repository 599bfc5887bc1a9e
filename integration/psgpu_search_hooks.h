/* integration/psgpu_search_hooks.h -- REFERENCE-SIDE BINDING.
 *
 * The three places where the reference runs hmm_vit_eval() over an active
 * HMM population all look the same:
 *
 *     hmm_context_set_senscore(X->hmmctx, senscr);        (hmm.h:226)
 *     for every active HMM h:  score = hmm_vit_eval(h);   (hmm.c:786-805)
 *                              best = max(best, score);
 *
 *   evaluate_channels()  ngram_search_fwdtree.c:701-715 (+ eval_*_chan :605-699)
 *   ngram_fwdflat_search() / fwdflat_eval_chan()  ngram_search_fwdflat.c:842, :444-480
 *   evaluate_hmms()      phone_loop_search.c:202-222
 *
 * A maintainer's patch changes exactly those two statements: the first
 * becomes "evaluate the whole population on the device now", the second
 * becomes "pick up the result".  This header expresses that patch as two
 * macro overrides so that the UNMODIFIED reference sources can be compiled
 * with it (integration/hook_*.c include them in place; nothing is copied):
 *
 *   hmm_context_set_senscore(ctx, scr) -> psgpu_<site>_pre_evaluate(...)
 *       walks the same population in the same order with the same activity
 *       tests, packs the hmm_t fields into psgpu_hmm_rec_t lines, runs ONE
 *       psgpu_hmm_vit_eval() and writes the updated fields back;
 *   hmm_vit_eval(h) -> the stored result (h->bestscore is what
 *       hmm_vit_eval returns, hmm.c:606,706,349,524).
 *
 * Both fall through to the reference behaviour while the HMM context has no
 * device context attached (hmm_context_t.udata == NULL, hmm.h:153: the field
 * is unused by the reference), so one library serves both paths and the
 * choice is per decoder: psgpu_search_attach(ps).
 */
#ifndef PSGPU_SEARCH_HOOKS_H
#define PSGPU_SEARCH_HOOKS_H

#include <pocketsphinx.h>
#include "pocketsphinx_internal.h"
#include "hmm.h"
#include "ngram_search.h"
#include "phone_loop_search.h"
#include "fsg_search_internal.h"
#include "allphone_search.h"
#include "kws_search.h"
#include "state_align_search.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Attach device HMM contexts to the decoder's n-gram search (fwdtree and
 * fwdflat share one hmm_context_t, ngram_search.h:203) and to its phone-loop
 * search (phone_loop_search.h:79); or, when the active search is an fsg / allphone
 * / kws / state_align search, to that search's hmm_context_t.  0 on success, -1 on
 * failure (decoder untouched). */
int psgpu_search_attach(ps_decoder_t *ps);
/* Detach and free them (also safe on a decoder that was never attached). */
void psgpu_search_detach(ps_decoder_t *ps);
/* Counters: batched device steps served / HMMs evaluated on the device. */
void psgpu_search_stats(ps_decoder_t *ps, long *n_batches, long *n_hmms);

void psgpu_fwdtree_pre_evaluate(ngram_search_t *ngs, int16 const *senscr, int frame_idx);
void psgpu_fwdflat_pre_evaluate(ngram_search_t *ngs, int16 const *senscr, int frame_idx);
void psgpu_phone_loop_pre_evaluate(phone_loop_search_t *pls, int16 const *senscr, int frame_idx);
/* the other consumers of hmm_vit_eval, same two-statement patch:
 *   fsg_search_step / fsg_search_hmm_eval   fsg_search.c:706, :335-385
 *   phmm_eval_all                           allphone_search.c:346-375
 *   kws_search_hmm_eval                     kws_search.c:194-226
 *   evaluate_hmms                           state_align_search.c:64-84 */
void psgpu_fsg_pre_evaluate(fsg_search_t *fsgs, int16 const *senscr);
void psgpu_allphone_pre_evaluate(allphone_search_t *allphs, int16 const *senscr);
void psgpu_kws_pre_evaluate(kws_search_t *kwss, int16 const *senscr);
void psgpu_state_align_pre_evaluate(state_align_search_t *sas, int16 const *senscr, int frame_idx);

/* hmm_vit_eval as seen by the hooked loops */
static inline int32
psgpu_hmm_vit_result(hmm_t *h)
{
    return h->ctx->udata ? h->bestscore : (hmm_vit_eval)(h);
}

#ifdef __cplusplus
}
#endif
#endif
