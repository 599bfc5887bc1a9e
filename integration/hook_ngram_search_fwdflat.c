/* The reference's ngram_search_fwdflat.c compiled with the HMM evaluation of
 * ngram_fwdflat_search() (:842) / fwdflat_eval_chan() (:444-480) routed
 * through psgpu (see psgpu_search_hooks.h). */
#include "psgpu_search_hooks.h"
#undef hmm_context_set_senscore
#define hmm_context_set_senscore(ctx, scr) psgpu_fwdflat_pre_evaluate(ngs, (scr), frame_idx)
#define hmm_vit_eval(h) psgpu_hmm_vit_result(h)
#include "ngram_search_fwdflat.c"
