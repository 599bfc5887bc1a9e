/* The reference's ngram_search_fwdtree.c compiled with the HMM evaluation of
 * evaluate_channels() (:701-715) routed through psgpu (see
 * psgpu_search_hooks.h).  The source is included from where it lies. */
#include "psgpu_search_hooks.h"
#undef hmm_context_set_senscore
#define hmm_context_set_senscore(ctx, scr) psgpu_fwdtree_pre_evaluate(ngs, (scr), frame_idx)
#define hmm_vit_eval(h) psgpu_hmm_vit_result(h)
#include "ngram_search_fwdtree.c"
