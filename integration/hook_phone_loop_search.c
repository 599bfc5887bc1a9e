/* The reference's phone_loop_search.c compiled with the HMM evaluation of
 * evaluate_hmms() (:202-222) routed through psgpu (see psgpu_search_hooks.h). */
#include "psgpu_search_hooks.h"
#undef hmm_context_set_senscore
#define hmm_context_set_senscore(ctx, scr) psgpu_phone_loop_pre_evaluate(pls, (scr), frame_idx)
#define hmm_vit_eval(h) psgpu_hmm_vit_result(h)
#include "phone_loop_search.c"
