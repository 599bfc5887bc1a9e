/* The reference's state_align_search.c compiled with its hmm_vit_eval loop routed through psgpu
 * (see psgpu_search_hooks.h).  The source is included from where it lies. */
#include "psgpu_search_hooks.h"
#undef hmm_context_set_senscore
#define hmm_context_set_senscore(ctx, scr) psgpu_state_align_pre_evaluate(sas, (scr), frame_idx)
#define hmm_vit_eval(h) psgpu_hmm_vit_result(h)
#include "state_align_search.c"
