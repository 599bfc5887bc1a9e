/* integration/psgpu_device_decode.h -- the first pass of the n-gram search entirely on the MI355X behind
 * the reference's own interfaces (ps_search_t vtable, ps_get_hyp, ps_seg_iter, ...).  See INTEGRATION.md section 2d. */
#ifndef PSGPU_DEVICE_DECODE_H
#define PSGPU_DEVICE_DECODE_H

#include <pocketsphinx.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psgpu_device_decode_s psgpu_device_decode_t;

/* Reads the decoder's search structures (tree, dictionary, dict2pid, beams, phone loop; the language model as
 * its own trie, or as a dense table for small vocabularies) and uploads them.  Needs psgpu_mgau_attach(ps) first,
 * the n-gram search with -fwdtree yes, the 1s_c_d_dd feature type, pl_window > 0.  NULL on failure.
 * Detach BEFORE ps_free(ps).  Dictionary / language-model changes after attach are refused at decode time
 * (detach and attach again); MLLR updates are picked up. */
psgpu_device_decode_t *psgpu_device_decode_attach(ps_decoder_t *ps);
void psgpu_device_decode_detach(psgpu_device_decode_t *d);

/* ---- the ps_search_t binding (pocketsphinx_internal.h:86-127) ------------------------------------------
 * After this, the decoder's own n-gram search object runs its first pass on the device: start() as the
 * reference, step() buffers the frame's feature vector, finish() runs scorer -> phone loop -> lexicon-tree
 * search for the utterance on the MI355X and puts the back-pointer table, score stack and frame marks into the
 * ngram_search_t in the reference's layout; the host's phone-loop search step becomes a no-op (the device
 * pipeline runs its own).  Unmodified ps_decode_raw(), ps_process_raw() + ps_end_utt(), ps_get_hyp(),
 * ps_seg_iter(), ps_get_lattice() work on top; with -fwdflat yes / -bestpath yes the reference's own later
 * passes run on the injected table.  0 on success. */
int psgpu_device_search_attach(psgpu_device_decode_t *d);
void psgpu_device_search_detach(psgpu_device_decode_t *d);
/* Results in mid-utterance (ps_get_hyp / ps_seg_iter between ps_process_raw calls): the utterance in progress is a LIVE utterance
 * of the device pipeline (psgpu_decode_live_begin / _step) -- every read-out hands the frames the device has not seen yet over and
 * the device search goes on from where it stopped, as the reference's does between ps_search_forward rounds
 * (ngram_search_fwdtree.c:1454-1495), so a live decode costs O(T).  Since the utterance's start: frames the device search kernel
 * has stepped through (= the frames searched so far when each was searched once), live steps, and times the utterance outgrew the
 * capacity it was begun with and was begun again (3,000 frames at first, doubling). */
void psgpu_device_search_live_stats(psgpu_device_decode_t *d, long *frames_searched, long *steps, long *restarts);

/* ---- a group of live decoders: N decoders (each: psgpu_mgau_attach, psgpu_device_decode_attach, psgpu_device_search_attach; the same
 * model, dictionary and LM; -fwdflat no), ONE device pipeline in streams mode (psgpu_decode_streams_*, member 0's object).  The
 * application drives every decoder with the UNMODIFIED calls -- ps_start_utt, ps_process_raw(..., FALSE, FALSE) as its audio arrives,
 * ps_get_hyp / ps_seg_iter, ps_end_utt -- and calls psgpu_live_group_step after a round of ps_process_raw calls: one launch set hands
 * over every member's new frames (at most max_step_frames a member a launch set: more are handed over in several) and lets all
 * searches go on; the members' read-outs then return what the CPU decoder would at that point.  (A read-out of a member that has
 * frames the device has not seen, and a member's ps_end_utt, step the group themselves.)  max_frames: the longest utterance. */
typedef struct psgpu_live_group_s psgpu_live_group_t;
psgpu_live_group_t *psgpu_live_group_create(psgpu_device_decode_t *const *members, int n, int max_frames, int max_step_frames);
int psgpu_live_group_step(psgpu_live_group_t *g);
void psgpu_live_group_free(psgpu_live_group_t *g);
/* frames the device searches have stepped through since the group was created (returned) and launch sets so far */
long psgpu_live_group_stats(psgpu_live_group_t *g, long *steps);

/* = ps_start_utt; ps_process_raw(pcm, n, FALSE, TRUE); ps_end_utt -- with front end, features, senone
 * scores, phone loop and lexicon-tree search on the device; afterwards the decoder's back-pointer
 * table, score stack and frame marks hold the result in the reference's layout, so ps_get_hyp(),
 * ps_seg_iter() etc. work as after a host decode.  Returns the number of frames searched, or -1.
 * (Not to be mixed with psgpu_device_search_attach on the same decoder.) */
int psgpu_device_decode_utt(psgpu_device_decode_t *d, int16 const *pcm, size_t n_samples);

/* B utterances through ONE launch set (front end ... search for the whole batch); then, per utterance,
 * _select(u) does the reference's start/end-of-utterance housekeeping and puts utterance u's tables into the
 * decoder, after which ps_get_hyp() / ps_seg_iter() / ps_get_lattice() read them.  Every utterance is decoded
 * from the state a decoder has after ps_start_stream() on its first utterance, so results do not depend on B
 * or on the order.  _run returns 0 or -1; _select the number of frames searched, or -1. */
int psgpu_device_decode_batch_run(psgpu_device_decode_t *d, const int16 *const pcm[], const size_t n[], int B);
int psgpu_device_decode_batch_select(psgpu_device_decode_t *d, int u);
/* ... into another decoder of the same configuration, on the caller's thread (psgpu_decode_batch's workers: the read-outs of a batch --
 * with -bestpath yes each one builds a lattice and searches it -- side by side).  st: the caller's staging buffers (zero-initialised,
 * grown as needed, psgpu_dd_stage_release); stream: the caller's (NULL: the default stream) */
typedef struct psgpu_dd_stage_s { int32_t *bp, *bss, *idx; size_t cap_bp, cap_bss, cap_idx; } psgpu_dd_stage_t;
int psgpu_device_decode_batch_select_into(psgpu_device_decode_t *d, int u, ps_decoder_t *ps, psgpu_dd_stage_t *st, void *stream);
void psgpu_dd_stage_release(psgpu_dd_stage_t *st);
/* frames the front end produced for utterance u of the last run */
int psgpu_device_decode_batch_n_frames(psgpu_device_decode_t *d, int u);

#ifdef __cplusplus
}
#endif
#endif
