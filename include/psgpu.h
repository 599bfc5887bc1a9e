/* psgpu.h -- C ABI of the MI355X-native acoustic-scoring / Viterbi engine.
 *
 * This is the drop-in boundary for PocketSphinx's one data-parallel hot path
 * (SURVEY.md section 8b).  Plain pointers and sizes only; no reference types,
 * no torch types.  Each entry point names the reference interface it
 * replaces (paths relative to the reference's src/).  The reference-side
 * binding that plugs these into `ps_mgau_t` / `acmod_t` is shown in
 * INTEGRATION.md and lives in integration/.
 *
 * Conventions
 *  - every function returns 0 on success and a negative PSGPU_E* code on
 *    failure (the reference's convention: `int`, negative = failure,
 *    acmod.h:98-111); psgpu_last_error() gives the message.
 *  - `*_dev` arguments are device (HBM) pointers; everything else is host.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *  - there is NO CPU fallback: without a usable gfx950 device every call
 *    fails with PSGPU_ENODEV.
 */
#ifndef PSGPU_H
#define PSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSGPU_OK        0
#define PSGPU_ENODEV   -1   /* no HIP device / wrong architecture */
#define PSGPU_EINVAL   -2   /* bad argument / unsupported model shape */
#define PSGPU_ENOMEM   -3   /* device or host allocation failed */
#define PSGPU_EHIP     -4   /* a HIP runtime call failed */
#define PSGPU_ESTATE   -5   /* call sequence violates the frame_eval contract */

#define PSGPU_MAX_TOPN  8

/* ---- library / device ------------------------------------------------ */
/* The ABI's version: major << 16 | minor.  A binding compiled against this header checks
 * `psgpu_abi_version() >> 16 == PSGPU_ABI_VERSION >> 16` (entry points are only ever added within a major version; the
 * minor counts the additions) and asks psgpu_capabilities() for what the loaded library serves before it relies on an
 * optional part -- the reference has the same two-step in its own plugin loading: a name in the vtable, then the calls
 * (acmod.h:98-116). */
#define PSGPU_ABI_VERSION  ((1 << 16) | 6)
#define PSGPU_CAP_PTM                 (1ull << 0)   /* psgpu_ptm_* (ptm_mgau.c) */
#define PSGPU_CAP_SEMI                (1ull << 1)   /* psgpu_semi_* (s2_semi_mgau.c) */
#define PSGPU_CAP_MS                  (1ull << 2)   /* psgpu_ms_* (ms_mgau.c, ms_gauden.c, ms_senone.c) */
#define PSGPU_CAP_HMM                 (1ull << 3)   /* psgpu_hmm_* (hmm.c) */
#define PSGPU_CAP_FE                  (1ull << 4)   /* psgpu_fe_*, psgpu_feat_* (fe/, feat/) */
#define PSGPU_CAP_FWDTREE             (1ull << 5)   /* psgpu_fwdtree_* (ngram_search_fwdtree.c) */
#define PSGPU_CAP_FWDFLAT             (1ull << 6)   /* psgpu_fwdflat_* (ngram_search_fwdflat.c) */
#define PSGPU_CAP_TRIE_LM             (1ull << 7)   /* psgpu_lm_* (lm/lm_trie.c) */
#define PSGPU_CAP_DECODE              (1ull << 8)   /* psgpu_decode_* pipeline objects */
#define PSGPU_CAP_STREAMS             (1ull << 9)   /* psgpu_decode_streams_* / live steps from feature vectors */
#define PSGPU_CAP_PTM_BATCH_ANY_SHAPE (1ull << 10)  /* psgpu_ptm_score_batch_dev for every shape psgpu_ptm_frame_eval serves */
#define PSGPU_CAP_STREAMS_PCM         (1ull << 11)  /* streams fed with PCM (per-stream front-end state + live CMN on the device) */
#define PSGPU_CAP_FWDTREE_SPLIT       (1ull << 12)  /* one utterance's tree search on several workgroups */
#define PSGPU_CAP_LM_SETS             (1ull << 14)  /* class words in a trie model; psgpu_lm_create_interp: an interpolated model set */
#define PSGPU_CAP_FEAT_TYPES          (1ull << 13)  /* psgpu_feat_create: every feature type of feat_init, -lda, -svspec, varnorm, agc max */
int32_t psgpu_abi_version(void);
uint64_t psgpu_capabilities(void);
const char *psgpu_version(void);
const char *psgpu_last_error(void);          /* thread-local message */
int psgpu_device_count(void);                /* >=0, or PSGPU_ENODEV */
int psgpu_set_device(int device);
int psgpu_get_device(void);                  /* the calling thread's current device, or PSGPU_ENODEV */
/* device memory helpers so that a C host needs no HIP headers */
int psgpu_malloc(void **dev_ptr, size_t bytes);
int psgpu_free(void *dev_ptr);
/* page-locked host memory: asynchronous copies into it do not go through a shared staging buffer */
int psgpu_host_alloc(void **host_ptr, size_t bytes);
int psgpu_host_free(void *host_ptr);
int psgpu_memcpy_h2d(void *dst_dev, const void *src, size_t bytes, void *stream);
int psgpu_memcpy_d2h(void *dst, const void *src_dev, size_t bytes, void *stream);
int psgpu_stream_sync(void *stream);

/* ---- PTM acoustic model ------------------------------------------------
 * Replaces the table set built by ptm_mgau_init() (ptm_mgau.c:804-896):
 *   mean/var  packed [n_mgau][n_feat][n_density][featlen[f]] floats, `var`
 *             already precomputed by gauden_dist_precompute (ms_gauden.c:263-308)
 *   det       [n_mgau][n_feat][n_density]
 *   mixw      [n_feat][n_density][n_sen] uint8 (read_sendump / read_mixw,
 *             ptm_mgau.c:456-774); 4-bit clustered sendumps are not accepted
 *   sen2cb    [n_sen] (ptm_mgau.c:877-879)
 *   logadd8   the uint8 table of logmath_init(base, SENSCR_SHIFT, 1)
 *             (ptm_mgau.c:817-825, util/logmath.c:62-162), logadd8_size >= 256
 * The tables are copied to the current device.  Accepted shapes: n_density <= 256,
 * topn 1..8, stream lengths summing to <= 64 -- by psgpu_ptm_frame_eval and by the
 * batched entries below alike (PSGPU_CAP_PTM_BATCH_ANY_SHAPE).  128 densities / top-4 /
 * 13-dim streams (en-us) run the specialised kernels (frames on lanes, closed-form top-N
 * with an exact fix-up); every other shape runs the exact sequential procedure, one
 * wavefront per (utterance, chain) (ptm_batch_topn_generic). */
typedef struct psgpu_ptm_model_s psgpu_ptm_model_t;

int psgpu_ptm_model_create(psgpu_ptm_model_t **out,
                           int32_t n_mgau, int32_t n_feat, int32_t n_density,
                           const int32_t *featlen, int32_t n_sen, int32_t topn,
                           int32_t ds_ratio,
                           const float *mean, const float *var, const float *det,
                           const uint8_t *mixw, const uint8_t *sen2cb,
                           const uint8_t *logadd8, int32_t logadd8_size);
void psgpu_ptm_model_free(psgpu_ptm_model_t *m);
int32_t psgpu_ptm_n_sen(const psgpu_ptm_model_t *m);
int32_t psgpu_ptm_n_chain(const psgpu_ptm_model_t *m);   /* n_mgau * n_feat */
int32_t psgpu_ptm_veclen(const psgpu_ptm_model_t *m);    /* sum of featlen */
int32_t psgpu_ptm_topn(const psgpu_ptm_model_t *m);

/* ---- batched scoring (the throughput path) ------------------------------
 * Replaces, for every frame of every utterance of a batch, one
 * ptm_mgau_frame_eval(..., compallsen=TRUE) call (ptm_mgau.c:408-454) made in
 * frame order: eval_topn + eval_cb (top-N selection with frame-to-frame
 * seeding, :87-226), ptm_mgau_codebook_norm (:265-295) and
 * ptm_mgau_senone_eval (:326-403).
 *
 *  feats_dev     [total_frames][veclen] fp32, utterances back to back
 *  utt_off_dev   [n_utt + 1] int32 frame offsets (utt u = frames
 *                utt_off[u] .. utt_off[u+1]-1); total_frames = utt_off[n_utt]
 *  seed_in_dev   NULL, or [n_utt][n_chain][topn] uint8: the top-N codeword
 *                lists carried into frame 0 of each utterance (the reference
 *                never resets them between utterances, acmod.c:407-421;
 *                SURVEY F7).  NULL = fresh state, cw = 0..topn-1
 *                (ptm_mgau_reset_fast_hist, ptm_mgau.c:777-802).
 *  seed_out_dev  NULL, or [n_utt][n_chain][topn] uint8: receives the lists
 *                carried out of each non-empty utterance's last frame.  Must
 *                not alias seed_in_dev.
 *  topn_score_dev [n_chain][total_frames][topn] int32  raw (pre-norm) scores
 *  topn_cw_dev    [n_chain][total_frames][topn] uint8  codewords
 *                CHAIN-MAJOR (a wavefront owns one chain x 64 frames and stores
 *                1 KB contiguous); both are outputs AND the workspace between
 *                the kernels.  The host-buffer wrapper below returns them
 *                frame-major, [total_frames][n_chain][topn].
 *  senscr_dev    [total_frames][n_sen] int16 senone scores, or NULL to stop
 *                after top-N selection.
 *  flags         PSGPU_PTM_RAW_SCORES: skip the final per-frame
 *                best-score subtraction (ptm_mgau.c:398-400) and store the
 *                un-normalised sums; the per-frame minimum goes to
 *                best_dev[total_frames] (int32) if non-NULL.
 */
#define PSGPU_PTM_RAW_SCORES 1u

int psgpu_ptm_score_batch_dev(psgpu_ptm_model_t *m,
                              const float *feats_dev, const int32_t *utt_off_dev,
                              int32_t n_utt, int32_t total_frames,
                              const uint8_t *seed_in_dev, uint8_t *seed_out_dev,
                              int32_t *topn_score_dev, uint8_t *topn_cw_dev,
                              int16_t *senscr_dev, int32_t *best_dev,
                              uint32_t flags, void *stream);

/* Host-buffer convenience wrapper around the call above (allocates device
 * buffers, copies in, runs, copies out, synchronises).  Any of the output
 * pointers may be NULL.  seed_cw (NULL or [n_utt][n_chain][topn]) is in/out:
 * read as the carried-in lists, overwritten with the carried-out lists
 * (empty utterances keep theirs). */
int psgpu_ptm_score_batch(psgpu_ptm_model_t *m,
                          const float *feats, const int32_t *utt_off, int32_t n_utt,
                          uint8_t *seed_cw,
                          int32_t *topn_score, uint8_t *topn_cw,
                          int16_t *senscr, int32_t *best, uint32_t flags);

/* Per-kernel timing of psgpu_ptm_score_batch_dev: when enabled, HIP events are
 * recorded on the launch stream around the main top-N kernel, the exact
 * fix-up launch and the senone kernel; psgpu_ptm_last_kernel_ms() waits for the
 * most recent call and returns the three durations (ms).  Generic event
 * helpers follow. */
int psgpu_ptm_kernel_timing(psgpu_ptm_model_t *m, int32_t enable);
int psgpu_ptm_last_kernel_ms(psgpu_ptm_model_t *m, float *ms3);
int psgpu_event_create(void **ev);
int psgpu_event_destroy(void *ev);
int psgpu_event_record(void *ev, void *stream);
int psgpu_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms); /* syncs on stop */

/* Launch only one of the two kernels (used by bench.py to time the dominant
 * kernel in isolation with HIP events; same arguments as the batch call). */
int psgpu_ptm_topn_dev(psgpu_ptm_model_t *m, const float *feats_dev,
                       const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                       const uint8_t *seed_in_dev, uint8_t *seed_out_dev,
                       int32_t *topn_score_dev, uint8_t *topn_cw_dev, void *stream);
int psgpu_ptm_senone_dev(psgpu_ptm_model_t *m, int32_t total_frames,
                         const int32_t *topn_score_dev, const uint8_t *topn_cw_dev,
                         int16_t *senscr_dev, int32_t *best_dev, uint32_t flags,
                         void *stream);

/* ---- MFCC front end (caller side of the scorer) ------------------
 * Replaces fe_start_utt + fe_process_frames(all samples) + fe_end_utt
 * (fe/fe_interface.c:318-327, 352-495, 526-541) as acmod_process_full_raw runs them
 * (acmod.c:552-557), i.e. fe_process_utt (fe_interface.c:505-524) plus the tail
 * frame, for batches of whole utterances; floating-point build of the reference
 * (fe/fe_type.h:58-60).  The object holds the reference's own precomputed tables
 * (fe_t / melfb_t, fe/fe_internal.h:70-161), which the caller reads out of its
 * fe_t; nothing is regenerated on the device.  Bit-identical to the reference up
 * to the double-precision log() of the mel spectrum (see csrc/psgpu_fe.hip). */
typedef struct psgpu_fe_s psgpu_fe_t;
typedef struct psgpu_fe_params_s {
    int32_t frame_size, frame_shift, fft_size;   /* fe_t.frame_size, frame_shift, fft_size */
    int32_t n_filt, num_cepstra, out_dim;        /* melfb_t.num_filters, fe_t.num_cepstra, feature_dimension */
    int32_t transform;                           /* fe_t.transform: 0 legacy, 1 dct, 2 htk */
    int32_t log_spec;                            /* fe_t.log_spec: 0, 1 raw, 2 smooth */
    int32_t remove_dc, remove_noise;             /* fe_t.remove_dc, fe_t.noise_stats != NULL */
    int32_t swap, dither;                        /* fe_t.swap, fe_t.dither */
    float alpha;                                 /* fe_t.pre_emphasis_alpha */
    float sqrt_inv_n, sqrt_inv_2n;               /* melfb_t.sqrt_inv_n, sqrt_inv_2n */
    int32_t dither_seed;                         /* fe_t.dither_seed (the -seed option; -1 by default).  With dither on every sample the
                                                  * front end consumes gets (s3_rand_int31() % 4 == 0) added, int16 arithmetic
                                                  * (fe_read_frame_int16 / fe_shift_frame_int16, fe_sigproc.c:868-870, :898-901), from
                                                  * ONE generator seeded at fe_init (fe_init_dither, fe_interface.c:311-316) and never
                                                  * again: the object draws for its utterances in the order it is given them, call after
                                                  * call, as one decoder of the reference does */
} psgpu_fe_params_t;

/* hamming [frame_size/2], ccc/sss [fft_size/4] (float64); spec_start / filt_start /
 * filt_width [n_filt]; filt_coeffs flattened (sum of widths); mel_cosine
 * [num_cepstra][n_filt]; lifter [num_cepstra] or NULL when lifter_val == 0. */
int psgpu_fe_create(psgpu_fe_t **out, const psgpu_fe_params_t *p, const double *hamming,
                    const double *ccc, const double *sss, const int16_t *spec_start,
                    const int16_t *filt_start, const int16_t *filt_width, const float *filt_coeffs,
                    const float *mel_cosine, const float *lifter);
void psgpu_fe_free(psgpu_fe_t *fe);
int32_t psgpu_fe_out_dim(const psgpu_fe_t *fe);
/* frames produced for an utterance of n_samples: full frames + the zero-padded tail
 * frame of fe_end_utt (always present when n_samples > 0) */
int64_t psgpu_fe_n_frames(const psgpu_fe_t *fe, int64_t n_samples);

/* The front end's only libm call is log(mel spectrum + 1e-4) (fe_sigproc.c:1215-1228), double precision; everything else on
 * the path is +, -, *, / in the reference's order and so bit-identical by construction.  out_dev[i] = log(x_dev[i]) as the
 * device computes it there -- for tests/test_fe_gpu.py, which holds it against the host's libm over the mel spectrum's
 * value range (2^24 samples and every value of the goldens). */
int psgpu_fe_log_dev(const double *x_dev, int64_t n, double *out_dev, void *stream);

/* pcm_dev: samples of n_utt utterances back to back; samp_off [n_utt + 1] HOST array of
 * sample offsets.  cep_dev [total_frames][out_dim]; frame_off_dev [n_utt + 1] receives the
 * frame offsets (the utt_off_dev of psgpu_feat_1s_c_d_dd_dev / psgpu_ptm_score_batch_dev),
 * frame_off (host, optional) the same.  noise_dev [n_utt][4][n_filt] float64 (power, noise,
 * floor, peak of noise_stats_t, fe_noise.c:70-101) and undefined_dev [n_utt] are the noise
 * tracker carried into and out of each utterance (the reference keeps it per decoder until
 * ps_start_stream); both NULL = every utterance starts from reset statistics.  Asynchronous
 * on `stream`; one call in flight per front-end object (it owns the scratch spectrum). */
int psgpu_fe_process_utts_dev(psgpu_fe_t *fe, const int16_t *pcm_dev, const int64_t *samp_off, int32_t n_utt,
                              double *noise_dev, int32_t *undefined_dev, float *cep_dev,
                              int32_t *frame_off_dev, int32_t *frame_off, void *stream);
/* host buffers, synchronous */
int psgpu_fe_process_utts(psgpu_fe_t *fe, const int16_t *pcm, const int64_t *samp_off, int32_t n_utt,
                          double *noise, int32_t *undefined, float *cep, int32_t *frame_off);

/* ---- dynamic features (caller side of the scorer) -------------------------------
 * Replaces feat_s2mfc2feat_live(begin = end = TRUE) (feat/feat.c:1310, :1275-1306)
 * for the "1s_c_d_dd" feature type with batch CMN and no AGC / LDA (the en-us
 * configuration, model/en-us/en-us/feat.params): cmn() (feat/cmn.c:166-208),
 * first/last frame replicated over a window of 3, feat_1s_c_d_dd_cep2feat
 * (feat.c:579-622).  cep [total_frames][cepsize] MFCC vectors of n_utt utterances
 * back to back (utt_off as above; NOT modified, unlike the reference, which
 * normalises its input in place); feat [total_frames][3*cepsize], directly usable
 * as feats_dev of psgpu_ptm_score_batch_dev.  Bit-exact. */
int psgpu_feat_1s_c_d_dd_dev(const float *cep_dev, const int32_t *utt_off_dev, int32_t n_utt,
                             int32_t cepsize, float *feat_dev, void *stream);
int psgpu_feat_1s_c_d_dd(const float *cep, const int32_t *utt_off, int32_t n_utt, int32_t cepsize,
                         float *feat);

/* ---- every feature type of feat_init (feat.c:704-915) for whole utterances: s2_4x, s3_1x39 / 1s_12c_12d_3p_12dd, 1s_c_d_dd,
 * 1s_c_d_ld_dd, cep_dcep / 1s_c_d, cep / 1s_c, 1s_3c / 1s_4c and the generic "%d,%d,...[:window]", with cmn() (cmn: 0 none, 1 batch --
 * the reference's "current" / "batch"; varnorm: unit variance, cmn.c:203-231), agc_max (agc: 0 none, 1 max; agc.c:110-127), the
 * transform of -lda (lda [lda_out][lda_in] as feat_read_lda keeps it, lda.c:63-159) and the subvector projection of -svspec (subvec
 * [n_subvec]: the components in the specification's order, feat.c:333-352).  psgpu_feat_compute_dev: cep_dev [total][cepsize] of
 * n_utt utterances back to back (utt_off_dev [n_utt + 1]) -> feat_dev [total][psgpu_feat_out_dim].  Bit-exact against
 * feat_s2mfc2feat_live(begin = end = TRUE).  (PSGPU_CAP_FEAT_TYPES.) */
typedef struct psgpu_feat_s psgpu_feat_t;
int psgpu_feat_create(psgpu_feat_t **out, const char *type, int32_t cepsize, int32_t cmn, int32_t varnorm, int32_t agc, const float *lda,
                      int32_t lda_out, int32_t lda_in, const int32_t *subvec, int32_t n_subvec);
void psgpu_feat_free(psgpu_feat_t *f);
int32_t psgpu_feat_out_dim(const psgpu_feat_t *f);
int32_t psgpu_feat_cepsize(const psgpu_feat_t *f);
int32_t psgpu_feat_window(const psgpu_feat_t *f);
int psgpu_feat_compute_dev(const psgpu_feat_t *f, const float *cep_dev, const int32_t *utt_off_dev, int32_t n_utt, float *feat_dev, void *stream);
int psgpu_feat_compute(const psgpu_feat_t *f, const float *cep, const int32_t *utt_off, int32_t n_utt, float *feat);

/* ---- the same two stages for live decoders: audio and cepstra arrive in pieces ---------------------------------
 * psgpu_fe_stream_step_dev: fe_process_frames on a piece of audio per stream (fe_interface.c:352-512; + fe_end_utt, :514-533,
 * when a stream's utterance ends): work_dev holds, per stream u, the samples its earlier steps left unframed followed by the new
 * ones at work_dev[samp[2u]] (samp[2u + 1] of them) with the pre-emphasis prior in the slot before; n_frames[u] frames are made of
 * them (full frames at multiples of frame_shift; a last frame shorter than frame_size is zero-padded); the noise tracker's state
 * per stream goes on in noise_dev [n][4][n_filt] / undefined_dev [n] as in psgpu_fe_process_utts_dev.  Cepstra back to back in
 * cep_dev, their offsets in frame_off_dev [n + 1].  No dither.
 * psgpu_feat_live_step_dev: feat_s2mfc2feat_live piece by piece (feat.c:1310-1420) with cmn_live's running mean (cmn_live.c:65-150) for
 * "1s_c_d_dd": per stream a list of ops (n cepstra, flags: 1 beginutt, 2 endutt, 4 feat_update_stats only) = the calls the
 * reference's acmod makes (ops_dev [..][2], op_off_dev [n + 1]); state_dev [n][psgpu_feat_live_state_words(cepsize)] carries mean, sum,
 * frame count and the feature window between steps (psgpu_feat_live_state_init: a new decoder's, the mean from -cmninit); the
 * feature frames the ops release go to feat_dev at row feat_off_dev[u].  Bit-exact against the reference's chunked
 * acmod_process_raw (oracle/ref_dump.c livefeat).  psgpu_decode_streams_step_pcm drives both. */
int psgpu_fe_stream_step_dev(psgpu_fe_t *fe, const int16_t *work_dev, const int64_t *samp, const int32_t *n_frames, int32_t n_streams,
                             double *noise_dev, int32_t *undefined_dev, float *cep_dev, int32_t *frame_off_dev, void *stream);
int32_t psgpu_fe_frame_size(const psgpu_fe_t *fe);
int32_t psgpu_fe_frame_shift(const psgpu_fe_t *fe);
int32_t psgpu_feat_live_state_words(int32_t cepsize);
int psgpu_feat_live_state_init(float *state_host, int32_t cepsize, const float *cmninit, int32_t n_init);
int psgpu_feat_live_step_dev(const float *cep_dev, const int32_t *cep_off_dev, const int32_t *ops_dev, const int32_t *op_off_dev,
                             const int32_t *feat_off_dev, int32_t n_streams, int32_t cepsize, float *state_dev, float *feat_dev, void *stream);

/* ---- per-call scoring state: the ps_mgau_t::frame_eval replacement -------
 * One object per decoder.  Replaces the mutable part of ptm_mgau_t
 * (ptm_mgau.h:68-97): the history ring hist[n_fast_hist] of top-N lists and
 * active-codebook sets (ptm_mgau.c:884-890), kept in HBM.
 *
 * psgpu_ptm_frame_eval() is ptm_mgau_frame_eval() (ptm_mgau.c:408-454) with
 * the reference's exact call contract (acmod.h:98-111):
 *   senscr          [n_sen] int16 out (host), fully written: listed senones
 *                   get their score, every other entry becomes -best
 *                   (ptm_mgau.c:398-400)
 *   senone_active   uint8 DELTAS as built by acmod_flags2list
 *                   (acmod.c:1223-1275); ignored when compallsen != 0
 *   feat            the frame's feature vector, streams concatenated
 *                   (veclen floats; the reference's mfcc_t** points at the
 *                   same contiguous floats, feat/feat.c:356-384)
 *   frame           frame number; slot = frame % n_fast_hist
 *   frame_idx       the caller's ps_mgau_t.frame_idx (acmod.h:113-116, written
 *                   by acmod_start_utt / acmod_advance / acmod_rewind,
 *                   acmod.c:419,862,874): codebooks are evaluated only when
 *                   frame >= frame_idx, otherwise the slot is reused.
 * The call is synchronous (returns with senscr filled).
 * psgpu_ptm_state_reset() is ptm_mgau_reset_fast_hist() (ptm_mgau.c:777-802).
 * psgpu_ptm_state_get_topn() copies one slot's lists out ([n_chain][topn]
 * int32 each, mgau_active one byte per codebook, any may be NULL; slot -1 =
 * the slot of the last call) -- the reference exposes the same data as
 * s->f->topn / s->f->mgau_active. */
typedef struct psgpu_ptm_state_s psgpu_ptm_state_t;

int psgpu_ptm_state_create(psgpu_ptm_state_t **out, psgpu_ptm_model_t *m, int32_t n_fast_hist);
void psgpu_ptm_state_free(psgpu_ptm_state_t *s);
int psgpu_ptm_state_reset(psgpu_ptm_state_t *s);
int psgpu_ptm_frame_eval(psgpu_ptm_state_t *s, int16_t *senscr,
                         const uint8_t *senone_active, int32_t n_senone_active,
                         const float *feat, int32_t frame, int32_t frame_idx,
                         int32_t compallsen);
int psgpu_ptm_state_get_topn(psgpu_ptm_state_t *s, int32_t slot, int32_t *cw, int32_t *score,
                             uint8_t *mgau_active);
/* Look-ahead (additive; the reference scores one frame per call).  Announces
 * the feature vectors of frames frame0 .. frame0+n_frames-1 ([n_frames][veclen])
 * that later psgpu_ptm_frame_eval calls will present -- in full-utterance decoding
 * acmod holds them all before the search starts (acmod.c:496-528).  When the first
 * of them is asked for as a fresh evaluation with every codebook active, all of
 * them are scored in ONE batched pass (the kernels of psgpu_ptm_score_batch_dev,
 * seeded with the ring) and the un-normalised rows come to the host; each later
 * call whose frame, feature vector and call pattern match is then answered from
 * that row (score - min over the active list, ptm_mgau.c:393-400) with no device
 * work.  Anything else -- a codebook subset on a fresh call (pass 2), frames out of
 * order, a different vector -- falls back to the per-call kernels after the ring has
 * been brought up to date, so results are identical either way.  A call with
 * n_frames == 0 drops the cache.  Models outside the batched kernels' shape ignore it. */
int psgpu_ptm_state_lookahead(psgpu_ptm_state_t *s, const float *feats, int32_t frame0, int32_t n_frames);
int psgpu_ptm_state_lookahead_stats(psgpu_ptm_state_t *s, int64_t *calls_served, int64_t *batches);
/* For a search component that lives on the device as well (psgpu_phone_loop_run_dev): the
 * cache's un-normalised rows [n_frames][n_sen] and all-senone minima [n_frames] on the device
 * (computing them now if the announcement has not been used yet; PSGPU_ESTATE when there is no
 * cache positioned at its first frame), and the bookkeeping of a fresh all-codebook
 * frame_eval call that such a component no longer makes: psgpu_ptm_state_mark_fresh(frame)
 * is that call without its scores (PSGPU_ESTATE unless `frame` is the cache's next fresh frame). */
int psgpu_ptm_state_lookahead_rows(psgpu_ptm_state_t *s, const int16_t **raw_dev, const int32_t **best_dev,
                                   int32_t *frame0, int32_t *n_frames);
int psgpu_ptm_state_mark_fresh(psgpu_ptm_state_t *s, int32_t frame);
/* Load one slot of the ring from a host image of ptm_fast_eval_t
 * (ptm_mgau.h:68-71): cw/score [n_chain][topn] int32, mgau_active one byte
 * per codebook (NULL = all active).  Lets a shim attached to a decoder that
 * has already decoded with the CPU scorer continue from its exact state. */
int psgpu_ptm_state_set_topn(psgpu_ptm_state_t *s, int32_t slot, const int32_t *cw,
                             const int32_t *score, const uint8_t *mgau_active);

/* ---- semi-continuous scorer ("s2_semi") -------------------------------------
 * Replaces s2_semi_mgau_frame_eval() (s2_semi_mgau.c:836-883) with the same
 * call contract as psgpu_ptm_frame_eval above.  Model tables as
 * s2_semi_mgau_init() holds them (:1235-1332):
 *   mean/var  packed [n_feat][n_density][featlen[f]] floats (one shared
 *             codebook, var precomputed by gauden_dist_precompute)
 *   det       [n_feat][n_density]
 *   mixw      [n_feat][n_density][n_sen] uint8, or, when mixw_cb != NULL,
 *             4-bit clustered: [n_feat][n_density][(n_sen+1)/2] nibbles indexing
 *             the 16-entry mixw_cb (read_sendump, :885-1080)
 *   topn_beam [n_feat] per-stream top-N beam, 0 = none (:1296-1299), or NULL
 * State = the ring topn_hist[n_topn_hist] / topn_hist_n (s2_semi_mgau.h:83-86). */
typedef struct psgpu_semi_model_s psgpu_semi_model_t;
typedef struct psgpu_semi_state_s psgpu_semi_state_t;

int psgpu_semi_model_create(psgpu_semi_model_t **out, int32_t n_feat, int32_t n_density,
                            const int32_t *featlen, int32_t n_sen, int32_t topn, int32_t ds_ratio,
                            const uint8_t *topn_beam,
                            const float *mean, const float *var, const float *det,
                            const uint8_t *mixw, const uint8_t *mixw_cb,
                            const uint8_t *logadd8, int32_t logadd8_size);
void psgpu_semi_model_free(psgpu_semi_model_t *m);
int psgpu_semi_state_create(psgpu_semi_state_t **out, psgpu_semi_model_t *m, int32_t n_topn_hist);
void psgpu_semi_state_free(psgpu_semi_state_t *s);
int psgpu_semi_state_reset(psgpu_semi_state_t *s);
int psgpu_semi_frame_eval(psgpu_semi_state_t *s, int16_t *senscr,
                          const uint8_t *senone_active, int32_t n_senone_active,
                          const float *feat, int32_t frame, int32_t frame_idx,
                          int32_t compallsen);
/* one slot of the ring: cw/score [n_feat][topn] int32, n_used [n_feat] int32
 * (topn_hist_n); slot -1 = the slot of the last call */
/* Batched compallsen scoring of whole utterances (additive, like psgpu_ptm_score_batch_dev):
 * feats [total_frames][veclen], utt_off [n_utt + 1] frame offsets, senscr [total_frames][n_sen].
 * = one s2_semi_mgau_frame_eval(compallsen) per frame in frame order, every utterance starting
 * from the lists of a freshly initialised scorer, frames numbered from 0 within the utterance
 * (down-sampling rule, s2_semi_mgau.c:173-175).  The frame-to-frame dependence of the top-N
 * lists is kept: one wavefront per (utterance, stream) walks its frames in order. */
int psgpu_semi_score_batch_dev(psgpu_semi_model_t *m, const float *feats_dev, const int32_t *utt_off_dev,
                               int32_t n_utt, int32_t total_frames, int16_t *senscr_dev, void *stream);
int psgpu_semi_score_batch(psgpu_semi_model_t *m, const float *feats, const int32_t *utt_off, int32_t n_utt,
                           int16_t *senscr);
/* ... with the top-N lists carried across calls, as s2_semi_mgau_frame_eval carries them from frame to frame and from an utterance's
 * last frames to the next one's first (its ring topn_hist[n_topn_hist], n_topn_hist = pl_window + 2, s2_semi_mgau.c:853-860, :1301-1322):
 *  seed_in_dev  [n_utt][n_feat][topn] uint8 codewords every utterance's first frame starts from (NULL: a new scorer's, codeword = rank)
 *  seed_out_dev the lists of every utterance's last frame (the next call's seed_in when the utterance goes on; untouched for an
 *               utterance without frames; NULL: not wanted)
 *  slot_out_dev the lists of every utterance's last frame ts with (frame_base + ts) % n_hist == n_hist - 1 -- ring slot n_hist - 1,
 *               which seeds the NEXT utterance's first frame; untouched when there is no such frame (NULL: not wanted)
 *  frame_base_dev [n_utt] int32 frames of each utterance scored by earlier calls (frame numbers go on from there: the ring's
 *               slots, the down-sampling rule -ds; NULL: 0).  The three list buffers must be distinct. */
int psgpu_semi_score_batch_carry_dev(psgpu_semi_model_t *m, const float *feats_dev, const int32_t *utt_off_dev, int32_t n_utt,
                                     int32_t total_frames, const uint8_t *seed_in_dev, uint8_t *seed_out_dev, uint8_t *slot_out_dev,
                                     int32_t n_hist, const int32_t *frame_base_dev, int16_t *senscr_dev, void *stream);
int32_t psgpu_semi_n_feat(const psgpu_semi_model_t *m);
int32_t psgpu_semi_topn(const psgpu_semi_model_t *m);
int32_t psgpu_semi_n_sen(const psgpu_semi_model_t *m);
int32_t psgpu_semi_veclen(const psgpu_semi_model_t *m);    /* sum of featlen */
int psgpu_semi_state_get_topn(psgpu_semi_state_t *s, int32_t slot, int32_t *cw, int32_t *score,
                              int32_t *n_used);
int psgpu_semi_state_set_topn(psgpu_semi_state_t *s, int32_t slot, const int32_t *cw,
                              const int32_t *score, const int32_t *n_used);

/* ---- multi-stream / continuous scorer ("ms") ----------------------------------
 * Replaces ms_cont_mgau_frame_eval() (ms_mgau.c:191-282): gauden_dist() /
 * compute_dist() (ms_gauden.c:424-509) + senone_eval() (ms_senone.c:357-407).
 * Tables as ms_mgau_init() holds them (ms_mgau.c:80-160):
 *   mean/var  packed [n_mgau][n_feat][n_density][featlen[f]], var precomputed
 *   det       [n_mgau][n_feat][n_density]
 *   pdf       senone weights (senprob_t) in the canonical order
 *             [n_sen][n_feat][n_density]; the reference keeps [feat][cw][sen] when
 *             there is a single codebook (ms_senone.c:198-209) -- transpose first
 *   sen2mgau  [n_sen] senone_t.mgau (ms_senone.c:283-320)
 *   logadd    add table of senone_t.lmath (= logmath_init(base, SENSCR_SHIFT, 1),
 *             ms_senone.c:276): logadd_size entries of logadd_width (1|2|4) bytes;
 *             log_zero = logmath_get_zero() of the same object
 *   topn      msg->topn (already clamped to n_density, ms_mgau.c:141-147); aw = senone_t.aw
 * The model object also carries the per-decoder state of the per-call entry
 * (msg->dist: list ids persist between calls, ms_gauden.c:438-440).
 *
 * psgpu_ms_frame_eval: senscr is IN/OUT -- only listed senones are written,
 * exactly as the reference leaves stale values in unlisted entries.  There is no
 * frame / frame_idx: the scorer is stateless in time (ms_mgau.c:207). */
typedef struct psgpu_ms_model_s psgpu_ms_model_t;

int psgpu_ms_model_create(psgpu_ms_model_t **out, int32_t n_mgau, int32_t n_feat, int32_t n_density,
                          const int32_t *featlen, int32_t n_sen, int32_t topn, int32_t aw,
                          const float *mean, const float *var, const float *det,
                          const uint8_t *pdf, const uint32_t *sen2mgau,
                          const void *logadd, int32_t logadd_size, int32_t logadd_width,
                          int32_t log_zero);
void psgpu_ms_model_free(psgpu_ms_model_t *m);
int32_t psgpu_ms_n_sen(const psgpu_ms_model_t *m);
int32_t psgpu_ms_veclen(const psgpu_ms_model_t *m);
int psgpu_ms_frame_eval(psgpu_ms_model_t *m, int16_t *senscr,
                        const uint8_t *senone_active, int32_t n_senone_active,
                        const float *feat, int32_t compallsen);
/* Look-ahead (optional): announce the feature vectors of frames frame0 .. frame0+n_frames-1
 * ([n_frames][veclen]); they are scored in one batched pass, and psgpu_ms_frame_eval_at()
 * answers every later call for one of them -- any pass, any active list, as long as the
 * vector handed in is the announced one -- from the host copy, with the reference's
 * active-list normalisation (ms_mgau.c:219-234) and its list-id side effect
 * (ms_gauden.c:438-440) preserved.  Results are identical with or without it.  n_frames 0
 * drops the cache.  psgpu_ms_frame_eval_at with a frame outside the cache (or -1) is
 * psgpu_ms_frame_eval. */
int psgpu_ms_lookahead(psgpu_ms_model_t *m, const float *feats, int32_t frame0, int32_t n_frames);
int psgpu_ms_lookahead_covers(const psgpu_ms_model_t *m, const float *feat, int32_t frame);
int psgpu_ms_lookahead_stats(const psgpu_ms_model_t *m, int64_t *served, int64_t *batches);
int psgpu_ms_frame_eval_at(psgpu_ms_model_t *m, int16_t *senscr, const uint8_t *senone_active,
                           int32_t n_senone_active, const float *feat, int32_t frame,
                           int32_t compallsen);
/* Batched compallsen scoring of total_frames independent frames (frames of any
 * number of utterances back to back: the scorer has no time dependence).
 *  list_id_dev / list_dist_dev  [n_mgau][n_feat][total_frames][topn] int32 / fp32:
 *                               the top-N lists, codebook-major (output and workspace).
 *                               May both be NULL for a fully continuous model (one
 *                               stream, senone i owns codebook i, topn < n_density):
 *                               its fused kernel goes from the densities to the senone
 *                               score without the lists; PSGPU_EINVAL for other shapes
 *  senscr_dev                   [total_frames][n_sen] int16, or NULL to stop after
 *                               the top-N kernel
 * psgpu_ms_batch_check() synchronises `stream` and returns PSGPU_ESTATE if some
 * frame had fewer than topn densities above WORST_DIST (see above). */
int psgpu_ms_score_batch_dev(psgpu_ms_model_t *m, const float *feats_dev, int32_t total_frames,
                             int32_t *list_id_dev, float *list_dist_dev, int16_t *senscr_dev,
                             void *stream);
int psgpu_ms_batch_check(psgpu_ms_model_t *m, void *stream);
/* The same, stopping after senone_eval's own int16 store (ms_mgau.c:219, :255: `senscr[s] = senone_eval(...)`): the rows a
 * device search normalises itself over ITS list -- best = the minimum over the listed senones, score - best clamped to
 * int16 (:226-234, :269-277), psgpu_fwdtree_search_dev raw_scores = 1 / psgpu_phone_loop_run_dev with a senone list. */
int psgpu_ms_score_batch_raw_dev(psgpu_ms_model_t *m, const float *feats_dev, int32_t total_frames,
                                 int32_t *list_id_dev, float *list_dist_dev, int16_t *senscr_dev, void *stream);
/* whether the batch entries need the list buffers (anything but a fully continuous model), their entries per frame
 * (n_mgau * n_feat * topn) */
int32_t psgpu_ms_batch_needs_lists(const psgpu_ms_model_t *m);
int32_t psgpu_ms_list_entries_per_frame(const psgpu_ms_model_t *m);
/* host-buffer convenience wrapper (allocates, copies, runs, checks, copies back) */
int psgpu_ms_score_batch(psgpu_ms_model_t *m, const float *feats, int32_t total_frames, int16_t *senscr);

/* ---- HMM Viterbi step ----------------------------------------------------
 * Replaces hmm_vit_eval() (hmm.c:786-805) and its hard-wired variants
 * hmm_vit_eval_3st_lr[_mpx] (:529-707) / _5st_lr[_mpx] (:222-525) for whole
 * active lists, i.e. the loops of evaluate_channels()
 * (ngram_search_fwdtree.c:605-715), fwdflat_eval_chan()
 * (ngram_search_fwdflat.c:444-480) and evaluate_hmms()
 * (phone_loop_search.c:202-222).
 *
 * psgpu_hmm_ctx_t replaces hmm_context_t (hmm.h:145-154): the shared tables
 *   tp    uint8 [n_tmat][n_emit][n_emit+1]  (tmat_t.tp flattened, tmat.h:60-66)
 *   sseq  uint16 [n_sseq][n_emit]           (bin_mdef_t.sseq, bin_mdef.h:119)
 * psgpu_hmm_rec_t carries the per-HMM fields of hmm_t (hmm.h:169-182) in one
 * 64-byte line.  senid[] holds senone ids for a non-multiplex HMM and
 * per-state ssids (BAD_SSID 0xffff = state not yet entered) for a multiplex
 * one, exactly as hmm_t.senid; tmatid_mpx = tmatid | PSGPU_HMM_MPX for
 * multiplex HMMs.  Unused states of a record are ignored.  n_emit_state is
 * 1..5 (HMM_MAX_NSTATE): 3 and 5 run the hard-wired left-to-right forms
 * (hmm.c:222-707), 1, 2 and 4 the any-topology form hmm_vit_eval_anytopo
 * (hmm.c:710-784) -- the dispatch of hmm_vit_eval (hmm.c:786-805).  The phone
 * loop and the searches are built for 3 and 5. */
#define PSGPU_HMM_MPX 0x8000u

typedef struct psgpu_hmm_rec_s {
    int32_t score[5];       /* hmm_t.score      */
    int32_t history[5];     /* hmm_t.history    */
    int32_t out_score;      /* hmm_t.out_score  */
    int32_t out_history;    /* hmm_t.out_history */
    int32_t bestscore;      /* hmm_t.bestscore (written) */
    uint16_t senid[5];      /* hmm_t.senid      */
    uint16_t tmatid_mpx;    /* hmm_t.tmatid | (hmm_t.mpx ? PSGPU_HMM_MPX : 0) */
} psgpu_hmm_rec_t;          /* 64 bytes */

typedef struct psgpu_hmm_ctx_s psgpu_hmm_ctx_t;

int psgpu_hmm_ctx_create(psgpu_hmm_ctx_t **out, int32_t n_emit_state, int32_t n_tmat,
                         const uint8_t *tp, int32_t n_sseq, const uint16_t *sseq, int32_t n_sen);
void psgpu_hmm_ctx_free(psgpu_hmm_ctx_t *c);
int32_t psgpu_hmm_n_emit_state(const psgpu_hmm_ctx_t *c);

/* One Viterbi step for n_active HMMs resident in HBM.
 *  recs_dev        record arena
 *  active_idx_dev  NULL (records 0..n_active-1) or [n_active] indices into the
 *                  arena: the active HMM list of the frame
 *  utt_of_hmm_dev  NULL (one utterance) or [arena] utterance number of every
 *                  record: selects the senone-score row and the best[] slot
 *  senscr_dev      int16 rows of senone scores (acmod_score output,
 *                  acmod.h:165), row u at senscr_dev + u * senscr_stride
 *  best_dev        NULL or int32 [n_utt]: max-folded with every HMM's returned
 *                  best score (the caller presets it, normally to WORST_SCORE
 *                  0xE0000000, hmm.h:84) */
int psgpu_hmm_vit_eval_dev(psgpu_hmm_ctx_t *c, psgpu_hmm_rec_t *recs_dev,
                           const int32_t *active_idx_dev, int32_t n_active,
                           const uint16_t *utt_of_hmm_dev,
                           const int16_t *senscr_dev, int32_t senscr_stride,
                           int32_t *best_dev, void *stream);


/* ---- phone-loop search of whole utterances -----------------------
 * Replaces phone_loop_search_start (phone_loop_search.c:165-184) + phone_loop_search_step
 * (:302-340) for every frame of n_utt utterances: the per-frame normalisation acmod_score
 * applies for the all-phones-active senone list, evaluate_hmms, store_scores, prune_hmms,
 * phone_transition, renormalisation.  One wavefront per utterance, lane = CI phone.
 *  c             tp / sseq tables (psgpu_hmm_ctx_create), 3 or 5 emitting states, non-multiplex
 *  ssid_dev, tmatid_dev [n_phones]   the CI phones' senone-sequence and transition-matrix ids
 *                                    (hmm_init(ctx, hmm, FALSE, pid2ssid, pid2tmatid), :113-117)
 *  raw_dev [total][raw_stride] int16 un-normalised senone scores (PSGPU_PTM_RAW_SCORES rows)
 *  ci_list_dev [n_list]              senone ids of the list acmod_flags2list builds when every CI
 *                                    phone is active (incl. its bridging entries): a frame's scores
 *                                    are raw - min over this list (ptm_mgau.c:393-400); OR
 *  best_dev [total]                  the all-senone minima when the decoder runs with -compallsen
 *  penalties_dev [total][n_phones]   pls->penalties after each step (what fwdtree reads through
 *                                    phone_loop_search_score, phone_loop_search.h:99)
 *  pen_now_dev   [total][n_phones]   the step's own ring entry (pen_buf), and
 *  state_dev     [total][n_phones][8] score[0..4], out_score, bestscore, frame of each HMM after
 *                                    the step: enough to continue stepping on the host from any frame
 *                                    (either may be NULL: a device-only pipeline needs the penalties alone)
 * total_frames = utt_off[n_utt].  Two launches: every frame's normaliser and the CI phones'
 * normalised scores, packed, in parallel; then one wavefront per utterance marching through its
 * frames with the next four frames' scores always in registers. */
typedef struct psgpu_phone_loop_params_s {
    int32_t n_phones;            /* <= 64 */
    int32_t window;              /* pl_window, 1..32 */
    int32_t beam, pbeam, pip;    /* phone_loop_search_t.beam / pbeam / pip */
    double penalty_weight;       /* pl_weight */
} psgpu_phone_loop_params_t;
int psgpu_phone_loop_run_dev(psgpu_hmm_ctx_t *c, const psgpu_phone_loop_params_t *p, const uint16_t *ssid_dev,
                             const int16_t *tmatid_dev, const uint16_t *ci_list_dev, int32_t n_list,
                             const int16_t *raw_dev, int64_t raw_stride, const int32_t *best_dev,
                             const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                             int32_t *penalties_dev, int32_t *pen_now_dev, int32_t *state_dev, void *stream);
/* An utterance IN PROGRESS: phone_loop_search_step is called once per frame and keeps its HMMs, its penalty ring (pen_buf,
 * pen_buf_ptr) and its best score between ps_process_raw calls (phone_loop_search.c:302-340).  As psgpu_phone_loop_run_dev for
 * the frames of ONE call (raw_dev / best_dev / penalties_dev hold this call's frames only: utt_off_dev [n_utt + 1] counts them),
 * with the utterances' state read from (resume != 0) and written to carry_dev [n_utt][psgpu_phone_loop_carry_words()] int32:
 * resume = 0 starts every utterance afresh (phone_loop_search_start).  The penalties of a sequence of calls are those of one
 * call over all the frames.  A call without frames for an utterance leaves its state alone. */
int32_t psgpu_phone_loop_carry_words(void);
/* utterance u of a carry buffer starts afresh at the next call although resume != 0 (the other utterances go on) */
int psgpu_phone_loop_carry_restart(int32_t *carry_dev, int32_t u, void *stream);
int psgpu_phone_loop_run_carry_dev(psgpu_hmm_ctx_t *c, const psgpu_phone_loop_params_t *p, const uint16_t *ssid_dev,
                                   const int16_t *tmatid_dev, const uint16_t *ci_list_dev, int32_t n_list,
                                   const int16_t *raw_dev, int64_t raw_stride, const int32_t *best_dev,
                                   const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                                   int32_t *penalties_dev, int32_t *carry_dev, int32_t resume, void *stream);
/* the context's own (non-blocking) stream: a stream a binding can run one decoder's chain of calls on without
 * serialising with other decoders; usable with psgpu_memcpy_* / psgpu_stream_sync.  (`stream` = NULL in the call
 * above is the default stream, as everywhere.) */
void *psgpu_hmm_ctx_stream(psgpu_hmm_ctx_t *c);

/* ---- lexicon-tree search of whole utterances --------------------------
 * Replaces ngram_fwdtree_start + ngram_fwdtree_search per frame + ngram_fwdtree_finish
 * (ngram_search_fwdtree.c:469-520, 1452-1495, 1497-1533) with the back-pointer helpers of
 * ngram_search.c (:301-498, 583-674).  The tables are the reference's own search structures
 * flattened to index arrays (what oracle/ref_dump.c `fwdtree` writes: the tree
 * create_search_channels built with roots first, single-phone word channels, dictionary and
 * dict2pid tables, `par` = sizes, beams, penalties, special word ids) and the language model
 * either as a dense table lm[w3][w2 + 1][w1 + 1] = ngram_tg_score(w3, w2, w1) >> SENSCR_SHIFT
 * over dictionary word ids (-1 = no history; small vocabularies) or as the model's own trie
 * (psgpu_fwdtree_set_lm).  Any tree size (<= 64 CI phones); per-frame work follows the active
 * channels.  Where the tree-level state fits (en-us with a few hundred words) it is kept in LDS
 * for the whole utterance, otherwise in a per-utterance slab in device memory -- same tables. */
typedef struct psgpu_fwdtree_s psgpu_fwdtree_t;
typedef struct psgpu_fwdtree_tables_s {
    const int32_t *par;                       /* [32] */
    const int32_t *node_ci, *node_ci2, *node_ssid, *node_tmat, *node_child, *node_sib, *node_penult_wid;
    const int32_t *homophone_set;
    const int32_t *w1_wid, *w1_ci, *w1_ci2, *w1_ssid, *w1_tmat, *w1_mpx;
    const int32_t *dict_pronlen, *dict_first, *dict_last, *dict_last2, *dict_basewid, *dict_filler;
    const int32_t *rssid_n, *rssid_ssid, *rssid_cimap, *ldiph_lc;
    const uint8_t *tp;
    const uint16_t *sseq;
    const int32_t *ci_tmat;
    const int32_t *lm;
    int32_t n_tmat, n_sseq;
} psgpu_fwdtree_tables_t;
int psgpu_fwdtree_create(psgpu_fwdtree_t **out, const psgpu_fwdtree_tables_t *t);
void psgpu_fwdtree_free(psgpu_fwdtree_t *m);
/* n_utt utterances, one workgroup each, every frame inside the kernel.  senscr_dev
 * [total][scr_stride] int16 = the scores acmod_score hands the search for each frame;
 * penalties_dev [total][n_ci] = pls->penalties as the search reads them at that frame;
 * utt_off_dev [n_utt + 1].  Per utterance u: the ten columns of the back-pointer table
 * (frame, valid, wid, bp, score, s_idx, real_wid, prev_real_wid, last_phone, last2_phone:
 * bptbl_t, ngram_search.h:112-124) at bp_dev + u*10*bp_cap, the right-context score stack at
 * bss_dev + u*bss_cap, bp_table_idx at idx_dev + u*(max_frames + 2), per-frame
 * {best_score, last_phone_best_score, bpidx, n_active_chan} at step_dev + u*max_frames*4, and
 * result_dev + u*8 = {n back-pointers, score-stack length, frames searched, status (1: a table
 * was full), best_score of the last frame (ngs->best_score), HMM evaluations of the utterance (low, high
 * word), listed senones summed over its frames (raw_scores mode)}.  Asynchronous on `stream`; the work
 * slab belongs to the handle, so one search at a time per handle.
 * raw_scores = 3: as 1, but the rows are final scores -- a scorer that does not normalise over the call's list
 * (s2_semi_mgau_frame_eval): the lists are still built (the result's counters), nothing is subtracted.  With 1 the
 * difference score - minimum is clamped to int16, as ms_cont_mgau_frame_eval stores it (ms_mgau.c:269-277; PTM's
 * differences never leave the range).
 * raw_scores = 1: senscr_dev holds the scorer's UN-normalised rows (PSGPU_PTM_RAW_SCORES) and
 * penalties_dev the phone loop's output per phone-loop frame (psgpu_phone_loop_run_dev): the kernel
 * then builds each frame's active senone list itself (compute_sen_active + acmod_flags2list,
 * bridging entries included), subtracts its minimum as the scorer would (ptm_mgau.c:393-400), and
 * reads the penalties of frame min(f + pl_window, T - 1) -- i.e. it is fed directly by the other
 * kernels, nothing passes through the host.
 * w1_ssid_out_dev (may be NULL) is the hand-over to the second pass: per utterance u the per-state
 * ssids (multiplex HMMs: hmm_mpx_ssid) its permanent single-phone word channels ended with, at
 * w1_ssid_out_dev + u*n_1ph*n_emit -- ngram_fwdflat_start clears those channels' scores but not
 * their ssids (ngram_search_fwdflat.c:385-392), so they are part of what psgpu_fwdflat_search_dev
 * takes over. */
int psgpu_fwdtree_search_dev(psgpu_fwdtree_t *m, const int16_t *senscr_dev, int64_t scr_stride,
                             const int32_t *penalties_dev, const int32_t *utt_off_dev, int32_t n_utt,
                             int32_t max_frames, int32_t bp_cap, int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev,
                             int32_t *idx_dev, int32_t *step_dev, int32_t *result_dev, int32_t raw_scores,
                             int32_t pl_window, int32_t *w1_ssid_out_dev, void *stream);
/* The same with a decoder session's carry-over.  The reference's permanent channels -- the lexicon tree's roots and the
 * single-phone words, multiplexed HMMs -- are cleared between utterances by hmm_clear (hmm.c:181-196), which resets scores
 * and histories but not the per-state ssids (hmm_mpx_ssid): utterance k + 1 of one ps_decoder_t starts with the ssids
 * utterance k left, and since a state's ssid decides which senone the search lists for it (compute_sen_active,
 * ngram_search_fwdtree.c:526-564), the active lists and with them every score's normaliser depend on it.
 * mpx_ssid_in_dev (NULL: freshly initialised channels, a new decoder) / mpx_ssid_out_dev (NULL: not wanted): per utterance
 * u at + u*n_mpx*n_emit, n_mpx = psgpu_fwdtree_n_mpx_channels = roots then single-phone words, [n_mpx][n_emit] int32
 * (entries of a non-multiplexed single-phone word are ignored / carry its senone ids).  The two may be one buffer.  A
 * caller that decodes a session's utterances one after another passes each call's output as the next call's input;
 * utterances of one call are independent of each other. */
int32_t psgpu_fwdtree_n_mpx_channels(const psgpu_fwdtree_t *m);
int psgpu_fwdtree_search_session_dev(psgpu_fwdtree_t *m, const int16_t *senscr_dev, int64_t scr_stride,
                                     const int32_t *penalties_dev, const int32_t *utt_off_dev, int32_t n_utt,
                                     int32_t max_frames, int32_t bp_cap, int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev,
                                     int32_t *idx_dev, int32_t *step_dev, int32_t *result_dev, int32_t raw_scores,
                                     int32_t pl_window, int32_t *w1_ssid_out_dev, const int32_t *mpx_ssid_in_dev,
                                     int32_t *mpx_ssid_out_dev, void *stream);
/* The NEXT psgpu_fwdtree_search_*_dev call on this handle also writes each utterance's hypothesis, as
 * psgpu_fwdtree_backtrace_dev would (same layout): the walk over the table is the search kernel's last step, one launch
 * less per batch.  One call's worth: the search call clears it.  (NULL, NULL, 0) withdraws it.
 * EXCLUSIVE USE: a psgpu_fwdtree_t carries per-call state -- this hand-over, psgpu_fwdtree_search_lag's, and the one work
 * slab its searches run in (grown on demand after synchronising the caller's stream only) -- so ONE caller, stream and
 * search at a time per handle: give every pipeline object / host thread a handle of its own (the tables are a few MB; the
 * Python mirror and integration/ create one per DecodePipeline / attachment). */
int psgpu_fwdtree_hyp_out(psgpu_fwdtree_t *m, int32_t *hyp_dev, int32_t *hyp_n_dev, int32_t max_words);
/* ngram_search_find_exit (ngram_search.c:500-544) + the backtrace of ngram_search_bp_hyp / the segment
 * iterator (:546-581, 903-1010) for every utterance of a batch, on the tables as the search left them:
 * hyp_dev + u*max_words*4 = {word id, start frame, end frame, path score at the word's end} per word in
 * spoken order; hyp_n_dev + u*4 = {number of words (when > max_words only the last max_words are
 * stored), path score of the exit, its back-pointer index (-1: no hypothesis), 0}. */
int psgpu_fwdtree_backtrace_dev(const psgpu_fwdtree_t *m, const int32_t *bp_dev, const int32_t *idx_dev,
                                const int32_t *result_dev, int32_t n_utt, int32_t max_frames, int32_t bp_cap,
                                int32_t max_words, int32_t *hyp_dev, int32_t *hyp_n_dev, void *stream);
/* number of permanently allocated single-phone word channels (ngs->n_1ph_words): the w1_ssid_out_dev row length is
 * this times the number of emitting states */
int32_t psgpu_fwdtree_n_single_phone_words(const psgpu_fwdtree_t *m);
/* *lds_layout = 1 when the tree-level state of this search lives in LDS; *slab_bytes_per_utt = device memory
 * the handle keeps per utterance of a batch.  Either pointer may be NULL.  (PSGPU_FWDTREE_LAYOUT=slab in the
 * environment at create forces the device-memory layout: the parity tests run both.) */
int psgpu_fwdtree_layout(const psgpu_fwdtree_t *m, int32_t *lds_layout, int64_t *slab_bytes_per_utt);
/* From the next search call on this handle uses the slab layout (all state in device memory, an evaluation list that holds every
 * channel) whatever psgpu_fwdtree_create chose: what a caller does about status 2 -- the LDS layout's evaluation list, sized by
 * what its 64 KB pool has left, was too short for some frame of this workload.  psgpu_decode_fetch_hyps does it by itself
 * (psgpu_decode_table_capacity's auto_grow) and repeats the call's search.  Scoring from top-N lists needs the LDS layout
 * (psgpu_fwdtree_can_score_lists answers 0 afterwards).  0, or PSGPU_EINVAL when the slab would exceed 8 GB per utterance. */
int psgpu_fwdtree_use_slab_layout(psgpu_fwdtree_t *m);
/* The slab layouts keep two per-utterance capacities small and grow them on demand: the tree nodes one frame may list (their channels
 * are kept compact, in list order; 65,536 at create) and the blocks of the pool the words' right-context channels come from (one block
 * per word with a live last phone, ngram_search_alloc_all_rc / ngram_search_free_all_rc, ngram_search.c:583-652; 2,048 at create).  A
 * search that needs more ends the utterance with status 4 / 5 (result word 3); psgpu_fwdtree_grow(m, that status) doubles the capacity
 * for the handle's later calls (up to every node / a block per dictionary word), and the caller repeats the search.  Status 6: a
 * frame's candidate / active-word counts outgrew the LDS arrays the word level works in; psgpu_fwdtree_grow(m, 6) moves them to the
 * slab for good --
 * psgpu_decode_fetch_hyps does both by itself (psgpu_decode_table_capacity's auto_grow).  0, or PSGPU_EINVAL (LDS layout; nothing left
 * to grow; arrays beyond 8 GB per utterance). */
int psgpu_fwdtree_grow(psgpu_fwdtree_t *m, int32_t status);
/* All three capacities at their ends at once (a channel place per tree node, a pool block per dictionary word, the word level's
 * arrays in the slab): no search on this handle can end with status 4 / 5 / 6 afterwards.  What a caller does whose searches cannot
 * be repeated -- psgpu_decode_streams_begin does it by itself: a stream's score rows are gone once searched, so a capacity met
 * in the middle of an utterance could not be raised and the utterance searched again as psgpu_decode_fetch_hyps does for batch
 * calls.  A no-op for the LDS layout.  0, or PSGPU_EINVAL (arrays beyond 8 GB per utterance; the capacities stay as they were). */
int psgpu_fwdtree_full_capacity(psgpu_fwdtree_t *m);


/* ---- the trigram language model on the device -----------------------
 * Replaces ngram_tg_score(ngs->lmset, w3, w2, w1, &n_used) (lm/ngram_model.c:451 ->
 * ngram_model_set_score, lm/ngram_model_set.c:685 -> ngram_ng_score, ngram_model.c:388 ->
 * ngram_model_trie_score / weight_score, lm/ngram_model_trie.c:710-742 -> lm_trie_score,
 * lm/lm_trie.c:813) for a model set holding ONE trie model without word classes: the
 * bit-packed reverse trie (lm/lm_trie.c:549-650, lm/bitarr.c:74), the 16-bit quantisation
 * tables (lm/lm_trie_quant.c:330-354), interpolation search included, bit-exact (float32
 * additions in the reference's order, weight_score's float multiply-add unfused).
 *
 * Tables, as integration/psgpu_lm_tables.c reads them out of a live ngram_model_t:
 *   unigrams   [n_unigrams + 1][3] uint32: {prob (float bits), backoff (float bits), next}
 *   ngram_mem  lm_trie_t.ngram_mem, the middle arrays then the longest one; level l < order - 2
 *              is middle l, level order - 2 the longest array
 *   quant      [2 * (order - 2) + 1][65536] float: middle l probabilities at row 2l, back-offs
 *              at row 2l + 1, the longest order's probabilities in the last row
 *   widmap     [n_words] dictionary word id -> model word id (ngram_model_set_t.widmap[w][0];
 *              -1 = not in the model) */
#define PSGPU_LM_MAX_LEVELS 4
typedef struct psgpu_lm_s psgpu_lm_t;
typedef struct psgpu_lm_tables_s {
    int32_t order, n_unigrams, n_words;
    const uint32_t *unigrams;
    const uint8_t *ngram_mem;
    uint64_t ngram_mem_size;
    uint32_t level_offset[PSGPU_LM_MAX_LEVELS];     /* byte offset of each level inside ngram_mem */
    uint32_t total_bits[PSGPU_LM_MAX_LEVELS], word_bits[PSGPU_LM_MAX_LEVELS], word_mask[PSGPU_LM_MAX_LEVELS];
    uint32_t max_vocab[PSGPU_LM_MAX_LEVELS], next_bits[PSGPU_LM_MAX_LEVELS], next_mask[PSGPU_LM_MAX_LEVELS];
    const float *quant;
    float lw;
    int32_t log_wip, log_zero;
    const int32_t *widmap;
    /* word classes (ngram_model_add_class / -lmctl class definitions; ngram_ng_score, lm/ngram_model.c:388-417): for a class word
     * widmap holds its class's TAG word (-1 when ngram_class_prob does not find it in the class: the look-up is log_zero), class_weight
     * [n_words] its in-class weight (0 for plain words; NULL: no classes), histmap [n_words] (or NULL: widmap) what the word is as a
     * history word -- the tag word whatever the weight.  (Zero-initialise the struct: older callers leave these NULL.) */
    const int32_t *class_weight, *histmap;
} psgpu_lm_tables_t;
int psgpu_lm_create(psgpu_lm_t **out, const psgpu_lm_tables_t *t);
void psgpu_lm_free(psgpu_lm_t *lm);
/* A model SET looked up without a current model (ngram_model_set_score with cur == -1, lm/ngram_model_set.c:685-727: after reading an
 * -lmctl file without -lmname, or ngram_model_set_interp): the log-sum over the members of lweights[i] + member i's look-up, through
 * the set's logmath table (addtab [addtab_size] of `width` 1 / 2 / 4 bytes = logadd_t.table, util/logmath.c:401-446; add_zero =
 * logmath_get_zero).  members [n_members]: handles of psgpu_lm_create, each with ITS widmap from the set's word ids; they stay the
 * caller's and must outlive the set handle.  A set with a current model needs none of this: the look-up is that member's.
 * (PSGPU_CAP_LM_SETS.) */
int psgpu_lm_create_interp(psgpu_lm_t **out, const psgpu_lm_t *const *members, const int32_t *lweights, int32_t n_members, const void *addtab,
                           int32_t width, int32_t addtab_size, int32_t add_zero, int32_t log_zero);
/* n independent look-ups: score_dev[i] = ngram_tg_score(lmset, w3[i], w2[i], w1[i], &n_used[i]);
 * w2 / w1 may be -1 (no history, as the search passes it); n_used_dev may be NULL. */
int psgpu_lm_tg_score_dev(const psgpu_lm_t *lm, const int32_t *w3_dev, const int32_t *w2_dev, const int32_t *w1_dev,
                          int64_t n, int32_t *score_dev, int32_t *n_used_dev, void *stream);
/* For the NEXT search call on this handle only: step through all but the last `lag` frames of every utterance, with the
 * penalties of all its frames available -- an utterance IN PROGRESS, as ps_search_forward (pocketsphinx.c:1173-1197) leaves
 * the phone loop (at frame n) and the n-gram search (at frame n - pl_window) between two ps_process_raw calls.  The tables
 * are then what the reference's search holds at that moment (ps_get_hyp in mid-utterance reads them).  0: the whole
 * utterance (the default).  Like psgpu_fwdtree_hyp_out this is per-call state of the handle: one caller at a time. */
int psgpu_fwdtree_search_lag(psgpu_fwdtree_t *m, int32_t lag);
/* For the NEXT search call on this handle only: a search that keeps its state between calls, as the reference's does between
 * two ps_search_forward rounds (ngram_fwdtree_search is called once per frame and everything it works on stays in the
 * ngram_search_t, ngram_search_fwdtree.c:1454-1495; pocketsphinx.c:1173-1197).  mode = PSGPU_SEARCH_KEEP: when the call stops
 * (at the lag, or at the utterances' ends) the state it stopped in is saved in the handle; PSGPU_SEARCH_RESUME: the call starts
 * from the state the handle's PREVIOUS search call saved -- frames already searched are not searched again -- and must be made
 * for the same utterances with the same table buffers and capacities, score rows and penalties of ALL frames so far (frames
 * numbered from the utterance's start: utt_off_dev [n_utt + 1] with more frames than before), the same raw_scores / pl_window;
 * PSGPU_SEARCH_KEEP | PSGPU_SEARCH_RESUME: both (a call in the middle of a live utterance).  The tables after every call are
 * the ones one call over the same frames would have written.  What is saved: the LDS layout's pool (~120 KB an utterance) and
 * the frame loop's counters; the slab layouts' state already lies in the handle's slab, which -- like the handle's other buffers --
 * must not be used by another search call in between (any search call without PSGPU_SEARCH_RESUME starts afresh). */
#define PSGPU_SEARCH_KEEP 1
#define PSGPU_SEARCH_RESUME 2
int psgpu_fwdtree_search_resume(psgpu_fwdtree_t *m, int32_t mode);
/* For the NEXT search call only: utterances in progress that grow at their own pace.  ext_dev [n_utt][3] int32 = {frames scored
 * so far, frame the search goes on to (<= the former), nonzero: an utterance that STARTS in this call takes its multiplexed
 * channels' ssids from mpx_ssid_in_dev -- a decoder's next utterance -- zero: it starts as a new decoder's} per utterance
 * replaces the back-to-back reading of utt_off_dev and the call's one lag; utt_off_dev [u] alone then places utterance u's score rows and penalties: frame f's at row utt_off_dev[u] + f.
 * Only the frames from the one the search resumes at are read, so a caller that keeps just those passes a start before its
 * buffer (a negative offset).  NULL: off.   psgpu_fwdtree_search_restart: utterance u of the handle's saved searches starts
 * afresh at the next PSGPU_SEARCH_RESUME call (ngram_fwdtree_start for it alone), the others go on. */
int psgpu_fwdtree_search_streams(psgpu_fwdtree_t *m, const int32_t *ext_dev);
int psgpu_fwdtree_search_restart(psgpu_fwdtree_t *m, int32_t u, void *stream);
/* Makes the tree search look its language scores up in `lm` (which must outlive it) instead of
 * the dense table of psgpu_fwdtree_tables_t.lm (which may then be NULL at create). */
int psgpu_fwdtree_set_lm(psgpu_fwdtree_t *m, const psgpu_lm_t *lm);

/* ---- the first pass of a batch of utterances as one device pipeline ---------------------------
 * 16-bit PCM -> MFCC -> 1s_c_d_dd features -> PTM senone scores (un-normalised rows) -> phone-loop
 * search -> lexicon-tree search -> back-pointer tables -> best exit and backtrace: the device side of
 * ps_decode_raw() (pocketsphinx.c:1030-1070) with -fwdflat no -bestpath no, followed by
 * ngram_search_bp_hyp (ngram_search.c:546-581), for n_utt utterances per call.  Every utterance is
 * decoded from the state a decoder has after ps_start_stream() on its first utterance (noise tracker
 * and top-N history reset), so results do not depend on the batch.  The object owns the buffers
 * between the stages; the stages are the handles of `cfg` (borrowed: they must outlive it). */
typedef struct psgpu_decode_s psgpu_decode_t;
typedef struct psgpu_decode_config_s {
    psgpu_fe_t *fe;                    /* front end (psgpu_fe_create) */
    psgpu_ptm_model_t *model;          /* scorer (psgpu_ptm_model_create) */
    psgpu_hmm_ctx_t *ctx;              /* tp / sseq (psgpu_hmm_ctx_create) */
    psgpu_fwdtree_t *ft;               /* the search (psgpu_fwdtree_create [+ psgpu_fwdtree_set_lm]) */
    psgpu_phone_loop_params_t pl;      /* phone_loop_search_t: n_phones, window, beams, pip, penalty weight */
    const uint16_t *pl_ssid;           /* HOST [pl.n_phones]: the CI phones' senone-sequence ids ... */
    const int16_t *pl_tmatid;          /* ... and transition-matrix ids */
    const uint16_t *ci_list;           /* HOST [n_ci_list]: the senone list acmod_flags2list builds when every CI */
    int32_t n_ci_list;                 /*   phone is active, bridging entries included (psgpu_phone_loop_run_dev) */
    int32_t pl_window;                 /* ps->pl_window: frames the phone loop runs ahead of the search (>= 1) */
    int32_t max_words;                 /* hypothesis records per utterance (0: 512) */
    /* The scorer.  PSGPU_SCORER_PTM (0): `model`.  Otherwise `model` is NULL and `scorer` is a psgpu_semi_model_t *
     * (PSGPU_SCORER_SEMI: s2_semi_mgau_frame_eval, s2_semi_mgau.c:837-883 -- its scores are final, neither the phone loop nor the
     * search subtracts anything) or a psgpu_ms_model_t * (PSGPU_SCORER_MS: ms_cont_mgau_frame_eval, ms_mgau.c:192-282 -- rows of
     * senone_eval values, normalised over each call's own list with the int16 clamp of :269-277), the two scorers acmod_init_am
     * (acmod.c:62-130) falls back to / is sent to by -senmgau.  Both are stateless across utterances in this pipeline: every
     * utterance is scored as by a new decoder.  psgpu_decode_score_mode needs the PTM scorer; psgpu_decode_session works with
     * the ms scorer too (it has no history of its own: the search's carry-over is what a session then holds) and is refused
     * for the semi-continuous one. */
    int32_t scorer_kind;
    void *scorer;
} psgpu_decode_config_t;
#define PSGPU_SCORER_PTM 0
#define PSGPU_SCORER_SEMI 1
#define PSGPU_SCORER_MS 2
int psgpu_decode_create(psgpu_decode_t **out, const psgpu_decode_config_t *cfg);
/* From PCM the pipeline computes en-us's feature type (1s_c_d_dd, batch CMN) unless another is installed here: any type of
 * psgpu_feat_create whose cepstrum size is the front end's and whose output is the scorer's vector (the semi-continuous models'
 * s2_4x, a model trained with -lda ...).  NULL: back to the default.  The handle stays the caller's. */
int psgpu_decode_set_feat(psgpu_decode_t *d, const psgpu_feat_t *feat);
void psgpu_decode_free(psgpu_decode_t *d);
/* after the scorer's tables were re-uploaded (MLLR): the new model handle, same shape */
int psgpu_decode_set_model(psgpu_decode_t *d, psgpu_ptm_model_t *model);
/* the same for a pipeline created with PSGPU_SCORER_SEMI / PSGPU_SCORER_MS: the new handle of that kind, same shape */
int psgpu_decode_set_scorer(psgpu_decode_t *d, void *scorer);
/* -compallsen yes (acmod.c:1098-1128): every senone is scored and the rows are normalised over ALL of them (the scorers' compallsen
 * branches, ptm_mgau.c:393-400, ms_mgau.c:213-236) instead of over what the phone loop / the search list; both then take the rows as
 * final scores.  PTM and multi-stream scorers; not together with psgpu_decode_score_mode(lists) or psgpu_decode_second_pass (the
 * device second pass scores and normalises its own lists). */
int psgpu_decode_compallsen(psgpu_decode_t *d, int32_t on);
/* Two pipeline objects taking turns.  The tree search is a latency-bound recurrence -- one workgroup per utterance, most
 * issue slots of its compute units idle -- and the stages before it are throughput-bound, so the front end and scorer of
 * one batch run BESIDE the search of another: one object per batch in flight, each on a stream with a hardware queue of
 * its own (psgpu_stream_create_dedicated).  The search kernel is built to leave room (168 registers a wave, ~50 KB of LDS a
 * workgroup: two workgroups on a compute unit leave a third of its registers, six of eight wave slots per SIMD and ~60 KB
 * of LDS), but two rules keep its placement sound, both enforced on the device by events once psgpu_decode_search_after
 * (a, b) and (b, a) have been called: (1) two searches are never resident together -- d's search waits for the search of
 * prev's latest call; (2) a search is dispatched onto a device that runs nothing else at that moment -- a call's first
 * stages wait until the search of prev's latest call has been dispatched (a search kernel dispatched while other kernels
 * hold LDS gets one workgroup per compute unit instead of two and takes twice as long, profiles/r03_overlap.txt).  The
 * caller alternates the objects and starts a call on one only when that object's previous results have been fetched.
 * psgpu_decode_wait_scored blocks the host until the latest call's stages before the search have finished.  prev = NULL
 * ends the arrangement; an object must not be freed while another names it as prev (psgpu_decode_search_after(x, NULL) first:
 * the successor keeps a bare pointer; pocketsphinx_amd/decode.py does this in close()). */
int psgpu_decode_search_after(psgpu_decode_t *d, psgpu_decode_t *prev);
int psgpu_decode_wait_scored(psgpu_decode_t *d);
/* a stream with a hardware queue of its own (streams of one priority may share a queue, and kernels of one queue never
 * overlap): hipExtStreamCreateWithCUMask with every compute unit enabled */
int psgpu_stream_create_dedicated(void **stream);
int psgpu_stream_destroy(void *stream);
/* lists != 0: no score rows -- the phone loop and the search evaluate the senones they list from the scorer's top-N lists
 * (psgpu_phone_loop_run_lists_dev, psgpu_fwdtree_search_lists_dev); the senone kernel is not run and rows_dev of the view is
 * NULL.  Same results bit for bit; 15.7 GB less traffic and a third less scorer time on the 512 x 30 s batch, but more time in
 * the latency-bound search: slower in total today, so rows are the default (PSGPU_DECODE_LISTS=1 changes it). */
int psgpu_decode_score_mode(psgpu_decode_t *d, int32_t lists);
/* Session mode.  The utterances of ONE call are decoded as by so many new reference decoders.  A reference decoder that
 * decodes utterances one after another carries state from each into the next: the PTM scorer's top-N lists (the first
 * frame of utterance k + 1 is seeded with the last lists of utterance k) and the per-state ssids of the
 * search's permanent multiplexed channels (psgpu_fwdtree_search_session_dev).  on != 0: every following call with
 * n_utt == 1 continues where the previous such call ended -- the object then behaves as one ps_decoder_t between two
 * ps_start_stream calls.  From PCM (psgpu_decode_first_pass[_dev]) the front end's noise tracker is carried as well
 * (noise_stats_t lives until ps_start_stream: fe_start_utt, fe_interface.c:318-326, does not reset it; the bundled
 * models' -cmn batch carries nothing); from feature vectors (psgpu_decode_first_pass_feat) what the front end and the
 * feature module carry is the caller's -- the ps_search_t binding takes its vectors from the reference's acmod.  Calling
 * it again (on or off) forgets the state: the next utterance is a new decoder's first.  Calls with n_utt != 1 neither
 * use nor change it. */
int psgpu_decode_session(psgpu_decode_t *d, int32_t on);
/* The session's state on the host, for a caller whose decoder also runs passes elsewhere (the reference's own second pass
 * re-scores the utterance and evaluates the single-phone channels again: what the next utterance inherits is then what
 * THAT pass left).  set: seed_cw [n_chain][topn] = the lists in slot n_fast_hist - 1 of the scorer's history ring (NULL: a
 * reset scorer), mpx_ssid [n_mpx][n_emit] (NULL: freshly initialised channels); get: what the last session utterance left
 * (*seed_valid = 0: no frame of it wrote that slot -- the lists given before still stand). */
int psgpu_decode_session_set(psgpu_decode_t *d, const uint8_t *seed_cw, const int32_t *mpx_ssid, void *stream);
int psgpu_decode_session_get(psgpu_decode_t *d, uint8_t *seed_cw, int32_t *seed_valid, int32_t *mpx_ssid, void *stream);
/* pcm_dev: the samples of n_utt utterances back to back, resident on the device; samp_off [n_utt + 1]
 * HOST array of sample offsets.  Asynchronous on `stream`; one call at a time per object.  Results stay on
 * the device (psgpu_decode_view) until fetched. */
int psgpu_decode_first_pass_dev(psgpu_decode_t *d, const int16_t *pcm_dev, const int64_t *samp_off, int32_t n_utt,
                                void *stream);
/* The NEXT call's front end and dynamic features, ahead of that call, on a stream of the object's own: for a caller that decodes
 * batch after batch with two objects taking turns (psgpu_decode_search_after) and has the next batch's samples on the device
 * already.  Issued while this object's latest search is still running and BEFORE the other object's next call, it runs beside
 * that call's scorer (whose top-N kernel uses no LDS) instead of at the start of this object's next call, where the spectrum kernel -- 4 KB of LDS a wavefront -- finds most
 * of every compute unit's LDS held by the resident search.  The next psgpu_decode_first_pass_dev with exactly this pcm_dev /
 * samp_off skips its front end.  Only for an input of the latest call's shape (same n_utt and samp_off: nothing is re-allocated
 * and the frame offsets the resident search reads stay what they are); otherwise, and for a session's single utterances, it does
 * nothing.  *started (may be NULL): 1 when the front end was issued.  A second pass that reads the features
 * (psgpu_decode_second_pass) must have been issued before.  Replaces nothing in the reference: scheduling of fe_process_frames +
 * feat_s2mfc2feat (fe_interface.c:345-560, feat.c:1310) for the batch entry. */
int psgpu_decode_front_end_ahead(psgpu_decode_t *d, const int16_t *pcm_dev, const int64_t *samp_off, int32_t n_utt, int32_t *started);
/* the same from host buffers: pcm[u][0..n[u]) are staged and copied to the device first */
int psgpu_decode_first_pass(psgpu_decode_t *d, const int16_t *const pcm[], const size_t n[], int32_t n_utt, void *stream);
/* entering after the front end: feat [total][3 * cepsize] HOST feature vectors as feat_s2mfc2feat_live leaves them
 * (acmod->feat_buf), frame_off [n_utt + 1] HOST.  For a binding whose host has already run the reference's own
 * front end (ps_process_raw -> ps_search_step per frame): both arrays are copied before the call returns. */
int psgpu_decode_first_pass_feat(psgpu_decode_t *d, const float *feat, const int32_t *frame_off, int32_t n_utt, void *stream);
/* Per-stage timing of psgpu_decode_first_pass_dev / psgpu_decode_first_pass: when enabled, HIP events are recorded on
 * the launch stream between the stages; ms[6] = front end, dynamic features, scorer (its three kernels), phone loop
 * (two kernels), lexicon-tree search kernel (the backtrace is its last step), and the time the stream waited for another
 * object's search before it (psgpu_decode_search_after; otherwise ~0) of the latest call (waits for it). */
int psgpu_decode_stage_timing(psgpu_decode_t *d, int32_t enable);
int psgpu_decode_last_stage_ms(psgpu_decode_t *d, float ms[6]);
/* what the last call left on the device (valid until the next call), for a second pass or a custom read-out:
 * tables as psgpu_fwdtree_search_dev writes them with the strides bp_cap / bss_cap / max_frames + 2,
 * hypotheses as psgpu_fwdtree_backtrace_dev writes them */
typedef struct psgpu_decode_view_s {
    int32_t n_utt, total_frames, max_frames, bp_cap, bss_cap, max_words;
    const int32_t *frame_off;          /* HOST [n_utt + 1] */
    const int32_t *frame_off_dev;
    const float *feat_dev;             /* [total][3 * cepsize] */
    const uint8_t *topn_cw_dev;        /* [n_chain][total][topn] */
    const int16_t *rows_dev;           /* [total][n_sen] un-normalised scores; NULL when the search scored its own senones
                                        * (psgpu_decode_score_mode) */
    const int32_t *penalties_dev;      /* [total][n_phones] */
    int32_t *bp_dev, *bss_dev, *idx_dev, *step_dev, *result_dev, *hyp_dev, *hyp_n_dev, *w1_ssid_dev;
    const int32_t *topn_score_dev;     /* [n_chain][total][topn] raw scores of the lists */
} psgpu_decode_view_t;
int psgpu_decode_view(const psgpu_decode_t *d, psgpu_decode_view_t *v);
/* hyp_n [n_utt][4], hyp [n_utt][max_words][4], result [n_utt][8] (any may be NULL) to the host; waits for the
 * stream.  This is the only transfer a caller that wants word sequences needs. */
int psgpu_decode_fetch_hyps(psgpu_decode_t *d, int32_t *hyp_n, int32_t *hyp, int32_t *result, void *stream);
/* Capacities of the per-utterance back-pointer table and right-context score stack: entries per frame of the call's
 * longest utterance (+ a constant); 0 keeps the current value.  Defaults 16 and 320 (a 100-word task writes 4-6 and 30-80
 * per frame; the 134,865-word task 30 and 800).  The reference grows both tables on demand (ngram_search_save_bp,
 * ngram_search.c:449-463, :468-480) and so never ends an utterance for lack of room.  auto_grow != 0 (the default) gives
 * the same behaviour here: when psgpu_decode_fetch_hyps finds an utterance that ended with status 1 (table full) it
 * doubles both allowances, allocates larger tables and repeats the SEARCH stage of that call on the scores still in the
 * object's buffers (the stages before it are not repeated), as often as needed; the larger allowance is kept for later
 * calls.  psgpu_decode_tables_grown = how many times that has happened since the object was created.  auto_grow = 0: an
 * utterance whose table fills up keeps status 1 and its hypothesis is that of the frames searched so far. */
int psgpu_decode_table_capacity(psgpu_decode_t *d, int32_t bp_per_frame, int32_t bss_per_frame, int32_t auto_grow);
/* the next psgpu_decode_first_pass* call only: psgpu_fwdtree_search_lag for its search (every stage before it runs over all
 * frames) -- partial results of an utterance in progress */
int psgpu_decode_search_lag(psgpu_decode_t *d, int32_t lag);
/* ---- one utterance IN PROGRESS (the device side of ps_process_raw called chunk by chunk, pocketsphinx.c:1173-1197, 1243-1282):
 * every stage goes on where the previous step's frames ended, so a live decode costs O(T) whatever the number of steps.
 *   psgpu_decode_live_begin   a new utterance of at most max_frames frames (session mode: psgpu_decode_session; what the utterance
 *                             inherits is what psgpu_decode_session_set / the previous utterance left).  Buffers for max_frames frames
 *                             are allocated here; a live utterance that outgrows them is begun again with a larger capacity and its
 *                             frames fed again (amortised O(T) when the capacity doubles).
 *   psgpu_decode_live_step    n_new more feature frames (host, [n_new][veclen] as for psgpu_decode_first_pass_feat): the scorer runs
 *                             on them from the previous frame's top-N lists (ptm_mgau.c:425-441), the phone loop steps through them
 *                             (psgpu_phone_loop_run_carry_dev), and the tree search goes on from the frame it stopped at up to `lag`
 *                             frames short of the frames scored so far (psgpu_fwdtree_search_resume; lag = 0: to the utterance's
 *                             end -- the utterance's last step).  After every step psgpu_decode_view / _fetch_hyps / _fetch_tables
 *                             return what ONE psgpu_decode_first_pass_feat call over the frames so far with psgpu_decode_search_lag(lag)
 *                             would have: the reference's tables at that moment.  n_new may be 0 (another lag).
 *   psgpu_decode_live_frames_searched   frames the search kernel has stepped through since live_begin, summed over the steps: the
 *                             utterance's frames searched so far when every frame was searched once. */
int psgpu_decode_live_begin(psgpu_decode_t *d, int32_t max_frames, void *stream);
/* the live utterance outgrew its capacity: it begins again with room for max_frames frames, from the session state it began with the
 * first time (which the steps so far have moved on); the caller feeds its frames again from the first one.
 * psgpu_decode_live_frames_searched keeps counting. */
int psgpu_decode_live_restart(psgpu_decode_t *d, int32_t max_frames, void *stream);
int psgpu_decode_live_step(psgpu_decode_t *d, const float *feat, int32_t n_new, int32_t lag, void *stream);
int64_t psgpu_decode_live_frames_searched(const psgpu_decode_t *d);
/* ---- MANY utterances in progress: a batch of live decoders (the serving shape of ps_process_raw called chunk by chunk on many
 * decoders at once).  Every stream is a new decoder's utterance, grows at its own pace, and costs what its frames cost:
 *   psgpu_decode_streams_begin    n_streams streams of at most max_frames frames an utterance, at most max_step_frames new frames a
 *                                 stream a step (buffers are sized from these).  The search runs pl_window frames behind the frames
 *                                 scored, as ps_search_forward keeps it, and to the utterance's end in the step that says it is the last.
 *   psgpu_decode_streams_step     feat: the streams' new feature frames (host, [sum n_new][veclen], stream after stream); n_new [n_streams]
 *                                 (host; 0: nothing for that stream this step); final_flags [n_streams] (host, or NULL: none): nonzero =
 *                                 the stream's utterance ends with these frames.  One launch set for all streams: batch scorer (every
 *                                 stream's lists seeded from its previous frame's), phone loop (psgpu_phone_loop_run_carry_dev), a copy
 *                                 kernel that keeps the score rows and penalties the searches have not reached yet, the tree search of all
 *                                 streams going on where each stopped (psgpu_fwdtree_search_resume / _streams).  Afterwards
 *                                 psgpu_decode_fetch_hyps / _fetch_tables return every stream's result record, hypothesis and tables as
 *                                 they stand -- for a stream in mid-utterance what ps_get_hyp would read at that point.  Tables do not
 *                                 grow here (psgpu_decode_table_capacity before _begin): a full one ends its stream with status 1.
 *   psgpu_decode_streams_restart  stream u's next frames begin a new utterance of a NEW decoder (nothing inherited);
 *   psgpu_decode_streams_next_utt the same decoder's next utterance (below).
 * psgpu_decode_live_frames_searched counts the frames the search stepped through, summed over the streams. */
int psgpu_decode_streams_begin(psgpu_decode_t *d, int32_t n_streams, int32_t max_frames, int32_t max_step_frames, void *stream);
int psgpu_decode_streams_step(psgpu_decode_t *d, const float *feat, const int32_t *n_new, const uint8_t *final_flags, void *stream);
int psgpu_decode_streams_restart(psgpu_decode_t *d, int32_t u, void *stream);
/* stream u's decoder goes on to its NEXT utterance (ps_start_utt after ps_end_utt on one decoder): like _restart, but the utterance
 * inherits what a decoder's does -- the scorer's ring slot that seeds its first frame and the multiplexed channels' per-state ssids
 * (psgpu_decode_session's carry-over, kept per stream).  The stream's previous utterance must have had its final step. */
int psgpu_decode_streams_next_utt(psgpu_decode_t *d, int32_t u, void *stream);
/* The streams fed with AUDIO (PSGPU_CAP_STREAMS_PCM): a batch of live decoders from ps_process_raw(full_utt = FALSE) on
 * (pocketsphinx.c:1210-1246).  _pcm_begin = psgpu_decode_streams_begin + per stream a new decoder's front half: the front end's
 * overflow samples and pre-emphasis prior, its noise tracker, the live cepstral mean (cmninit [n_cmninit]: -cmninit, config_macro.h:510,
 * "40,3,-1" by default) and the feature window, all on the device; grow_feat = the reference decoder's acmod_set_grow (TRUE with
 * -fwdflat yes, ngram_search.c:147): it decides how the reference cuts a call's cepstra into pieces -- and whether an utterance's last
 * cepstra can be dropped at the feature buffer's end (acmod.c:718-723).  _step_pcm: pcm = the step's samples, stream after stream
 * (n_samples [n] of them), one ps_process_raw call per stream with a non-zero count; final_flags[u]: + ps_end_utt.  The host walks the
 * reference's buffer counters for the step (which frames, which pieces, how many feature frames: integer bookkeeping); the
 * arithmetic runs on the device and the step's feature frames go straight into psgpu_decode_streams_step's stages.  n_new_out [n]
 * (or NULL) receives the feature frames each stream gained.  A stream's next utterance: psgpu_decode_streams_next_utt (noise tracker,
 * mean and window go on, as fe_start_utt / acmod_start_utt leave them) or _restart (a new decoder). */
/* The reference's counters for ONE utterance from its start, host only (no device): chunks [n_chunks] = the sample counts of the
 * ps_process_raw calls, final_ = ps_end_utt after them.  ops_out [ops_cap][2] (or NULL) receives the feat_s2mfc2feat_live calls as
 * (cepstra, flags: 1 beginutt, 2 endutt, 4 statistics only); *n_cepstra the frames the front end made, *n_feat_frames the feature frames
 * the searches received. */
int psgpu_live_pieces(int32_t frame_size, int32_t frame_shift, int32_t window, int32_t pl_window, int32_t grow_feat, const int64_t *chunks,
                      int32_t n_chunks, int32_t final_, int32_t *ops_out, int32_t ops_cap, int32_t *n_ops, int32_t *n_cepstra, int32_t *n_feat_frames);
int psgpu_decode_streams_pcm_begin(psgpu_decode_t *d, int32_t n_streams, int32_t max_frames, int32_t max_step_frames, const float *cmninit,
                                   int32_t n_cmninit, int32_t grow_feat, void *stream);
int psgpu_decode_streams_step_pcm(psgpu_decode_t *d, const int16_t *pcm, const int64_t *n_samples, const uint8_t *final_flags, int32_t *n_new_out,
                                  void *stream);
int32_t psgpu_decode_tables_grown(const psgpu_decode_t *d);
/* utterance u's tables to the host, cut to the sizes `result` reported: bp [10][n_bp] (column-major: ten columns of
 * n_bp), bss [n_bss], idx [n_idx]; waits for the stream.  What a binding needs to fill a bptbl_t array.  After
 * psgpu_decode_second_pass: the second pass's tables (with the result records psgpu_decode_fetch_hyps returns then). */
int psgpu_decode_fetch_tables(psgpu_decode_t *d, int32_t u, int32_t n_bp, int32_t n_bss, int32_t n_idx, int32_t *bp,
                              int32_t *bss, int32_t *idx, void *stream);
/* ... entries [bp0, bp0 + n_bp) of the table (bp [10][n_bp]), [bss0, bss0 + n_bss) of the score stack, [idx0, idx0 + n_idx) of the
 * frame marks: while an utterance is in progress (psgpu_decode_live_step) the tables only grow -- entries of frames already searched do
 * not change (ngram_search_save_bp updates an entry within its frame only, ngram_search.c:358-443) -- so a caller that holds an
 * earlier read-out asks for the rest. */
int psgpu_decode_fetch_tables_range(psgpu_decode_t *d, int32_t u, int32_t bp0, int32_t n_bp, int32_t bss0, int32_t n_bss, int32_t idx0,
                                    int32_t n_idx, int32_t *bp, int32_t *bss, int32_t *idx, void *stream);

/* ---- flat-lexicon second pass of whole utterances, first version ----------
 * Replaces ngram_fwdflat_start + ngram_fwdflat_search per frame + ngram_fwdflat_finish
 * (ngram_search_fwdflat.c:223-414, 416-877, 925-960).  `ft` are the first pass's static tables (the tree arrays
 * are not used); on top of them the second pass needs the pronunciations as word-internal ssids
 * (dict2pid_internal; -1 at the first and last position), the CI phones' ssids (bin_mdef_pid2ssid), which words
 * the language model knows (ngram_model_set_known_wid(lmset, dict_basewid(w)): build_fwdflat_wordlist :240-243),
 * its two beams, the end-point filter and start-frame window (-fwdflatefwid, -fwdflatsfwin) and
 * fwdflat_fwdtree_lw_ratio (ngram_search.c:122-125) -- what oracle/ref_dump.c `fwdflat` writes. */
typedef struct psgpu_fwdflat_s psgpu_fwdflat_t;
typedef struct psgpu_fwdflat_tables_s {
    const psgpu_fwdtree_tables_t *ft;
    const int32_t *pron_off;                  /* [n_w + 1] */
    const int32_t *pron_ci, *pron_ssid;       /* [pron_off[n_w]] */
    const int32_t *ci_ssid;                   /* [n_ci] */
    const int32_t *lm_known;                  /* [n_w] */
    int32_t fwdflatbeam, fwdflatwbeam, min_ef_width, max_sf_win;
    float lwf;
} psgpu_fwdflat_tables_t;
int psgpu_fwdflat_create(psgpu_fwdflat_t **out, const psgpu_fwdflat_tables_t *t);
void psgpu_fwdflat_free(psgpu_fwdflat_t *m);
/* language scores from the device trie instead of ft->lm (which may then be NULL at create) */
int psgpu_fwdflat_set_lm(psgpu_fwdflat_t *m, const psgpu_lm_t *lm);
/* n_utt utterances, one workgroup each, every frame inside the kernel.  senscr_dev [total][scr_stride] int16 =
 * the scores acmod_score hands the second pass for each frame; utt_off_dev [n_utt + 1].  The first pass is taken
 * over as psgpu_fwdtree_search_dev left it: bp1_dev (ten columns per utterance at u*10*bp1_cap) and result1_dev
 * (u*8: back-pointer count, .., frames searched); w1_ssid_dev [n_utt][n_1ph][n_emit] (may be NULL) = the per-state
 * ssids its permanent single-phone channels ended with (hmm_clear at ngram_fwdflat_start keeps them).  The
 * utterance's vocabulary (build_fwdflat_wordlist) is built on the host from three columns of that table.
 * Outputs as psgpu_fwdtree_search_dev's: back-pointer columns, score stack, bp_table_idx, per-frame
 * {best_score, 0, bpidx, n_active_words}, result {n back-pointers, score-stack length, frames searched, status,
 * best_score}.  Synchronous on `stream`. */
int psgpu_fwdflat_search_dev(psgpu_fwdflat_t *m, const int16_t *senscr_dev, int64_t scr_stride,
                             const int32_t *utt_off_dev, int32_t n_utt, int32_t max_frames,
                             int32_t bp1_cap, const int32_t *bp1_dev, const int32_t *result1_dev,
                             const int32_t *w1_ssid_dev, int32_t bp_cap, int32_t bss_cap, int32_t *bp_dev,
                             int32_t *bss_dev, int32_t *idx_dev, int32_t *step_dev, int32_t *result_dev, void *stream);
/* The same with the scores produced inside the kernel (PTM models), so that the second pass needs nothing from the host
 * but its launch: feats_dev [total][veclen] float (psgpu_feat_1s_c_d_dd_dev's output), `ptm` = device pointers to the
 * scorer's tables (psgpu_ptm_model_view), topn_seed_dev [n_utt][n_mgau * n_feat][topn] = the codewords of the history
 * slot pass-2 frame 0 is seeded from (ptm_mgau.c:425-441: slot n_fast_hist - 1 as the first pass left it, i.e. the lists
 * of the last first-pass frame t with t % n_fast_hist == n_fast_hist - 1; psgpu_ptm_score_batch_dev's topn_cw rows).
 * Per frame the kernel restates ptm_mgau_frame_eval as the second pass calls it: its own active senone list, only the
 * codebooks that list touches scanned and normalised (which is why these scores are not a shift of the first pass's rows). */
typedef struct psgpu_ptm_view_s {
    const float *mean, *var, *det;            /* as psgpu_ptm_model_create took them */
    const uint8_t *mixw, *sen2cb, *logadd8;
    int32_t n_mgau, n_feat, n_density, n_sen, veclen, topn, logadd8_size;
    int32_t featlen[16], featoff[16];
    const uint8_t *mixw_sen;                  /* the weights senone-major: [n_sen][n_feat][n_density rounded up to 64] */
} psgpu_ptm_view_t;
int psgpu_ptm_model_view(const psgpu_ptm_model_t *m, psgpu_ptm_view_t *out);

/* The lexicon-tree search scoring its own senones.  The reference's first pass asks the scorer for the senones of its
 * active channels only -- some 400 of en-us's 5126 per frame (ptm_mgau_frame_eval with the search's active list,
 * ptm_mgau.c:408-454) -- and so does this entry: instead of score rows it takes what the batched scorer's FIRST step
 * leaves, the top-N lists of every (codebook, stream) chain and frame (psgpu_ptm_score_batch_dev's topn_score_dev /
 * topn_cw_dev, chain-major, computed with senscr_dev = NULL), and evaluates ptm_mgau_codebook_norm (:265-295) and
 * ptm_mgau_senone_eval (:326-403) for the listed senones inside the kernel, frame by frame (csrc/psgpu_sen_dev.h).  The
 * full rows -- 10 KB per frame written by the senone kernel and read back -- never exist.  Everything else as
 * psgpu_fwdtree_search_session_dev with raw_scores = 1 (the frame's normaliser is the minimum over the listed senones,
 * bridging entries included; penalties_dev from psgpu_phone_loop_run_lists_dev).  Needs the LDS layout and a scorer of
 * 3 streams x top-4 with at most 128 chains: psgpu_fwdtree_can_score_lists says whether this model pair qualifies. */
int32_t psgpu_fwdtree_can_score_lists(const psgpu_fwdtree_t *m, const psgpu_ptm_view_t *v);
/* ... and the phone loop feeding it: psgpu_phone_loop_run_dev with the all-phones-active list's senones (ci_list_dev, some 130)
 * evaluated from the same lists instead of read from score rows */
int psgpu_phone_loop_run_lists_dev(psgpu_hmm_ctx_t *c, const psgpu_phone_loop_params_t *pp, const uint16_t *ssid_dev,
                                   const int16_t *tmatid_dev, const uint16_t *ci_list_dev, int32_t n_list,
                                   const psgpu_ptm_view_t *v, const int32_t *topn_score_dev, const uint8_t *topn_cw_dev,
                                   const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                                   int32_t *penalties_dev, int32_t *pen_now_dev, int32_t *state_dev, void *stream);
int psgpu_fwdtree_search_lists_dev(psgpu_fwdtree_t *m, const psgpu_ptm_view_t *v, const int32_t *topn_score_dev,
                                   const uint8_t *topn_cw_dev, int32_t total_frames,
                                   const int32_t *penalties_dev, const int32_t *utt_off_dev, int32_t n_utt,
                                   int32_t max_frames, int32_t bp_cap, int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev,
                                   int32_t *idx_dev, int32_t *step_dev, int32_t *result_dev, int32_t pl_window,
                                   int32_t *w1_ssid_out_dev, const int32_t *mpx_ssid_in_dev, int32_t *mpx_ssid_out_dev, void *stream);
int psgpu_fwdflat_search_feats_dev(psgpu_fwdflat_t *m, const psgpu_ptm_view_t *ptm, const float *feats_dev,
                                   const int32_t *topn_seed_dev, const int32_t *utt_off_dev, int32_t n_utt,
                                   int32_t max_frames, int32_t bp1_cap, const int32_t *bp1_dev,
                                   const int32_t *result1_dev, const int32_t *w1_ssid_dev, int32_t bp_cap,
                                   int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev, int32_t *idx_dev,
                                   int32_t *step_dev, int32_t *result_dev, void *stream);

/* The same taking the batch scorer's lists of the same frames along (psgpu_ptm_score_batch_dev's topn_score_dev /
 * topn_cw_dev and its open-entry flags, psgpu_ptm_batch_open_flags).  The reference's second pass re-scores the carried
 * lists of every codebook and scans the codebooks its active senones touch (ptm_mgau_codebook_eval, ptm_mgau.c:228-254);
 * for a touched codebook the outcome is the top-N of all its densities whatever list it started from, unless scores tie
 * or leave the key range (the closed form) -- i.e. the first pass's list of that frame wherever its entry
 * is not flagged open.  The kernel takes those lists and scans only open entries.  Same results. */
int psgpu_ptm_batch_open_flags(psgpu_ptm_model_t *m, void *stream, const uint8_t **flags_dev);
int psgpu_fwdflat_search_feats_lists_dev(psgpu_fwdflat_t *m, const psgpu_ptm_view_t *ptm, const float *feats_dev,
                                         const int32_t *topn_seed_dev, const int32_t *topn_score_dev,
                                         const uint8_t *topn_cw_dev, const uint8_t *open_flags_dev, int32_t total_frames,
                                         const int32_t *utt_off_dev, int32_t n_utt,
                                         int32_t max_frames, int32_t bp1_cap, const int32_t *bp1_dev,
                                         const int32_t *result1_dev, const int32_t *w1_ssid_dev, int32_t bp_cap,
                                         int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev, int32_t *idx_dev,
                                         int32_t *step_dev, int32_t *result_dev, void *stream);

/* The flat-lexicon second pass (ngram_fwdflat_search, src/ngram_search_fwdflat.c) as a stage of the pipeline object: after
 * psgpu_decode_first_pass* it runs psgpu_fwdflat_search_feats_lists_dev on what that call left in the object's buffers -- the
 * first pass's tables and permanent channels' ssids, the call's feature rows, the batch scorer's lists with their open flags,
 * each utterance's seeding lists (slot n_fast_hist - 1 of the scorer's history) -- into second-pass tables of its own, and
 * replaces the call's hypotheses by this pass's: psgpu_decode_fetch_hyps / psgpu_decode_fetch_tables then return the SECOND
 * pass's hypotheses, result records and tables (the first pass's stay where psgpu_decode_view shows them).  Full tables
 * grow as psgpu_decode_table_capacity says, for either pass.  PTM scorer only (the pass scores its own senones from that
 * scorer's model); synchronous (the pass's vocabulary is built on the host from the first pass's table).  `flat` must have been
 * created from the same dictionary / LM as the pipeline's search. */
int psgpu_decode_second_pass(psgpu_decode_t *d, psgpu_fwdflat_t *flat, void *stream);

/* Host-buffer form used by the search-side shim: n records in, the same n
 * records updated in place, *best = max(WORST_SCORE, returned best scores).
 * senscr is the frame's n_sen int16 scores.  Synchronous. */
int psgpu_hmm_vit_eval(psgpu_hmm_ctx_t *c, psgpu_hmm_rec_t *recs, int32_t n,
                       const int16_t *senscr, int32_t *best);

#ifdef __cplusplus
}
#endif
#endif /* PSGPU_H */
