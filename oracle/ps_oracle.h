/* oracle/ps_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the PocketSphinx hot path (acoustic scoring and
 * the per-HMM Viterbi step) used as the checker for the HIP kernels and as the
 * "port" CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against fixtures under tests/golden/ that were produced by the
 * unmodified reference compiled into oracle/_ref (generator:
 * oracle/make_golden.py + oracle/ref_dump.c), and, when oracle/_ref is
 * present, against the live reference library.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to /root/reference/src).
 */
#ifndef PS_ORACLE_H
#define PS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSO_MAX_NEG_INT32 ((int32_t)0x80000000)   /* prim_type.h:168 */
#define PSO_WORST_DIST    PSO_MAX_NEG_INT32       /* tied_mgau_common.h:60 */
#define PSO_WORST_SCORE   ((int32_t)0xE0000000)   /* hmm.h:84 */
#define PSO_SENSCR_SHIFT  10                      /* hmm.h:73 */
#define PSO_MAX_NEG_ASCR  96                      /* tied_mgau_common.h:82 */
#define PSO_BAD_SSID      0xffff                  /* hmm.h:89 */
#define PSO_TMAT_WORST    255                     /* tmat.h: 8-bit floor */

/* ---------------- PTM scorer (ptm_mgau.c) ---------------- */

typedef struct pso_topn_s {
    int32_t cw;
    int32_t score;
} pso_topn_t;

typedef struct pso_ptm_s pso_ptm_t;

/* Tables are borrowed (not copied); they must outlive the object.
 *  mean/var : packed [n_mgau][n_feat][n_density][featlen[f]]  (ms_gauden.c:211-221)
 *  det      : [n_mgau][n_feat][n_density]
 *  mixw     : [n_feat][n_density][n_sen] (8-bit) or [..][(n_sen+1)/2] with a
 *             16-entry mixw_cb (4-bit)                         (ptm_mgau.c:456-661)
 *  logadd8  : uint8 table of logmath_init(base, 10, 1)          (logmath.c:62-162)
 */
pso_ptm_t *pso_ptm_new(int n_mgau, int n_feat, int n_density, const int32_t *featlen,
                       int n_sen, int topn, int ds_ratio, int n_fast_hist,
                       const float *mean, const float *var, const float *det,
                       const uint8_t *mixw, const uint8_t *mixw_cb,
                       const uint8_t *sen2cb,
                       const uint8_t *logadd8, int logadd8_size);
void pso_ptm_free(pso_ptm_t *s);
/* ptm_mgau_reset_fast_hist (ptm_mgau.c:777-802) */
void pso_ptm_reset_hist(pso_ptm_t *s);
/* acmod writes mgau->frame_idx directly (acmod.c:419,862,874) */
void pso_ptm_set_frame_idx(pso_ptm_t *s, int frame_idx);
int  pso_ptm_get_frame_idx(const pso_ptm_t *s);

/* ptm_mgau_frame_eval (ptm_mgau.c:408-454).  feat = the frame's dynamic
 * feature vector, streams concatenated (sum featlen floats).  If raw_topn is
 * non-NULL it receives the [n_mgau][n_feat][topn] lists as they stand after
 * codebook evaluation and BEFORE normalisation (only written when the
 * codebooks are actually evaluated, i.e. frame >= frame_idx; returns 1 then,
 * 0 when the history slot was reused). */
int pso_ptm_frame_eval(pso_ptm_t *s, int16_t *senscr,
                       const uint8_t *senone_active, int32_t n_senone_active,
                       const float *feat, int32_t frame, int32_t compallsen,
                       pso_topn_t *raw_topn);

/* current slot's (normalised) top-N, [n_mgau][n_feat][topn] */
const pso_topn_t *pso_ptm_cur_topn(const pso_ptm_t *s);

/* Convenience driver: score T frames of one utterance with compallsen,
 * doing what acmod_start_utt/acmod_advance do to frame_idx.  If
 * reset_hist != 0 the top-N history is reset first (fresh decoder).
 * Outputs (any may be NULL): senscr [T][n_sen]; topn_cw uint8
 * [T][n_mgau][n_feat][topn]; topn_raw int32 same shape. */
void pso_ptm_score_utt(pso_ptm_t *s, const float *feats, int T, int reset_hist,
                       int16_t *senscr, uint8_t *topn_cw, int32_t *topn_raw);

/* ---------------- semi-continuous scorer (s2_semi_mgau.c) ---------------- */

typedef struct pso_semi_s pso_semi_t;

/* One shared codebook, n_feat streams.
 *  mean/var : packed [n_feat][n_density][featlen[f]]
 *  det      : [n_feat][n_density]
 *  mixw     : [n_feat][n_density][n_sen] (8-bit) or [..][(n_sen+1)/2] with a
 *             16-entry mixw_cb (4-bit clustered)            (s2_semi_mgau.c:885-1080)
 *  topn_beam: [n_feat] per-stream beam, 0 = none             (:1296-1299) */
pso_semi_t *pso_semi_new(int n_feat, int n_density, const int32_t *featlen, int n_sen,
                         int topn, int ds_ratio, int n_hist, const uint8_t *topn_beam,
                         const float *mean, const float *var, const float *det,
                         const uint8_t *mixw, const uint8_t *mixw_cb,
                         const uint8_t *logadd8, int logadd8_size);
void pso_semi_free(pso_semi_t *s);
void pso_semi_reset_hist(pso_semi_t *s);             /* state after s2_semi_mgau_init (:1311-1322) */
void pso_semi_set_frame_idx(pso_semi_t *s, int frame_idx);
/* s2_semi_mgau_frame_eval (s2_semi_mgau.c:836-883) */
int pso_semi_frame_eval(pso_semi_t *s, int16_t *senscr,
                        const uint8_t *senone_active, int32_t n_senone_active,
                        const float *feat, int32_t frame, int32_t compallsen);
/* current slot: lists [n_feat][topn] and the per-stream counts topn_hist_n */
const pso_topn_t *pso_semi_cur_topn(const pso_semi_t *s, uint8_t *n_used);

/* ---------------- multi-stream / continuous scorer (ms_mgau.c, ms_gauden.c, ms_senone.c) ------ */

typedef struct pso_ms_s pso_ms_t;

/*  mean/var : packed [n_mgau][n_feat][n_density][featlen[f]]   (ms_gauden.c:211-221)
 *  det      : [n_mgau][n_feat][n_density]
 *  pdf      : senone mixture weights in the canonical order [n_sen][n_feat][n_density]
 *             (the reference keeps [feat][cw][sen] when there is one codebook,
 *             ms_senone.c:198-209; the caller transposes)
 *  sen2mgau : [n_sen] codebook of each senone                   (ms_senone.c:283-320)
 *  logadd   : the add table of senone_t.lmath = logmath_init(base, SENSCR_SHIFT, 1)
 *             (ms_senone.c:276), entries of `logadd_width` bytes; log_zero = lmath->zero */
pso_ms_t *pso_ms_new(int n_mgau, int n_feat, int n_density, const int32_t *featlen,
                     int n_sen, int topn, int aw,
                     const float *mean, const float *var, const float *det,
                     const uint8_t *pdf, const uint32_t *sen2mgau,
                     const void *logadd, int logadd_size, int logadd_width, int32_t log_zero);
void pso_ms_free(pso_ms_t *s);
/* ms_cont_mgau_frame_eval (ms_mgau.c:191-282).  senscr is IN/OUT: entries of
 * unlisted senones keep their previous contents, as in the reference. */
int pso_ms_frame_eval(pso_ms_t *s, int16_t *senscr,
                      const uint8_t *senone_active, int32_t n_senone_active,
                      const float *feat, int32_t compallsen);

/* ---------------- dynamic features (feat/feat.c, feat/cmn.c) ---------------- */

/* Whole-utterance feature computation of the "1s_c_d_dd" type with batch CMN and
 * no AGC, as feat_s2mfc2feat_live(begin = end = TRUE) does it (feat.c:1310 ->
 * feat_s2mfc2feat_block_utt :1275-1306): cmn() (cmn.c:166-208: mean over frames
 * whose c0 >= 0, summed in frame order, subtracted from every frame), the first
 * and last frame replicated over a window of 3, then per frame
 * [cep | cep[t+2]-cep[t-2] | (cep[t+3]-cep[t-1]) - (cep[t+1]-cep[t-3])]
 * (feat_1s_c_d_dd_cep2feat, feat.c:579-622).  cep [T][cepsize] is not modified;
 * out [T][3*cepsize]. */
void pso_dynfeat_1s_c_d_dd(const float *cep, int T, int cepsize, float *out);

/* ---------------- MFCC front end (fe/fe_sigproc.c, fe/fe_noise.c, fe/fe_interface.c) ---------------- */

/* Floating-point build of the reference front end (frame_t = powspec_t =
 * window_t = float64, mfcc_t = float32; fe/fe_type.h:58-60).  All tables are the
 * reference's own precomputed ones (fe_create_hamming, fe_create_twiddle,
 * fe_build_melfilters, fe_compute_melcosine; fe_sigproc.c:552-722,780-795,
 * 886-903) and are passed in, never regenerated. */
typedef struct pso_fe_s {
    int32_t frame_size, frame_shift, fft_size, fft_order;
    int32_t n_filt, num_cepstra, out_dim;      /* out_dim = fe_t.feature_dimension */
    int32_t transform;                         /* 0 legacy, 1 dct, 2 htk (fe_internal.h:65-69) */
    int32_t log_spec;                          /* 0, 1 raw, 2 smooth (fe_internal.h:59-62) */
    int32_t remove_dc, remove_noise, has_lifter;
    float alpha, sqrt_inv_n, sqrt_inv_2n;
    const double *hamming;                     /* [frame_size/2] */
    const double *ccc, *sss;                   /* [fft_size/4] */
    const int16_t *spec_start, *filt_start, *filt_width;   /* [n_filt] */
    const float *filt_coeffs;                  /* flattened */
    const float *mel_cosine;                   /* [num_cepstra][n_filt] */
    const float *lifter;                       /* [num_cepstra] or NULL */
} pso_fe_t;

/* Number of cepstral frames fe_start_utt + fe_process_frames(all samples) +
 * fe_end_utt produce for n samples (fe_interface.c:398-403, 526-541):
 * 1 + (n - frame_size)/frame_shift full frames when n >= frame_size, plus the
 * zero-padded tail frame whenever any sample is left over (always, for n > 0). */
int pso_fe_n_frames(const pso_fe_t *fe, long n);

/* One utterance, acmod_process_full_raw order (acmod.c:552-557).  noise
 * [4][n_filt] = power, noise, floor, peak of noise_stats_t, *undefined = its
 * "initialise on next frame" flag; both are read and updated (the reference
 * keeps them across utterances until ps_start_stream, pocketsphinx.c:1081).
 * cep [pso_fe_n_frames][out_dim].  Returns the number of frames. */
int pso_fe_process_utt(const pso_fe_t *fe, const int16_t *pcm, long n, float *cep,
                       double *noise, int32_t *undefined);

/* libm's log over an array (the front end's one transcendental) */
void pso_libm_log(const double *x, int64_t n, double *out);

/* ---------------- shared helpers ---------------- */

/* acmod_flags2list (acmod.c:1223-1275): bit flags -> uint8 delta list.
 * flags: one byte per senone (non-zero = active).  Returns n_active and
 * writes the deltas (at most n_sen entries). */
int pso_flags2list(const uint8_t *flags, int n_sen, uint8_t *deltas);

/* ---------------- HMM Viterbi step (hmm.c) ---------------- */

typedef struct pso_hmm_s {
    int32_t score[5];
    int32_t history[5];
    int32_t out_score;
    int32_t out_history;
    uint16_t ssid;          /* non-mpx: senone sequence id */
    uint16_t senid[5];      /* non-mpx: senone ids; mpx: per-state ssids */
    int32_t bestscore;
    int16_t tmatid;
    int32_t frame;
    uint8_t mpx;
    uint8_t n_emit_state;
} pso_hmm_t;

typedef struct pso_hmm_ctx_s {
    int n_emit_state;
    const uint8_t *tp;        /* [n_tmat][n_emit][n_emit+1] (tmat.h:60-66) */
    const int16_t *senscore;  /* current frame's senone scores */
    const uint16_t *sseq;     /* [n_sseq][n_emit] */
} pso_hmm_ctx_t;

/* hmm_vit_eval (hmm.c:786-805) and its 3/5-state (mpx) variants */
int32_t pso_hmm_vit_eval(const pso_hmm_ctx_t *ctx, pso_hmm_t *h);

#ifdef __cplusplus
}
#endif
#endif
