/* integration/psgpu_fe_shim.c -- REFERENCE-SIDE code.
 * Reads the precomputed tables out of a fe_t and hands whole utterances to
 * psgpu_fe_process_utts(); compiled against the unmodified reference. */
#include <string.h>

#include <pocketsphinx.h>
#include "util/ckd_alloc.h"
#include "fe/fe_internal.h"
#include "pocketsphinx_internal.h"
#include "psgpu.h"
#include "psgpu_fe_shim.h"

struct psgpu_fe_shim_s {
    psgpu_fe_t *dev;
    int32 out_dim, n_filt;
    double *noise;                 /* [4][n_filt]: power, noise, floor, peak */
    int32_t undefined;
};

psgpu_fe_shim_t *
psgpu_fe_wrap(fe_t *fe)
{
    psgpu_fe_shim_t *s;
    psgpu_fe_params_t p;
    melfb_t *mel = fe->mel_fb;
    float *cosine;
    int i, rc;

    memset(&p, 0, sizeof p);
    p.frame_size = fe->frame_size; p.frame_shift = fe->frame_shift; p.fft_size = fe->fft_size;
    p.n_filt = mel->num_filters; p.num_cepstra = fe->num_cepstra; p.out_dim = fe->feature_dimension;
    p.transform = fe->transform; p.log_spec = fe->log_spec; p.remove_dc = fe->remove_dc;
    p.remove_noise = fe->noise_stats != NULL; p.swap = fe->swap; p.dither = fe->dither; p.dither_seed = fe->dither_seed;
    p.alpha = fe->pre_emphasis_alpha; p.sqrt_inv_n = mel->sqrt_inv_n; p.sqrt_inv_2n = mel->sqrt_inv_2n;
    /* mel_cosine is a ckd_calloc_2d block: rows are contiguous, but go through the row pointers anyway */
    cosine = ckd_calloc((size_t)fe->num_cepstra * mel->num_filters, sizeof *cosine);
    for (i = 0; i < fe->num_cepstra; ++i)
        memcpy(cosine + (size_t)i * mel->num_filters, mel->mel_cosine[i], sizeof(float) * mel->num_filters);
    s = ckd_calloc(1, sizeof *s);
    rc = psgpu_fe_create(&s->dev, &p, fe->hamming_window, fe->ccc, fe->sss, mel->spec_start, mel->filt_start,
                         mel->filt_width, mel->filt_coeffs, cosine, mel->lifter_val ? mel->lifter : NULL);
    ckd_free(cosine);
    if (rc != PSGPU_OK) {
        E_ERROR("psgpu_fe_create: %s\n", psgpu_last_error());
        ckd_free(s);
        return NULL;
    }
    s->out_dim = p.out_dim; s->n_filt = p.n_filt;
    s->noise = ckd_calloc((size_t)4 * p.n_filt, sizeof *s->noise);
    s->undefined = 1;
    return s;
}

void
psgpu_fe_shim_free(psgpu_fe_shim_t *s)
{
    if (!s) return;
    psgpu_fe_free(s->dev);
    ckd_free(s->noise);
    ckd_free(s);
}

struct psgpu_fe_s *
psgpu_fe_shim_release(psgpu_fe_shim_t *s)
{
    psgpu_fe_t *dev = s->dev;
    ckd_free(s->noise);
    ckd_free(s);
    return dev;
}

void
psgpu_fe_shim_reset_noise(psgpu_fe_shim_t *s)
{
    s->undefined = 1;
}

int
psgpu_fe_process_utt(psgpu_fe_shim_t *s, int16 const *spch, size_t nsamps, mfcc_t ***cep_block, int32 *nframes)
{
    int64_t off[2] = { 0, (int64_t)nsamps };
    int32_t fo[2];
    int64_t nfr = psgpu_fe_n_frames(s->dev, (int64_t)nsamps);
    mfcc_t **cep = (mfcc_t **)ckd_calloc_2d(nfr ? nfr : 1, s->out_dim, sizeof(mfcc_t));
    if (psgpu_fe_process_utts(s->dev, spch, off, 1, s->noise, &s->undefined, cep[0], fo) != PSGPU_OK) {
        E_ERROR("psgpu_fe_process_utts: %s\n", psgpu_last_error());
        ckd_free_2d(cep);
        return -1;
    }
    *cep_block = cep;
    *nframes = (int32)nfr;
    return 0;
}

int
psgpu_process_raw_full(ps_decoder_t *ps, psgpu_fe_shim_t *s, int16 const *data, size_t n_samples)
{
    mfcc_t **cep;
    int32 nfr;
    int rv;
    if (psgpu_fe_process_utt(s, data, n_samples, &cep, &nfr) < 0)
        return -1;
    rv = ps_process_cep(ps, cep, nfr, FALSE, TRUE);
    ckd_free_2d(cep);
    return rv;
}
