// psgpu_fe.hip -- the MFCC front end on gfx950: 16-bit PCM -> cepstra, for batches
// of whole utterances (the caller side of the scorer).
//
// Replaces fe_start_utt + fe_process_frames(all samples) + fe_end_utt as
// acmod_process_full_raw runs them (reference src/acmod.c:552-557) in the
// floating-point build (frame_t = powspec_t = window_t = float64, mfcc_t =
// float32; src/fe/fe_type.h:58-60): fe_spch_to_frame (pre-emphasis, zero padding,
// optional DC removal, Hamming window; src/fe/fe_sigproc.c:745-749,802-860),
// fe_fft_real (:1052-1149), fe_spec_magnitude (:1152-1191), fe_mel_spec
// (:1194-1213), fe_remove_noise (src/fe/fe_noise.c:268-364), fe_mel_cep with its
// three transforms and two log-spectrum modes (:1217-1342), fe_lifter (:1313).
//
// Arithmetic: every operation is the reference's, in the reference's order, in
// the reference's type (float64 except the float32 cepstral accumulators), no
// FMA contraction -- so everything up to the log mel spectrum and everything after
// it is bit-identical by construction.  The one libm call on the path,
// log(mfspec + 1e-4), goes through the device's double-precision log, which like
// glibc's is faithful but not correctly rounded; a last-bit difference there
// survives into a float32 cepstrum only when a float64 sum sits within 2^-29
// (relative) of a float32 rounding boundary.  tests/test_fe_gpu.py measures it:
// no differing value on any bundled recording.  The tables (window, twiddles,
// mel filters, DCT matrix, lifter) are the host's, uploaded, never regenerated.
//
// Three kernels:
//   fe_spectrum_kernel  one wave per frame: PCM -> windowed frame in LDS -> real
//                       FFT in LDS (n/4 independent butterflies per stage spread
//                       over the 64 lanes) -> power spectrum -> mel spectrum
//                       (lane = filter, sequential sum over its DFT bins) -> HBM
//                       float64 [frame][n_filt]
//   fe_noise_kernel     one wave per utterance, lane = mel channel: the noise
//                       tracker is a per-channel recurrence over the frames of an
//                       utterance; the 9-wide gain smoothing reads neighbours with
//                       cross-lane shuffles, summed in ascending channel order
//   fe_cepstrum_kernel  16 frames per workgroup: log, transform with float32
//                       accumulators (thread = (frame, coefficient), sequential over
//                       the filters as the reference), lifter -> float32 cepstra
// HBM traffic per frame: 2*frame_shift bytes of PCM in (samples are shared by
// overlapping frames and come from L2), 4*out_dim out, plus 8*n_filt written and
// read once or twice as the scratch mel spectrum.
#include "psgpu_internal.h"
#include <cstring>
#include <vector>

struct FeDev {
    int32_t frame_size, frame_shift, fft_size, fft_order;
    int32_t n_filt, num_cepstra, out_dim;
    int32_t transform, log_spec, remove_dc, remove_noise, has_lifter, swap;
    float alpha, sqrt_inv_n, sqrt_inv_2n;
    const double *hamming, *ccc, *sss;
    const int16_t *spec_start, *filt_start, *filt_width;
    const float *filt_coeffs, *mel_cosine, *lifter;
};

struct psgpu_fe_s {
    FeDev d;
    void *tables;             // one allocation holding every table
    double *mfspec;           // scratch [cap_frames][n_filt]
    int64_t cap_frames;
    int64_t *samp_off_dev;    // [cap_utt + 1]
    int32_t cap_utt;
    std::vector<int32_t> h_frame_off;   // host copies that outlive the asynchronous uploads
    std::vector<int64_t> h_samp_off;
};

static inline int64_t fe_n_frames(int32_t frame_size, int32_t frame_shift, int64_t n)
{
    // fe_interface.c:398-403 full frames; fe_end_utt :526-541 always finds left-over samples
    if (n <= 0) return 0;
    if (n < frame_size) return 1;
    return 1 + (n - frame_size) / frame_shift + 1;
}

constexpr int kFeFpb = 4;     // frames (waves) per workgroup of the spectrum kernel

__global__ __launch_bounds__(64 * kFeFpb)
void fe_spectrum_kernel(FeDev p, const int16_t *__restrict__ pcm, const int64_t *__restrict__ samp_off,
                        const int32_t *__restrict__ frame_off, int32_t n_utt, int32_t total_frames,
                        double *__restrict__ mfspec)
{
    extern __shared__ double s_x[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int frame = blockIdx.x * kFeFpb + w;
    const bool live = frame < total_frames;
    double *x = s_x + (size_t)w * p.fft_size;
    const int n = p.fft_size, m = p.fft_order, fs = p.frame_size;

    if (live) {
        // utterance of this frame: last u with frame_off[u] <= frame
        int lo = 0, hi = n_utt;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (frame_off[mid] <= frame) lo = mid; else hi = mid;
        }
        const int t = frame - frame_off[lo];
        const int64_t s0 = samp_off[lo], ns = samp_off[lo + 1] - s0;
        const int64_t start = (int64_t)t * p.frame_shift;
        const int len = (int)((ns - start < fs) ? ns - start : fs);
        const int16_t *in = pcm + s0 + start;
        auto sample = [&](int i) {                  // fe_read_frame_int16 :871-873: byte order of the input
            const uint16_t r = (uint16_t)in[i];
            return (double)(int16_t)(p.swap ? (uint16_t)((r << 8) | (r >> 8)) : r);
        };
        // fe_spch_to_frame: pre-emphasis against the sample before (0 at the utterance start), zero padding
        for (int i = lane; i < n; i += 64) {
            double v = 0.0;
            if (i < len) {
                v = sample(i);
                if (p.alpha != 0.0f) {
                    const double prev = (i > 0 || start > 0) ? sample(i - 1) : 0.0;
                    v = v - prev * (double)p.alpha;
                }
            }
            x[i] = v;
        }
    }
    __syncthreads();
    if (p.remove_dc) {
        // fe_hamming_window :802-815: the mean is summed in sample order
        if (live && lane == 0) {
            double mean = 0.0;
            for (int i = 0; i < fs; ++i) mean += x[i];
            mean /= (double)fs;
            // broadcast through the word behind the frames
            s_x[(size_t)kFeFpb * n + w] = mean;
        }
        __syncthreads();
        if (live) {
            const double mean = s_x[(size_t)kFeFpb * n + w];
            for (int i = lane; i < fs; i += 64) x[i] -= mean;
        }
        __syncthreads();
    }
    if (live) {
        for (int i = lane; i < fs; i += 64) {       // :826-829, symmetric half window
            const int half = fs >> 1;
            if (i < half) x[i] = x[i] * p.hamming[i];
            else if (i >= fs - half) x[i] = x[i] * p.hamming[fs - 1 - i];
        }
    }
    __syncthreads();
    if (live) {
        for (int i = lane; i < n; i += 64) {        // bit reversal :1062-1075 as pair swaps
            const int r = (int)(__brev((unsigned)i) >> (32 - m));
            if (i < r) { const double a = x[i]; x[i] = x[r]; x[r] = a; }
        }
    }
    __syncthreads();
    if (live) {
        for (int i = lane; i < (n >> 1); i += 64) { // :1081-1085
            const double a = x[2 * i], b = x[2 * i + 1];
            x[2 * i] = a + b;
            x[2 * i + 1] = a - b;
        }
    }
    __syncthreads();
    for (int k = 1; k < m; ++k) {                   // :1088-1144; n/4 independent work items per stage
        if (live) {
            const int half = 1 << k, quarter = half >> 1, blk = half << 1;
            for (int wi = lane; wi < (n >> 2); wi += 64) {
                const int j = wi & (quarter - 1), base = (wi >> (k - 1)) * blk;
                if (j == 0) {
                    const double a = x[base], b = x[base + half];
                    x[base] = a + b;
                    x[base + half] = a - b;
                    x[base + half + quarter] = -x[base + half + quarter];
                }
                else {
                    const int i1 = base + j, i2 = base + half - j, i3 = base + half + j, i4 = base + blk - j;
                    const double cc = p.ccc[j << (m - k - 1)], ss = p.sss[j << (m - k - 1)];
                    const double x3 = x[i3], x4 = x[i4], x1 = x[i1], x2 = x[i2];
                    const double t1 = x3 * cc + x4 * ss;
                    const double t2 = x3 * ss - x4 * cc;
                    x[i4] = x2 - t2;
                    x[i3] = -x2 - t2;
                    x[i2] = x1 - t1;
                    x[i1] = x1 + t1;
                }
            }
        }
        __syncthreads();
    }
    if (live) {
        for (int j = lane; j <= (n >> 1); j += 64) {   // fe_spec_magnitude, in place (j and n-j belong to j alone)
            const double re = x[j];
            if (j == 0) x[0] = re * re;
            else { const double im = x[n - j]; x[j] = re * re + im * im; }
        }
    }
    __syncthreads();
    if (live) {
        for (int f = lane; f < p.n_filt; f += 64) {     // fe_mel_spec :1194-1213
            const float *co = p.filt_coeffs + p.filt_start[f];
            const double *sp = x + p.spec_start[f];
            const int wd = p.filt_width[f];
            double a = 0.0;
            for (int j = 0; j < wd; ++j) a += sp[j] * (double)co[j];
            mfspec[(size_t)frame * p.n_filt + f] = a;
        }
    }
}

// fe_remove_noise, floating-point branches.  state [n_utt][4][n_filt] = power, noise, floor,
// peak; undefined [n_utt]; both optional (NULL: every utterance starts undefined, state dropped).
__global__ __launch_bounds__(64)
void fe_noise_kernel(FeDev p, const int32_t *__restrict__ frame_off, double *__restrict__ mfspec,
                     double *__restrict__ state, int32_t *__restrict__ undefined)
{
    const int u = blockIdx.x, lane = threadIdx.x, nf = p.n_filt;
    const int t0 = frame_off[u], T = frame_off[u + 1] - t0;
    if (T <= 0) return;
    const bool on = lane < nf;
    const int c = on ? lane : 0;
    // noise_stats_t constants (fe_noise.c:59-68, 203-213)
    const double l_pow = 0.7, c_pow = 1 - 0.7, l_a = 0.995, c_a = 1 - 0.995, l_b = 0.5, c_b = 1 - 0.5;
    const double l_t = 0.85, mu_t = 0.2, max_gain = 20, inv_max_gain = 1.0 / 20;
    double power = 0, noise = 0, floor_ = 0, peak = 0;
    bool undef = true;
    if (state) {
        undef = undefined[u] != 0;
        const double *st = state + (size_t)u * 4 * nf;
        power = st[c]; noise = st[nf + c]; floor_ = st[2 * nf + c]; peak = st[3 * nf + c];
    }
    const int l1 = c - 4 > 0 ? c - 4 : 0, l2 = c + 4 < nf - 1 ? c + 4 : nf - 1;
    double *mf = mfspec + (size_t)t0 * nf + c;
    double cur = *mf;
    for (int t = 0; t < T; ++t) {
        const double in = cur;
        if (t + 1 < T) cur = mf[(size_t)(t + 1) * nf];             // next frame's value while this one computes
        if (undef) {                                                // :282-298
            power = in;
            noise = in / max_gain;
            floor_ = in / max_gain;
            peak = 0.0;
            undef = false;
        }
        power = l_pow * power + c_pow * in;                         // :301-309
        noise = (power >= noise) ? l_a * noise + c_a * power : l_b * noise + c_b * power;     // fe_lower_envelope
        double sig = power - noise;                                 // :315-323
        if (sig < 1.0) sig = 1.0;
        floor_ = (sig >= floor_) ? l_a * floor_ + c_a * sig : l_b * floor_ + c_b * sig;       // :327
        const double cur_in = sig;                                  // fe_temp_masking :131-152
        peak *= l_t;
        if (sig < l_t * peak) sig = peak * mu_t;
        if (cur_in > peak) peak = cur_in;
        if (sig < floor_) sig = floor_;                             // :331-334
        double gain = (sig < max_gain * power) ? sig / power : max_gain;                      // :337-345
        if (gain < inv_max_gain) gain = inv_max_gain;
        double coef = 0.0;                                          // fe_weight_smooth :155-184, ascending order
#pragma unroll
        for (int d = -4; d <= 4; ++d) {
            const int j = c + d;
            const double g = __shfl(gain, j & 63);
            if (j >= l1 && j <= l2) coef += g;
        }
        if (on) mf[(size_t)t * nf] = in * (coef / (double)(l2 - l1 + 1));
    }
    if (state && on) {
        double *st = state + (size_t)u * 4 * nf;
        st[c] = power; st[nf + c] = noise; st[2 * nf + c] = floor_; st[3 * nf + c] = peak;
        if (lane == 0) undefined[u] = 0;
    }
}

constexpr int kCepFr = 16;    // frames per workgroup of the cepstrum kernel
constexpr int kCepThreads = 256;

__global__ __launch_bounds__(kCepThreads)
void fe_cepstrum_kernel(FeDev p, const double *__restrict__ mfspec, int32_t total_frames, float *__restrict__ cep)
{
    extern __shared__ double s_log[];               // [kCepFr][n_filt] doubles, then [kCepFr][num_cepstra] floats
    const int nf = p.n_filt, nc = p.num_cepstra, od = p.out_dim;
    float *s_c = (float *)(s_log + (size_t)kCepFr * nf);
    const int f0 = blockIdx.x * kCepFr;
    const int nfr = total_frames - f0 < kCepFr ? total_frames - f0 : kCepFr;
    for (int e = threadIdx.x; e < nfr * nf; e += kCepThreads)      // LOG_FLOOR :1215,1228
        s_log[e] = log(mfspec[(size_t)f0 * nf + e] + 1e-4);
    __syncthreads();
    if (p.log_spec == 1) {                                          // RAW_LOG_SPEC :1233-1237
        for (int e = threadIdx.x; e < nfr * od; e += kCepThreads) {
            const int fr = e / od, i = e - fr * od;
            float v = (float)s_log[fr * nf + i];
            if (p.has_lifter && i < nc) v = v * p.lifter[i];
            cep[(size_t)(f0 + fr) * od + i] = v;
        }
        return;
    }
    const bool smooth = p.log_spec == 2;
    for (int e = threadIdx.x; e < nfr * nc; e += kCepThreads) {
        const int fr = e / nc, i = e - fr * nc;
        const double *ml = s_log + fr * nf;
        float o;
        if (smooth || p.transform != 0) {                           // fe_dct2 :1288-1310
            if (i == 0) {
                o = (float)ml[0];
                for (int j = 1; j < nf; ++j) o = (float)((double)o + ml[j]);
                o = o * ((p.transform == 2 && !smooth) ? p.sqrt_inv_2n : p.sqrt_inv_n);
            }
            else {
                const float *mc = p.mel_cosine + i * nf;
                o = 0.0f;
                for (int j = 0; j < nf; ++j) o = (float)((double)o + ml[j] * (double)mc[j]);
                o = o * p.sqrt_inv_2n;
            }
        }
        else {                                                      // fe_spec2cep :1257-1285
            if (i == 0) {
                o = (float)(ml[0] / 2);
                for (int j = 1; j < nf; ++j) o = (float)((double)o + ml[j]);
                o = (float)((double)o / (double)nf);
            }
            else {
                const float *mc = p.mel_cosine + i * nf;
                o = 0.0f;
                for (int j = 0; j < nf; ++j) {
                    const int beta = j == 0 ? 1 : 2;
                    o = (float)((double)o + ml[j] * (double)mc[j] * (double)beta);
                }
                o = (float)((double)o / ((double)nf * 2));
            }
        }
        if (smooth) s_c[fr * nc + i] = o;
        else {
            if (p.has_lifter) o = o * p.lifter[i];                  // fe_lifter :1313-1323
            cep[(size_t)(f0 + fr) * od + i] = o;
        }
    }
    if (!smooth) return;
    __syncthreads();
    for (int e = threadIdx.x; e < nfr * od; e += kCepThreads) {     // SMOOTH_LOG_SPEC :1240-1248, fe_dct3 :1326-1338
        const int fr = e / od, i = e - fr * od;
        const float *c = s_c + fr * nc;
        double a = (double)c[0] * 0.707106781186548;                // SQRT_HALF is a double constant (fe_internal.h:106)
        for (int j = 1; j < nc; ++j) a += (double)(c[j] * p.mel_cosine[j * nf + i]);
        a = a * (double)p.sqrt_inv_2n;
        float v = (float)a;
        if (p.has_lifter && i < nc) v = v * p.lifter[i];
        cep[(size_t)(f0 + fr) * od + i] = v;
    }
}

extern "C" {

int psgpu_fe_create(psgpu_fe_t **out, const psgpu_fe_params_t *pp, const double *hamming,
                    const double *ccc, const double *sss, const int16_t *spec_start,
                    const int16_t *filt_start, const int16_t *filt_width, const float *filt_coeffs,
                    const float *mel_cosine, const float *lifter)
{
    PSGPU_REQUIRE(out && pp && hamming && ccc && sss && spec_start && filt_start && filt_width && filt_coeffs &&
                  mel_cosine, "psgpu_fe_create: NULL argument");
    *out = nullptr;
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;
    const psgpu_fe_params_t &q = *pp;
    PSGPU_REQUIRE(q.frame_shift >= 1 && q.frame_size >= q.frame_shift, "frame size %d / shift %d", q.frame_size,
                  q.frame_shift);
    int order = 0;
    while ((1 << order) < q.fft_size) ++order;
    PSGPU_REQUIRE(q.fft_size >= 8 && (1 << order) == q.fft_size && q.fft_size >= q.frame_size,
                  "fft size %d must be a power of two >= the frame size %d", q.fft_size, q.frame_size);
    if (q.fft_size > 4096) { psgpu_set_error("fft size %d > 4096 not supported", q.fft_size); return PSGPU_EINVAL; }
    PSGPU_REQUIRE(q.n_filt >= 1 && q.n_filt <= 256 && q.num_cepstra >= 1 && q.num_cepstra <= q.n_filt,
                  "n_filt %d / num_cepstra %d", q.n_filt, q.num_cepstra);
    PSGPU_REQUIRE(q.transform >= 0 && q.transform <= 2 && q.log_spec >= 0 && q.log_spec <= 2, "transform / log_spec");
    PSGPU_REQUIRE(q.out_dim == (q.log_spec ? q.n_filt : q.num_cepstra), "out_dim %d inconsistent", q.out_dim);
    if (q.remove_noise && q.n_filt > 64) {
        psgpu_set_error("noise removal with %d > 64 mel filters not supported", q.n_filt);
        return PSGPU_EINVAL;
    }
    if (q.dither) {
        // fe_read_frame_int16 (fe_sigproc.c:876-878) adds s3_rand_int31() noise sample by sample,
        // seeded from the clock by default: there is nothing deterministic to reproduce
        psgpu_set_error("dither is not supported");
        return PSGPU_EINVAL;
    }
    int ncoef = 0;
    for (int i = 0; i < q.n_filt; ++i) {
        PSGPU_REQUIRE(filt_width[i] >= 0 && spec_start[i] >= 0 && spec_start[i] + filt_width[i] <= q.fft_size / 2 + 1,
                      "mel filter %d outside the spectrum", i);
        PSGPU_REQUIRE(filt_start[i] == ncoef, "filt_start[%d] = %d, expected %d", i, filt_start[i], ncoef);
        ncoef += filt_width[i];
    }
    psgpu_fe_s *fe = new psgpu_fe_s();
    FeDev &d = fe->d;
    d.frame_size = q.frame_size; d.frame_shift = q.frame_shift; d.fft_size = q.fft_size; d.fft_order = order;
    d.n_filt = q.n_filt; d.num_cepstra = q.num_cepstra; d.out_dim = q.out_dim; d.transform = q.transform;
    d.log_spec = q.log_spec; d.remove_dc = q.remove_dc; d.remove_noise = q.remove_noise; d.has_lifter = lifter != nullptr;
    d.swap = q.swap != 0;
    d.alpha = q.alpha; d.sqrt_inv_n = q.sqrt_inv_n; d.sqrt_inv_2n = q.sqrt_inv_2n;
    // one blob: doubles first, then floats, then int16s
    const size_t n_ham = q.frame_size / 2, n_tw = q.fft_size / 4, n_cos = (size_t)q.num_cepstra * q.n_filt;
    const size_t n_lift = lifter ? q.num_cepstra : 0;
    const size_t bytes = 8 * (n_ham + 2 * n_tw) + 4 * (ncoef + n_cos + n_lift) + 2 * 3 * (size_t)q.n_filt;
    std::vector<uint8_t> h(bytes);
    size_t o = 0;
    auto put = [&](const void *src, size_t nb) { memcpy(h.data() + o, src, nb); size_t at = o; o += nb; return at; };
    const size_t o_ham = put(hamming, 8 * n_ham), o_c = put(ccc, 8 * n_tw), o_s = put(sss, 8 * n_tw);
    const size_t o_fc = put(filt_coeffs, 4 * (size_t)ncoef), o_mc = put(mel_cosine, 4 * n_cos);
    const size_t o_li = lifter ? put(lifter, 4 * n_lift) : 0;
    const size_t o_ss = put(spec_start, 2 * (size_t)q.n_filt), o_fs = put(filt_start, 2 * (size_t)q.n_filt);
    const size_t o_fw = put(filt_width, 2 * (size_t)q.n_filt);
    hipError_t e = hipMalloc(&fe->tables, bytes);
    if (e == hipSuccess) e = hipMemcpy(fe->tables, h.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        psgpu_set_error("psgpu_fe_create: %s", hipGetErrorString(e));
        hipFree(fe->tables); delete fe;
        return e == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP;
    }
    const uint8_t *b = (const uint8_t *)fe->tables;
    d.hamming = (const double *)(b + o_ham); d.ccc = (const double *)(b + o_c); d.sss = (const double *)(b + o_s);
    d.filt_coeffs = (const float *)(b + o_fc); d.mel_cosine = (const float *)(b + o_mc);
    d.lifter = lifter ? (const float *)(b + o_li) : nullptr;
    d.spec_start = (const int16_t *)(b + o_ss); d.filt_start = (const int16_t *)(b + o_fs);
    d.filt_width = (const int16_t *)(b + o_fw);
    *out = fe;
    return PSGPU_OK;
}

void psgpu_fe_free(psgpu_fe_t *fe)
{
    if (!fe) return;
    hipFree(fe->tables); hipFree(fe->mfspec); hipFree(fe->samp_off_dev);
    delete fe;
}

int32_t psgpu_fe_out_dim(const psgpu_fe_t *fe) { return fe ? fe->d.out_dim : 0; }

int64_t psgpu_fe_n_frames(const psgpu_fe_t *fe, int64_t n_samples)
{
    return fe ? fe_n_frames(fe->d.frame_size, fe->d.frame_shift, n_samples) : 0;
}

int psgpu_fe_process_utts_dev(psgpu_fe_t *fe, const int16_t *pcm_dev, const int64_t *samp_off, int32_t n_utt,
                              double *noise_dev, int32_t *undefined_dev, float *cep_dev,
                              int32_t *frame_off_dev, int32_t *frame_off, void *stream)
{
    PSGPU_REQUIRE(fe && samp_off && n_utt >= 0, "psgpu_fe_process_utts_dev: bad argument");
    PSGPU_REQUIRE((noise_dev == nullptr) == (undefined_dev == nullptr), "noise state needs both arrays");
    if (n_utt == 0) return PSGPU_OK;
    PSGPU_REQUIRE(pcm_dev && cep_dev && frame_off_dev, "psgpu_fe_process_utts_dev: NULL device buffer");
    hipStream_t st = (hipStream_t)stream;
    const FeDev &d = fe->d;
    std::vector<int32_t> &fo = fe->h_frame_off;
    fo.resize((size_t)n_utt + 1);
    fe->h_samp_off.assign(samp_off, samp_off + n_utt + 1);
    int64_t total = 0;
    fo[0] = 0;
    for (int u = 0; u < n_utt; ++u) {
        PSGPU_REQUIRE(samp_off[u + 1] >= samp_off[u], "samp_off must be non-decreasing");
        total += fe_n_frames(d.frame_size, d.frame_shift, samp_off[u + 1] - samp_off[u]);
        PSGPU_REQUIRE(total < (int64_t)1 << 31, "more than 2^31 frames in one call");
        fo[u + 1] = (int32_t)total;
    }
    if (frame_off) memcpy(frame_off, fo.data(), sizeof(int32_t) * (n_utt + 1));
    if (n_utt > fe->cap_utt) {
        PSGPU_HIP(hipFree(fe->samp_off_dev)); fe->samp_off_dev = nullptr; fe->cap_utt = 0;
        PSGPU_HIP(hipMalloc((void **)&fe->samp_off_dev, sizeof(int64_t) * ((size_t)n_utt + 1)));
        fe->cap_utt = n_utt;
    }
    if (total > fe->cap_frames) {
        PSGPU_HIP(hipFree(fe->mfspec)); fe->mfspec = nullptr; fe->cap_frames = 0;
        PSGPU_HIP(hipMalloc((void **)&fe->mfspec, sizeof(double) * (size_t)total * d.n_filt));
        fe->cap_frames = total;
    }
    PSGPU_HIP(hipMemcpyAsync(fe->samp_off_dev, fe->h_samp_off.data(), sizeof(int64_t) * ((size_t)n_utt + 1), hipMemcpyHostToDevice, st));
    PSGPU_HIP(hipMemcpyAsync(frame_off_dev, fo.data(), sizeof(int32_t) * ((size_t)n_utt + 1), hipMemcpyHostToDevice, st));
    if (total == 0) return PSGPU_OK;
    const int T = (int)total;
    const size_t lds1 = sizeof(double) * ((size_t)kFeFpb * d.fft_size + kFeFpb);
    if (lds1 > 64 * 1024)
        PSGPU_HIP(hipFuncSetAttribute((const void *)fe_spectrum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    hipLaunchKernelGGL(fe_spectrum_kernel, dim3((T + kFeFpb - 1) / kFeFpb), dim3(64 * kFeFpb), lds1, st, d, pcm_dev,
                       fe->samp_off_dev, frame_off_dev, n_utt, T, fe->mfspec);
    if (d.remove_noise)
        hipLaunchKernelGGL(fe_noise_kernel, dim3(n_utt), dim3(64), 0, st, d, frame_off_dev, fe->mfspec, noise_dev,
                           undefined_dev);
    const size_t lds3 = sizeof(double) * (size_t)kCepFr * d.n_filt + sizeof(float) * (size_t)kCepFr * d.num_cepstra;
    hipLaunchKernelGGL(fe_cepstrum_kernel, dim3((T + kCepFr - 1) / kCepFr), dim3(kCepThreads), lds3, st, d, fe->mfspec, T,
                       cep_dev);
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

int psgpu_fe_process_utts(psgpu_fe_t *fe, const int16_t *pcm, const int64_t *samp_off, int32_t n_utt,
                          double *noise, int32_t *undefined, float *cep, int32_t *frame_off)
{
    PSGPU_REQUIRE(fe && samp_off && n_utt >= 0 && frame_off, "psgpu_fe_process_utts: bad argument");
    PSGPU_REQUIRE((noise == nullptr) == (undefined == nullptr), "noise state needs both arrays");
    if (n_utt == 0) return PSGPU_OK;
    PSGPU_REQUIRE(samp_off[0] == 0, "samp_off must start at 0");
    const FeDev &d = fe->d;
    const int64_t ns = samp_off[n_utt];
    int64_t total = 0;
    for (int u = 0; u < n_utt; ++u) total += fe_n_frames(d.frame_size, d.frame_shift, samp_off[u + 1] - samp_off[u]);
    PSGPU_REQUIRE(pcm || ns == 0, "NULL pcm");
    PSGPU_REQUIRE(cep || total == 0, "NULL cep");
    int16_t *dp = nullptr; float *dc = nullptr; int32_t *dfo = nullptr, *dun = nullptr; double *dno = nullptr;
    auto cleanup = [&]() { hipFree(dp); hipFree(dc); hipFree(dfo); hipFree(dun); hipFree(dno); };
#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) {                 \
        psgpu_set_error("%s -> %s", #call, hipGetErrorString(e_)); cleanup();          \
        return e_ == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP; } } while (0)
    const size_t nst = (size_t)n_utt * 4 * d.n_filt;
    TRY(hipMalloc((void **)&dp, sizeof(int16_t) * (size_t)(ns > 0 ? ns : 1)));
    TRY(hipMalloc((void **)&dc, sizeof(float) * (size_t)(total > 0 ? total : 1) * d.out_dim));
    TRY(hipMalloc((void **)&dfo, sizeof(int32_t) * ((size_t)n_utt + 1)));
    if (ns > 0) TRY(hipMemcpy(dp, pcm, sizeof(int16_t) * (size_t)ns, hipMemcpyHostToDevice));
    if (noise) {
        TRY(hipMalloc((void **)&dno, sizeof(double) * nst));
        TRY(hipMalloc((void **)&dun, sizeof(int32_t) * (size_t)n_utt));
        TRY(hipMemcpy(dno, noise, sizeof(double) * nst, hipMemcpyHostToDevice));
        TRY(hipMemcpy(dun, undefined, sizeof(int32_t) * (size_t)n_utt, hipMemcpyHostToDevice));
    }
    int rc = psgpu_fe_process_utts_dev(fe, dp, samp_off, n_utt, dno, dun, dc, dfo, frame_off, nullptr);
    if (rc == PSGPU_OK) {
        TRY(hipDeviceSynchronize());
        if (total > 0) TRY(hipMemcpy(cep, dc, sizeof(float) * (size_t)total * d.out_dim, hipMemcpyDeviceToHost));
        if (noise) {
            TRY(hipMemcpy(noise, dno, sizeof(double) * nst, hipMemcpyDeviceToHost));
            TRY(hipMemcpy(undefined, dun, sizeof(int32_t) * (size_t)n_utt, hipMemcpyDeviceToHost));
        }
    }
#undef TRY
    cleanup();
    return rc;
}

}  // extern "C"
