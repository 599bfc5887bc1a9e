"""Host-side mirror of the reference front end object (fe_t, reference
src/fe/fe_internal.h:108-161; fe_process_utt + fe_end_utt, src/fe/fe_interface.c:
505-541): 16-bit PCM of whole utterances -> MFCC frames on the MI355X.
Arithmetic in csrc/psgpu_fe.hip."""
import ctypes as C

import numpy as np

from . import capi


class _Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("frame_size", "frame_shift", "fft_size", "n_filt", "num_cepstra", "out_dim",
                                         "transform", "log_spec", "remove_dc", "remove_noise", "swap", "dither")] + \
               [(n, C.c_float) for n in ("alpha", "sqrt_inv_n", "sqrt_inv_2n")] + [("dither_seed", C.c_int32)]


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class FrontEnd:
    """`t`: the reference front end's parameters and precomputed tables (the arrays
    ref_dump mfcc writes / an integration reads out of its fe_t): par
    [frame_size, frame_shift, fft_size, fft_order, n_filt, num_cepstra, out_dim,
    transform, log_spec, remove_dc, remove_noise, lifter_val, swap, dither], dither_seed (optional: -1), alpha,
    sqrt_inv_n, sqrt_inv_2n, hamming, ccc, sss, spec_start, filt_start, filt_width,
    filt_coeffs, mel_cosine, lifter (absent when lifter_val == 0)."""

    def __init__(self, t):
        par = [int(v) for v in t["par"]]
        p = _Params()
        (p.frame_size, p.frame_shift, p.fft_size) = par[0:3]
        (p.n_filt, p.num_cepstra, p.out_dim, p.transform, p.log_spec, p.remove_dc, p.remove_noise) = par[4:11]
        p.swap, p.dither = par[12], par[13]
        p.dither_seed = int(t["dither_seed"][0]) if "dither_seed" in t else -1
        p.alpha = float(t["alpha"][0]); p.sqrt_inv_n = float(t["sqrt_inv_n"][0]); p.sqrt_inv_2n = float(t["sqrt_inv_2n"][0])
        k = dict(hamming=np.ascontiguousarray(t["hamming"], np.float64), ccc=np.ascontiguousarray(t["ccc"], np.float64),
                 sss=np.ascontiguousarray(t["sss"], np.float64),
                 spec_start=np.ascontiguousarray(t["spec_start"], np.int16),
                 filt_start=np.ascontiguousarray(t["filt_start"], np.int16),
                 filt_width=np.ascontiguousarray(t["filt_width"], np.int16),
                 filt_coeffs=np.ascontiguousarray(t["filt_coeffs"], np.float32),
                 mel_cosine=np.ascontiguousarray(t["mel_cosine"], np.float32))
        lifter = np.ascontiguousarray(t["lifter"], np.float32) if "lifter" in t else None
        L = capi.lib()
        L.psgpu_fe_n_frames.restype = C.c_int64
        L.psgpu_fe_n_frames.argtypes = [C.c_void_p, C.c_int64]
        self.h = C.c_void_p()
        capi.check(L.psgpu_fe_create(C.byref(self.h), C.byref(p), _vp(k["hamming"]), _vp(k["ccc"]), _vp(k["sss"]),
                                     _vp(k["spec_start"]), _vp(k["filt_start"]), _vp(k["filt_width"]),
                                     _vp(k["filt_coeffs"]), _vp(k["mel_cosine"]),
                                     _vp(lifter) if lifter is not None else None), "psgpu_fe_create")
        self.out_dim = p.out_dim
        self.n_filt = p.n_filt

    def close(self):
        if self.h:
            capi.lib().psgpu_fe_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def n_frames(self, n_samples):
        return int(capi.lib().psgpu_fe_n_frames(self.h, int(n_samples)))

    def process_utts(self, pcms, noise=None, undefined=None):
        """pcms: list of int16 arrays (one per utterance).  Returns (cep [T][out_dim],
        frame_off [n_utt+1]).  noise [n_utt][4][n_filt] float64 + undefined [n_utt]
        int32: the noise tracker carried into / out of each utterance, updated in
        place; None = every utterance starts from reset statistics."""
        pcms = [np.ascontiguousarray(p, np.int16).reshape(-1) for p in pcms]
        off = np.zeros(len(pcms) + 1, np.int64)
        off[1:] = np.cumsum([p.size for p in pcms])
        pcm = np.concatenate(pcms) if pcms else np.zeros(0, np.int16)
        total = sum(self.n_frames(p.size) for p in pcms)
        cep = np.empty((total, self.out_dim), np.float32)
        fo = np.zeros(len(pcms) + 1, np.int32)
        if noise is not None:
            assert noise.dtype == np.float64 and noise.shape == (len(pcms), 4, self.n_filt) and noise.flags.c_contiguous
            assert undefined.dtype == np.int32 and undefined.shape == (len(pcms),)
        capi.check(capi.lib().psgpu_fe_process_utts(self.h, _vp(pcm), _vp(off), len(pcms),
                                                    _vp(noise) if noise is not None else None,
                                                    _vp(undefined) if noise is not None else None,
                                                    _vp(cep), _vp(fo)), "psgpu_fe_process_utts")
        return cep, fo
