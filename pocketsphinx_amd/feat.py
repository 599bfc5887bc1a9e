"""Host-side mirror of the reference's whole-utterance dynamic feature
computation (feat_s2mfc2feat_live(begin, end), reference src/feat/feat.c:1310)
for the "1s_c_d_dd" type with batch CMN; arithmetic in csrc/psgpu_feat.hip."""
import ctypes as C

import numpy as np

from . import capi


def dynfeat_1s_c_d_dd(cep, utt_lens):
    """cep [T][cepsize] fp32 (utterances back to back) -> features [T][3*cepsize]."""
    cep = np.ascontiguousarray(cep, np.float32)
    off = np.zeros(len(utt_lens) + 1, np.int32)
    off[1:] = np.cumsum(np.asarray(utt_lens, np.int64))
    if cep.shape[0] != int(off[-1]):
        raise ValueError("cep has %d frames, utt_lens sum to %d" % (cep.shape[0], int(off[-1])))
    out = np.empty((cep.shape[0], 3 * cep.shape[1]), np.float32)
    capi.check(capi.lib().psgpu_feat_1s_c_d_dd(cep.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                               len(utt_lens), int(cep.shape[1]),
                                               out.ctypes.data_as(C.c_void_p)), "psgpu_feat_1s_c_d_dd")
    return out


class FeatType:
    """Mirror of feat_init(type, cmn, varnorm, agc, ..., cepsize) (reference src/feat/feat.c:704-915) + feat_read_lda / feat_set_subvecs
    for whole utterances: every feature type the reference parses, batch CMN ("current" / "batch") with or without unit variance, agc
    none / max, a linear transform [lda_out][dim] and a subvector list (the components in -svspec's order).  compute() = what
    feat_s2mfc2feat_live(begin = end = TRUE) returns, the streams / subvectors of a frame side by side."""

    def __init__(self, type_name, cepsize=13, cmn="batch", varnorm=False, agc="none", lda=None, subvec=None):
        if cmn not in ("none", "batch", "current") or agc not in ("none", "max"):
            raise ValueError("cmn none / batch / current and agc none / max are served")
        self._lda = np.ascontiguousarray(lda, np.float32) if lda is not None else None
        self._sv = np.ascontiguousarray(subvec, np.int32).reshape(-1) if subvec is not None else None
        h = C.c_void_p()
        L = capi.lib()
        capi.check(L.psgpu_feat_create(C.byref(h), str(type_name).encode(), int(cepsize), 0 if cmn == "none" else 1, int(bool(varnorm)),
                                       0 if agc == "none" else 1,
                                       self._lda.ctypes.data_as(C.c_void_p) if self._lda is not None else None,
                                       0 if self._lda is None else int(self._lda.shape[0]), 0 if self._lda is None else int(self._lda.shape[1]),
                                       self._sv.ctypes.data_as(C.c_void_p) if self._sv is not None else None,
                                       0 if self._sv is None else int(self._sv.size)), "psgpu_feat_create")
        self.h = h
        L.psgpu_feat_out_dim.argtypes = [C.c_void_p]; L.psgpu_feat_cepsize.argtypes = [C.c_void_p]
        self.out_dim = int(L.psgpu_feat_out_dim(h)); self.cepsize = int(L.psgpu_feat_cepsize(h))

    def compute(self, cep, utt_lens):
        cep = np.ascontiguousarray(cep, np.float32)
        off = np.zeros(len(utt_lens) + 1, np.int32)
        off[1:] = np.cumsum(np.asarray(utt_lens, np.int64))
        if cep.shape != (int(off[-1]), self.cepsize):
            raise ValueError("cep shape %r != (%d, %d)" % (cep.shape, int(off[-1]), self.cepsize))
        out = np.empty((cep.shape[0], self.out_dim), np.float32)
        capi.check(capi.lib().psgpu_feat_compute(self.h, cep.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), len(utt_lens),
                                                 out.ctypes.data_as(C.c_void_p)), "psgpu_feat_compute")
        return out

    def close(self):
        if getattr(self, "h", None):
            capi.lib().psgpu_feat_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
