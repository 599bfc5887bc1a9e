"""Host-side mirror of the reference's HMM evaluation interface.

``HmmContext`` mirrors hmm_context_t (reference src/hmm.h:145-154): the shared
transition matrices and senone-sequence table, resident in HBM.
``HmmContext.vit_eval`` is hmm_vit_eval() (src/hmm.c:786-805) applied to a
whole array of HMMs for one frame; the arithmetic runs in
csrc/psgpu_hmm.hip.  HMM state travels as a numpy structured array with the
layout of psgpu_hmm_rec_t (include/psgpu.h).
"""
import ctypes as C

import numpy as np

from . import capi

MPX = 0x8000
WORST_SCORE = np.int32(-0x20000000)      # (int32)0xE0000000, hmm.h:84
BAD_SSID = 0xffff

HMM_REC = np.dtype([("score", np.int32, 5), ("history", np.int32, 5),
                    ("out_score", np.int32), ("out_history", np.int32),
                    ("bestscore", np.int32), ("senid", np.uint16, 5),
                    ("tmatid_mpx", np.uint16)])
assert HMM_REC.itemsize == 64


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HmmContext:
    def __init__(self, tp, sseq, n_sen):
        """tp uint8 [n_tmat][n_emit][n_emit+1]; sseq uint16 [n_sseq][n_emit]."""
        tp = np.ascontiguousarray(tp, np.uint8)
        sseq = np.ascontiguousarray(sseq, np.uint16)
        self.n_tmat, self.n_emit = int(tp.shape[0]), int(tp.shape[1])
        assert tp.shape[2] == self.n_emit + 1 and sseq.shape[1] == self.n_emit
        self.n_sen = int(n_sen)
        h = C.c_void_p()
        capi.check(capi.lib().psgpu_hmm_ctx_create(C.byref(h), self.n_emit, self.n_tmat, _p(tp),
                                                   int(sseq.shape[0]), _p(sseq), self.n_sen),
                   "psgpu_hmm_ctx_create")
        self.h = h

    def vit_eval(self, recs, senscr):
        """One frame for every record (updated in place).  Returns the best score."""
        assert recs.dtype == HMM_REC and recs.flags.c_contiguous
        senscr = np.ascontiguousarray(senscr, np.int16)
        assert senscr.size == self.n_sen
        best = C.c_int32()
        capi.check(capi.lib().psgpu_hmm_vit_eval(self.h, _p(recs), int(recs.size), _p(senscr),
                                                 C.byref(best)), "psgpu_hmm_vit_eval")
        return int(best.value)

    def close(self):
        if getattr(self, "h", None):
            capi.lib().psgpu_hmm_ctx_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
