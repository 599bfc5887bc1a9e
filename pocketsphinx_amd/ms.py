"""Host-side mirror of the reference's multi-stream / continuous scorer
(ms_mgau_model_t, reference src/ms_mgau.h:95-110).  ``MsMgau.frame_eval`` has
the argument meaning of ms_cont_mgau_frame_eval (src/ms_mgau.c:191-282);
``score_frames`` is the batched compallsen form.  Arithmetic: csrc/psgpu_ms.hip."""
import ctypes as C

import numpy as np

from . import capi


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class MsMgau:
    def __init__(self, tables, topn=None, aw=None):
        L = capi.lib()
        t = tables
        self.n_mgau = int(t["n_mgau"][0]); self.n_feat = int(t["n_feat"][0])
        self.n_density = int(t["n_density"][0]); self.n_sen = int(t["n_sen"][0])
        self.topn = min(int(topn if topn is not None else t["max_topn"][0]), self.n_density)
        self.aw = int(aw if aw is not None else t["aw"][0])
        featlen = np.ascontiguousarray(t["featlen"], np.int32)
        self.veclen = int(featlen.sum())
        h = C.c_void_p()
        capi.check(L.psgpu_ms_model_create(
            C.byref(h), self.n_mgau, self.n_feat, self.n_density, _p(featlen), self.n_sen, self.topn, self.aw,
            _p(np.ascontiguousarray(t["mean"], np.float32)), _p(np.ascontiguousarray(t["var"], np.float32)),
            _p(np.ascontiguousarray(t["det"], np.float32)), _p(np.ascontiguousarray(t["pdf"], np.uint8)),
            _p(np.ascontiguousarray(t["sen2mgau"], np.uint32)), _p(np.ascontiguousarray(t["logadd"], np.uint8)),
            int(t["logadd_size"][0]), int(t["logadd_width"][0]), int(t["log_zero"][0])),
            "psgpu_ms_model_create")
        self.h = h
        self.senscr = np.zeros(self.n_sen, np.int16)      # acmod->senone_scores: persists between calls

    def frame_eval(self, feat, active=None, compallsen=True, frame=-1):
        """One ms_cont_mgau_frame_eval call.  `frame` >= 0 lets the call be answered from
        the look-ahead cache when that frame was announced with this very vector."""
        feat = np.ascontiguousarray(feat, np.float32).reshape(-1)
        assert feat.size == self.veclen
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        capi.check(capi.lib().psgpu_ms_frame_eval_at(self.h, _p(self.senscr), _p(act),
                                                     0 if act is None else act.size, _p(feat), int(frame),
                                                     int(bool(compallsen))), "psgpu_ms_frame_eval_at")
        return self.senscr.copy()

    def lookahead(self, feats, frame0=0):
        feats = np.ascontiguousarray(feats, np.float32)
        assert feats.ndim == 2 and feats.shape[1] == self.veclen
        capi.check(capi.lib().psgpu_ms_lookahead(self.h, _p(feats), int(frame0), int(feats.shape[0])),
                   "psgpu_ms_lookahead")

    def lookahead_stats(self):
        a, b = C.c_int64(), C.c_int64()
        capi.check(capi.lib().psgpu_ms_lookahead_stats(self.h, C.byref(a), C.byref(b)), "psgpu_ms_lookahead_stats")
        return a.value, b.value

    def score_frames(self, feats):
        feats = np.ascontiguousarray(feats, np.float32)
        assert feats.ndim == 2 and feats.shape[1] == self.veclen
        out = np.empty((feats.shape[0], self.n_sen), np.int16)
        capi.check(capi.lib().psgpu_ms_score_batch(self.h, _p(feats), int(feats.shape[0]), _p(out)),
                   "psgpu_ms_score_batch")
        return out

    def close(self):
        if getattr(self, "h", None):
            capi.lib().psgpu_ms_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
