"""Host-side mirror of the reference's semi-continuous scorer interface
(s2_semi_mgau_t, reference src/s2_semi_mgau.h:64-91).  ``SemiMgau.frame_eval``
has the argument meaning of s2_semi_mgau_frame_eval (src/s2_semi_mgau.c:836-883)
plus the caller's ``frame_idx``; the arithmetic runs in csrc/psgpu_semi.hip."""
import ctypes as C

import numpy as np

from . import capi


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class SemiMgau:
    def __init__(self, tables, topn=None, ds_ratio=None, topn_beam=None, n_topn_hist=None):
        L = capi.lib()
        t = tables
        self.n_feat = int(t["n_feat"][0]); self.n_density = int(t["n_density"][0])
        self.n_sen = int(t["n_sen"][0])
        self.topn = int(topn if topn is not None else t["max_topn"][0])
        self.ds_ratio = int(ds_ratio if ds_ratio is not None else t["ds_ratio"][0])
        self.n_hist = int(n_topn_hist if n_topn_hist is not None else t["n_fast_hist"][0])
        featlen = np.ascontiguousarray(t["featlen"], np.int32)
        self.veclen = int(featlen.sum())
        beam = np.ascontiguousarray(topn_beam if topn_beam is not None else t["topn_beam"], np.uint8)
        cb = np.ascontiguousarray(t["mixw_cb"], np.uint8) if "mixw_cb" in t else None
        m = C.c_void_p()
        capi.check(L.psgpu_semi_model_create(
            C.byref(m), self.n_feat, self.n_density, _p(featlen), self.n_sen, self.topn, self.ds_ratio,
            _p(beam), _p(np.ascontiguousarray(t["mean"], np.float32)),
            _p(np.ascontiguousarray(t["var"], np.float32)), _p(np.ascontiguousarray(t["det"], np.float32)),
            _p(np.ascontiguousarray(t["mixw"], np.uint8)), _p(cb),
            _p(np.ascontiguousarray(t["logadd8"], np.uint8)), int(t["logadd8"].size)),
            "psgpu_semi_model_create")
        self.m = m
        s = C.c_void_p()
        capi.check(L.psgpu_semi_state_create(C.byref(s), m, self.n_hist), "psgpu_semi_state_create")
        self.s = s
        self.frame_idx = 0

    def frame_eval(self, feat, frame, active=None, compallsen=True, frame_idx=None):
        feat = np.ascontiguousarray(feat, np.float32).reshape(-1)
        assert feat.size == self.veclen
        scr = np.empty(self.n_sen, np.int16)
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        capi.check(capi.lib().psgpu_semi_frame_eval(
            self.s, _p(scr), _p(act), 0 if act is None else act.size, _p(feat), int(frame),
            int(self.frame_idx if frame_idx is None else frame_idx), int(bool(compallsen))),
            "psgpu_semi_frame_eval")
        return scr

    def score_utts(self, feats, utt_lens):
        """Batched compallsen scoring of whole utterances (each from a fresh top-N state)."""
        feats = np.ascontiguousarray(feats, np.float32)
        off = np.zeros(len(utt_lens) + 1, np.int32)
        off[1:] = np.cumsum(np.asarray(utt_lens, np.int64))
        assert feats.ndim == 2 and feats.shape == (int(off[-1]), self.veclen)
        out = np.empty((feats.shape[0], self.n_sen), np.int16)
        capi.check(capi.lib().psgpu_semi_score_batch(self.m, _p(feats), _p(off), len(utt_lens), _p(out)),
                   "psgpu_semi_score_batch")
        return out

    def cur_topn(self, slot=-1):
        cw = np.empty((self.n_feat, self.topn), np.int32)
        sc = np.empty((self.n_feat, self.topn), np.int32)
        n = np.empty(self.n_feat, np.int32)
        capi.check(capi.lib().psgpu_semi_state_get_topn(self.s, int(slot), _p(cw), _p(sc), _p(n)),
                   "psgpu_semi_state_get_topn")
        return cw, sc, n

    def reset_hist(self):
        capi.check(capi.lib().psgpu_semi_state_reset(self.s), "psgpu_semi_state_reset")

    def close(self):
        L = capi.lib()
        if getattr(self, "s", None):
            L.psgpu_semi_state_free(self.s); self.s = None
        if getattr(self, "m", None):
            L.psgpu_semi_model_free(self.m); self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
