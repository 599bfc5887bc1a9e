"""Host-side mirror of the reference's PTM scorer interface.

``PtmModel`` owns the device copy of the tables ptm_mgau_init() builds
(reference src/ptm_mgau.c:804-896).  ``PtmMgau.score_utts`` is the batched
equivalent of calling ptm_mgau_frame_eval(..., compallsen=TRUE)
(src/ptm_mgau.c:408-454) for every frame of every utterance in order.
All arithmetic happens in the HIP kernels of libpsgpu.so.
"""
import ctypes as C

import numpy as np

from . import capi

RAW_SCORES = 1


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def expand_clustered_mixw(mixw4, mixw_cb, n_sen):
    """The one-byte-per-senone weights of a clustered sendump, as ptm_mgau_senone_eval looks them up (ptm_mgau.c:375-379):
        dcw = row[sen / 2];  dcw = (dcw & 1) ? dcw >> 4 : dcw & 0x0f;  weight = mixw_cb[dcw]
    The nibble is chosen by the low bit of the BYTE (not of the senone): both senones of a byte get the same weight.  A
    per-(stream, density, senone) constant, expanded once; integration/psgpu_mgau_shim.c does the same from ptm_mgau_t."""
    rows = np.ascontiguousarray(mixw4, np.uint8)
    dcw = np.where(rows & 1, rows >> 4, rows & 0x0f)
    w = np.ascontiguousarray(mixw_cb, np.uint8)[dcw]            # [n_feat][n_density][(n_sen + 1) / 2]
    return np.ascontiguousarray(np.repeat(w, 2, axis=-1)[..., :n_sen])


class PtmModel:
    def __init__(self, tables, topn=None, ds_ratio=None):
        L = capi.lib()
        t = tables
        if "mixw_cb" in t:
            t = dict(t)
            t["mixw"] = expand_clustered_mixw(t["mixw"], t["mixw_cb"], int(t["n_sen"][0]))
        self.n_mgau = int(t["n_mgau"][0]); self.n_feat = int(t["n_feat"][0])
        self.n_density = int(t["n_density"][0]); self.n_sen = int(t["n_sen"][0])
        self.topn = int(topn if topn is not None else t["max_topn"][0])
        self.ds_ratio = int(ds_ratio if ds_ratio is not None else t["ds_ratio"][0])
        featlen = np.ascontiguousarray(t["featlen"], np.int32)
        self.veclen = int(featlen.sum())
        self.n_chain = self.n_mgau * self.n_feat
        h = C.c_void_p()
        capi.check(L.psgpu_ptm_model_create(
            C.byref(h), self.n_mgau, self.n_feat, self.n_density, _p(featlen), self.n_sen,
            self.topn, self.ds_ratio,
            _p(np.ascontiguousarray(t["mean"], np.float32)),
            _p(np.ascontiguousarray(t["var"], np.float32)),
            _p(np.ascontiguousarray(t["det"], np.float32)),
            _p(np.ascontiguousarray(t["mixw"], np.uint8)),
            _p(np.ascontiguousarray(t["sen2cb"], np.uint8)),
            _p(np.ascontiguousarray(t["logadd8"], np.uint8)), int(t["logadd8"].size)),
            "psgpu_ptm_model_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            capi.lib().psgpu_ptm_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PtmMgau:
    """Batched PTM scorer (compallsen semantics) over host numpy buffers."""

    def __init__(self, model):
        self.m = model

    def score_utts(self, feats, utt_lens, seed_cw=None, raw_scores=False,
                   want_topn=True, want_scores=True):
        """feats [total_frames][veclen] fp32; utt_lens: frames per utterance.
        Returns dict(senscr int16 [T][n_sen], topn_cw uint8 [T][n_chain][topn],
        topn_score int32 same, best int32 [T], seed_cw)."""
        m = self.m
        feats = np.ascontiguousarray(feats, np.float32)
        off = np.zeros(len(utt_lens) + 1, np.int32)
        off[1:] = np.cumsum(np.asarray(utt_lens, np.int64))
        T = int(off[-1])
        if feats.shape != (T, m.veclen):
            raise ValueError("feats shape %r != (%d, %d)" % (feats.shape, T, m.veclen))
        scr = np.empty((T, m.n_sen), np.int16) if want_scores else None
        cw = np.empty((T, m.n_chain, m.topn), np.uint8) if want_topn else None
        sc = np.empty((T, m.n_chain, m.topn), np.int32) if want_topn else None
        best = np.empty(T, np.int32) if want_scores else None
        if seed_cw is not None:
            seed_cw = np.ascontiguousarray(seed_cw, np.uint8).copy()
            assert seed_cw.shape == (len(utt_lens), m.n_chain, m.topn)
        capi.check(capi.lib().psgpu_ptm_score_batch(
            m.h, _p(feats), _p(off), len(utt_lens), _p(seed_cw), _p(sc), _p(cw), _p(scr),
            _p(best), RAW_SCORES if raw_scores else 0), "psgpu_ptm_score_batch")
        return dict(senscr=scr, topn_cw=cw, topn_score=sc, best=best, seed_cw=seed_cw)


class PtmState:
    """Per-decoder scorer state: mirrors ptm_mgau_t's mutable part (history
    ring of top-N lists, src/ptm_mgau.h:68-97).  ``frame_eval`` has the
    argument meaning of ptm_mgau_frame_eval (src/ptm_mgau.c:408-454) plus the
    caller's ``frame_idx`` (ps_mgau_t.frame_idx, src/acmod.h:113-116)."""

    def __init__(self, model, n_fast_hist):
        self.m = model
        self.n_hist = int(n_fast_hist)
        h = C.c_void_p()
        capi.check(capi.lib().psgpu_ptm_state_create(C.byref(h), model.h, self.n_hist),
                   "psgpu_ptm_state_create")
        self.h = h
        self.frame_idx = 0

    def reset_hist(self):
        capi.check(capi.lib().psgpu_ptm_state_reset(self.h), "psgpu_ptm_state_reset")

    def frame_eval(self, feat, frame, active=None, compallsen=True, frame_idx=None):
        feat = np.ascontiguousarray(feat, np.float32).reshape(-1)
        assert feat.size == self.m.veclen
        scr = np.empty(self.m.n_sen, np.int16)
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        capi.check(capi.lib().psgpu_ptm_frame_eval(
            self.h, _p(scr), _p(act), 0 if act is None else act.size, _p(feat), int(frame),
            int(self.frame_idx if frame_idx is None else frame_idx), int(bool(compallsen))),
            "psgpu_ptm_frame_eval")
        return scr

    def lookahead(self, feats, frame0):
        """Announce upcoming frames (psgpu_ptm_state_lookahead)."""
        feats = np.ascontiguousarray(feats, np.float32).reshape(-1, self.m.veclen)
        capi.check(capi.lib().psgpu_ptm_state_lookahead(self.h, _p(feats), int(frame0), int(feats.shape[0])),
                   "psgpu_ptm_state_lookahead")

    def lookahead_stats(self):
        a, b = C.c_int64(), C.c_int64()
        capi.check(capi.lib().psgpu_ptm_state_lookahead_stats(self.h, C.byref(a), C.byref(b)), "lookahead_stats")
        return int(a.value), int(b.value)

    def cur_topn(self, slot=-1):
        cw = np.empty((self.m.n_chain, self.m.topn), np.int32)
        sc = np.empty((self.m.n_chain, self.m.topn), np.int32)
        capi.check(capi.lib().psgpu_ptm_state_get_topn(self.h, int(slot), _p(cw), _p(sc), None),
                   "psgpu_ptm_state_get_topn")
        return cw, sc

    def close(self):
        if getattr(self, "h", None):
            capi.lib().psgpu_ptm_state_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
