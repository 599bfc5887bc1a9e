"""Host-side mirror of the batch first pass (include/psgpu.h, psgpu_decode_*): PCM of a batch of utterances ->
hypotheses, every stage on the MI355X (csrc/psgpu_decode.hip).  The device side of ps_decode_raw() with
-fwdflat no -bestpath no (reference src/pocketsphinx.c:1030-1070) followed by ngram_search_bp_hyp
(src/ngram_search.c:546-581) for a batch."""
import ctypes as C

import numpy as np

from . import capi
from .fe import FrontEnd
from .hmm import HmmContext
from .ptm import PtmModel
from .search import FwdtreeSearch


class _PlPar(C.Structure):
    _fields_ = [("n_phones", C.c_int32), ("window", C.c_int32), ("beam", C.c_int32), ("pbeam", C.c_int32),
                ("pip", C.c_int32), ("penalty_weight", C.c_double)]


class _Config(C.Structure):
    _fields_ = [("fe", C.c_void_p), ("model", C.c_void_p), ("ctx", C.c_void_p), ("ft", C.c_void_p), ("pl", _PlPar),
                ("pl_ssid", C.c_void_p), ("pl_tmatid", C.c_void_p), ("ci_list", C.c_void_p), ("n_ci_list", C.c_int32),
                ("pl_window", C.c_int32), ("max_words", C.c_int32), ("scorer_kind", C.c_int32), ("scorer", C.c_void_p)]


SCORER_PTM, SCORER_SEMI, SCORER_MS = 0, 1, 2


class DecodeView(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_utt", "total_frames", "max_frames", "bp_cap", "bss_cap", "max_words")] + \
               [(n, C.c_void_p) for n in ("frame_off", "frame_off_dev", "feat_dev", "topn_cw_dev", "rows_dev", "penalties_dev",
                                          "bp_dev", "bss_dev", "idx_dev", "step_dev", "result_dev", "hyp_dev", "hyp_n_dev",
                                          "w1_ssid_dev", "topn_score_dev")]


def ci_senone_list(sseq, pl_ssid, n_sen):
    """the senone list acmod_flags2list (reference src/acmod.c:1223-1275) builds when every CI phone is active: the
    listed senones in order, gaps > 255 bridged by extra entries"""
    fl = np.zeros(n_sen, bool)
    fl[np.asarray(sseq)[np.asarray(pl_ssid)].reshape(-1)] = True
    out, last = [], 0
    for s in np.nonzero(fl)[0]:
        while s - last > 255:
            last += 255; out.append(last)
        out.append(int(s)); last = int(s)
    return np.array(out, np.uint16)


class DecodePipeline:
    """fe_tables: FrontEnd tables; ptm_tables: PtmModel tables; static / par: FwdtreeSearch tables; trace: the phone
    loop's parameters as `ref_dump fwdtree` writes them (pl_par = n_phones, window, beam, pbeam, pip, pl_window;
    pl_weight; pl_ssid; pl_tmat); lm: an NGramTrieLM or None (dense table in `static`)."""

    def __init__(self, fe_tables, ptm_tables, static, par, trace, lm=None, max_words=512, scorer=None):
        """scorer: None -- the PTM scorer built from ptm_tables; or a SemiMgau / MsMgau object (then ptm_tables is not used):
        the scorers acmod_init_am falls back to / is sent to by -senmgau (reference src/acmod.c:62-130).  fe_tables None: no
        front end -- feature vectors come from the caller (run_feat), e.g. the s2_4x vectors of a semi-continuous model."""
        from .ms import MsMgau
        from .semi import SemiMgau
        self.fe = FrontEnd(fe_tables) if fe_tables is not None else None
        self.scorer = scorer
        if scorer is None:
            self.model = PtmModel(ptm_tables)
            kind, sh, n_sen = SCORER_PTM, None, self.model.n_sen
        elif isinstance(scorer, SemiMgau):
            self.model = None
            kind, sh, n_sen = SCORER_SEMI, scorer.m, scorer.n_sen
        elif isinstance(scorer, MsMgau):
            self.model = None
            kind, sh, n_sen = SCORER_MS, scorer.h, scorer.n_sen
        else:
            raise TypeError("scorer: None, a SemiMgau or an MsMgau")
        self.n_sen = n_sen
        self.veclen = scorer.veclen if scorer is not None else None
        self.search = FwdtreeSearch(static, par, lm=lm)
        self.ctx = HmmContext(static["tp"], static["sseq"], n_sen)
        self.max_words = int(max_words)
        pl = [int(v) for v in trace["pl_par"]]
        ssid = np.ascontiguousarray(trace["pl_ssid"], np.uint16)
        tmat = np.ascontiguousarray(trace["pl_tmat"], np.int16)
        cil = ci_senone_list(static["sseq"], trace["pl_ssid"], n_sen)
        cfg = _Config(self.fe.h if self.fe is not None else None, self.model.h if self.model is not None else None, self.ctx.h, self.search.h,
                      _PlPar(pl[0], pl[1], pl[2], pl[3], pl[4], float(trace["pl_weight"][0])),
                      ssid.ctypes.data, tmat.ctypes.data, cil.ctypes.data, int(cil.size), pl[5], self.max_words, kind, sh)
        self._prev, self._next = None, []
        self.h = C.c_void_p()
        capi.check(capi.lib().psgpu_decode_create(C.byref(self.h), C.byref(cfg)), "psgpu_decode_create")
        self.n_utt = 0

    def close(self):
        if getattr(self, "h", None):
            for q in list(getattr(self, "_next", [])):       # objects whose searches wait for this one's: unlinked first
                if getattr(q, "h", None):
                    capi.lib().psgpu_decode_search_after(q.h, None)
                q._prev = None
            self._next = []
            if getattr(self, "_prev", None) is not None and self in self._prev._next:
                self._prev._next.remove(self)
            self._prev = None
            capi.lib().psgpu_decode_free(self.h)
            self.h = None
            self.search.close(); self.ctx.close()
            if self.model is not None:
                self.model.close()
            if self.fe is not None:
                self.fe.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_feat(self, feat):
        """psgpu_decode_set_feat: the feature type (a FeatType, or None for en-us's 1s_c_d_dd) the pipeline computes from its front end's
        cepstra: the semi-continuous models' s2_4x, a model with -lda ..."""
        self._feat = feat                                  # (the C object keeps a bare pointer)
        capi.check(capi.lib().psgpu_decode_set_feat(self.h, feat.h if feat is not None else None), "psgpu_decode_set_feat")

    def search_after(self, prev):
        """This object's search waits (on the device) for the search of `prev`'s latest call -- two objects taking turns, one
        batch's front end and scorer beside the other's search: see psgpu_decode_search_after."""
        capi.check(capi.lib().psgpu_decode_search_after(self.h, prev.h if prev is not None else None), "psgpu_decode_search_after")
        # (the C object keeps a bare pointer to prev: hold it here so that it cannot be collected first, and let go of
        #  whoever waits on this object when it closes)
        if getattr(self, "_prev", None) is not None and self in self._prev._next:
            self._prev._next.remove(self)
        self._prev = prev
        if prev is not None:
            prev._next.append(self)

    def wait_scored(self):
        capi.check(capi.lib().psgpu_decode_wait_scored(self.h), "psgpu_decode_wait_scored")

    def stage_timing(self, on=True):
        capi.check(capi.lib().psgpu_decode_stage_timing(self.h, int(bool(on))), "psgpu_decode_stage_timing")

    def last_stage_ms(self):
        ms = (C.c_float * 6)()
        capi.check(capi.lib().psgpu_decode_last_stage_ms(self.h, ms), "psgpu_decode_last_stage_ms")
        return dict(zip(("front_end", "features", "scorer", "phone_loop", "search", "search_wait"), [float(v) for v in ms]))

    def score_mode(self, lists):
        """psgpu_decode_score_mode: True -- no score rows, the phone loop and the search score the senones they list"""
        capi.check(capi.lib().psgpu_decode_score_mode(self.h, int(bool(lists))), "psgpu_decode_score_mode")

    def table_capacity(self, bp_per_frame=0, bss_per_frame=0, auto_grow=True):
        """psgpu_decode_table_capacity: back-pointer / score-stack entries per frame of the longest utterance (0: keep);
        auto_grow: a full table makes fetch() repeat the search with doubled tables, as the reference grows its own"""
        capi.check(capi.lib().psgpu_decode_table_capacity(self.h, int(bp_per_frame), int(bss_per_frame), int(bool(auto_grow))),
                   "psgpu_decode_table_capacity")

    def tables_grown(self):
        return int(capi.lib().psgpu_decode_tables_grown(self.h))

    def session(self, on=True):
        """psgpu_decode_session: on -- every following one-utterance call continues the decoder session (the scorer's seeding
        history slot, the multiplexed permanent channels' per-state ssids); calling it again forgets the state"""
        capi.check(capi.lib().psgpu_decode_session(self.h, int(bool(on))), "psgpu_decode_session")

    def run_dev(self, pcm_dev, samp_off, stream=None):
        """pcm_dev: torch int16 tensor on the device (utterances back to back), samp_off: int64 numpy [n_utt + 1].
        Asynchronous on `stream` (default: torch's current stream)."""
        import torch
        samp_off = np.ascontiguousarray(samp_off, np.int64)
        self._keep = (pcm_dev, samp_off)
        st = C.c_void_p(stream if stream is not None else torch.cuda.current_stream().cuda_stream)
        self.n_utt = int(samp_off.size - 1)
        capi.check(capi.lib().psgpu_decode_first_pass_dev(self.h, C.c_void_p(pcm_dev.data_ptr()), samp_off.ctypes.data_as(C.c_void_p),
                                                          self.n_utt, st), "psgpu_decode_first_pass_dev")
        self._stream = st

    def compallsen(self, on=True):
        """psgpu_decode_compallsen: -compallsen yes -- rows normalised over all senones, taken as final by the phone loop and the search"""
        capi.check(capi.lib().psgpu_decode_compallsen(self.h, int(bool(on))), "psgpu_decode_compallsen")

    def front_end_ahead(self, pcm_dev, samp_off):
        """psgpu_decode_front_end_ahead: the NEXT call's front end on the object's own stream, while its latest search is still
        running (two objects taking turns: search_after).  The next run_dev with the same pcm_dev / samp_off skips its front end.
        Returns whether it was issued (only for an input of the latest call's shape)."""
        samp_off = np.ascontiguousarray(samp_off, np.int64)
        self._keep_ahead = (pcm_dev, samp_off)
        started = C.c_int32(0)
        capi.check(capi.lib().psgpu_decode_front_end_ahead(self.h, C.c_void_p(pcm_dev.data_ptr()), samp_off.ctypes.data_as(C.c_void_p),
                                                           int(samp_off.size - 1), C.byref(started)), "psgpu_decode_front_end_ahead")
        return bool(started.value)

    def run(self, pcms, stream=None):
        """pcms: list of int16 numpy arrays (host)."""
        import torch
        pcms = [np.ascontiguousarray(p, np.int16).reshape(-1) for p in pcms]
        n = len(pcms)
        ptrs = (C.c_void_p * max(n, 1))(*[p.ctypes.data for p in pcms])
        lens = (C.c_size_t * max(n, 1))(*[p.size for p in pcms])
        st = C.c_void_p(stream if stream is not None else torch.cuda.current_stream().cuda_stream)
        self.n_utt = n
        capi.check(capi.lib().psgpu_decode_first_pass(self.h, ptrs, lens, n, st), "psgpu_decode_first_pass")
        self._stream = st

    def run_feat(self, feats, utt_lens, stream=None):
        """psgpu_decode_first_pass_feat: feature vectors from the host, [total][veclen] float32 as feat_s2mfc2feat_live leaves
        them in acmod->feat_buf, utterances back to back (utt_lens frames each)"""
        import torch
        feats = np.ascontiguousarray(feats, np.float32)
        off = np.zeros(len(utt_lens) + 1, np.int32); off[1:] = np.cumsum(np.asarray(utt_lens, np.int64))
        assert feats.ndim == 2 and feats.shape[0] == int(off[-1])
        st = C.c_void_p(stream if stream is not None else torch.cuda.current_stream().cuda_stream)
        self.n_utt = len(utt_lens)
        capi.check(capi.lib().psgpu_decode_first_pass_feat(self.h, feats.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                                           self.n_utt, st), "psgpu_decode_first_pass_feat")
        self._stream = st

    def search_lag(self, lag):
        """psgpu_decode_search_lag: the next run*'s search stops `lag` frames short of every utterance's end"""
        capi.check(capi.lib().psgpu_decode_search_lag(self.h, int(lag)), "psgpu_decode_search_lag")

    def live_begin(self, max_frames, stream=None):
        """psgpu_decode_live_begin: one utterance in progress (session mode), at most max_frames frames"""
        import torch
        st = C.c_void_p(stream if stream is not None else torch.cuda.current_stream().cuda_stream)
        self.n_utt = 1
        self._stream = st
        capi.check(capi.lib().psgpu_decode_live_begin(self.h, int(max_frames), st), "psgpu_decode_live_begin")

    def live_restart(self, max_frames):
        """psgpu_decode_live_restart: the live utterance begins again with a larger capacity (its frames are fed again)"""
        capi.check(capi.lib().psgpu_decode_live_restart(self.h, int(max_frames), self._stream), "psgpu_decode_live_restart")

    def live_step(self, feats, lag):
        """psgpu_decode_live_step: feats [n_new][veclen] float32 more frames (may be empty); the search goes on up to `lag` frames
        short of the frames so far (0: to the utterance's end).  fetch() / tables() as after a run_feat of the frames so far."""
        feats = np.ascontiguousarray(feats, np.float32)
        n = int(feats.shape[0]) if feats.ndim == 2 else 0
        capi.check(capi.lib().psgpu_decode_live_step(self.h, feats.ctypes.data_as(C.c_void_p) if n else None, n, int(lag), self._stream),
                   "psgpu_decode_live_step")

    def streams_begin(self, n_streams, max_frames, max_step_frames, stream=None):
        """psgpu_decode_streams_begin: n_streams utterances in progress at once (new decoders), each growing at its own pace"""
        import torch
        st = C.c_void_p(stream if stream is not None else torch.cuda.current_stream().cuda_stream)
        self.n_utt = int(n_streams)
        self._stream = st
        capi.check(capi.lib().psgpu_decode_streams_begin(self.h, int(n_streams), int(max_frames), int(max_step_frames), st),
                   "psgpu_decode_streams_begin")

    def streams_step(self, feats, final=None):
        """psgpu_decode_streams_step: feats = one [n_new][veclen] float32 array per stream (empty: nothing this step); final = per stream
        whether its utterance ends with these frames.  fetch() / tables() afterwards: every stream's results as they stand."""
        n = self.n_utt
        assert len(feats) == n
        cnt = np.array([int(f.shape[0]) if f is not None and f.ndim == 2 else 0 for f in feats], np.int32)
        parts = [np.ascontiguousarray(f, np.float32) for f, c in zip(feats, cnt) if c]
        allf = np.concatenate(parts) if parts else np.zeros((0, 1), np.float32)
        fin = np.zeros(n, np.uint8) if final is None else np.array([1 if x else 0 for x in final], np.uint8)
        capi.check(capi.lib().psgpu_decode_streams_step(self.h, allf.ctypes.data_as(C.c_void_p) if parts else None,
                                                        cnt.ctypes.data_as(C.c_void_p), fin.ctypes.data_as(C.c_void_p), self._stream),
                   "psgpu_decode_streams_step")

    def streams_pcm_begin(self, n_streams, max_frames, max_step_frames, cmninit=(40.0, 3.0, -1.0), grow_feat=True, stream=None):
        """psgpu_decode_streams_pcm_begin: the streams fed with audio -- a batch of live decoders from ps_process_raw(full_utt = FALSE) on.
        cmninit: the reference's -cmninit (its default); grow_feat: the reference decoder's acmod_set_grow (-fwdflat yes)."""
        import torch
        st = C.c_void_p(stream if stream is not None else torch.cuda.current_stream().cuda_stream)
        self.n_utt = int(n_streams)
        self._stream = st
        ci = np.ascontiguousarray(cmninit, np.float32)
        capi.check(capi.lib().psgpu_decode_streams_pcm_begin(self.h, int(n_streams), int(max_frames), int(max_step_frames),
                                                             ci.ctypes.data_as(C.c_void_p), int(ci.size), int(bool(grow_feat)), st),
                   "psgpu_decode_streams_pcm_begin")

    def streams_step_pcm(self, pcms, final=None):
        """psgpu_decode_streams_step_pcm: pcms = one int16 array per stream (None / empty: nothing this step) -- one ps_process_raw call
        each; final = per stream whether its utterance ends (ps_end_utt).  Returns the feature frames every stream gained."""
        n = self.n_utt
        assert len(pcms) == n
        cnt = np.array([int(x.size) if x is not None else 0 for x in pcms], np.int64)
        parts = [np.ascontiguousarray(x, np.int16).reshape(-1) for x, c in zip(pcms, cnt) if c]
        allp = np.concatenate(parts) if parts else np.zeros(1, np.int16)
        fin = np.zeros(n, np.uint8) if final is None else np.array([1 if x else 0 for x in final], np.uint8)
        gained = np.zeros(n, np.int32)
        capi.check(capi.lib().psgpu_decode_streams_step_pcm(self.h, allp.ctypes.data_as(C.c_void_p) if parts else None,
                                                            cnt.ctypes.data_as(C.c_void_p), fin.ctypes.data_as(C.c_void_p),
                                                            gained.ctypes.data_as(C.c_void_p), self._stream), "psgpu_decode_streams_step_pcm")
        return gained

    def streams_restart(self, u):
        capi.check(capi.lib().psgpu_decode_streams_restart(self.h, int(u), self._stream), "psgpu_decode_streams_restart")

    def streams_next_utt(self, u):
        """psgpu_decode_streams_next_utt: stream u's decoder goes on to its next utterance (the session's carry-over, per stream)"""
        capi.check(capi.lib().psgpu_decode_streams_next_utt(self.h, int(u), self._stream), "psgpu_decode_streams_next_utt")

    def live_frames_searched(self):
        f = capi.lib().psgpu_decode_live_frames_searched
        f.restype = C.c_int64
        return int(f(self.h))

    def fetch(self, want_hyp=True):
        """(hyp_n [n][4], hyp [n][max_words][4] or None, result [n][8]) on the host; waits for the pipeline."""
        n = self.n_utt
        hn = np.zeros((n, 4), np.int32); res = np.zeros((n, 8), np.int32)
        hyp = np.zeros((n, self.max_words, 4), np.int32) if want_hyp else None
        capi.check(capi.lib().psgpu_decode_fetch_hyps(self.h, hn.ctypes.data_as(C.c_void_p),
                                                      hyp.ctypes.data_as(C.c_void_p) if want_hyp else None,
                                                      res.ctypes.data_as(C.c_void_p), self._stream), "psgpu_decode_fetch_hyps")
        return hn, hyp, res

    def second_pass(self, flat):
        """psgpu_decode_second_pass: the flat-lexicon second pass (a FwdflatSearch built from the same tables) on what the latest
        run* left in the object; fetch() / tables() then return the second pass's hypotheses, result records and tables"""
        capi.check(capi.lib().psgpu_decode_second_pass(self.h, flat.h, self._stream), "psgpu_decode_second_pass")

    def view(self):
        v = DecodeView()
        capi.check(capi.lib().psgpu_decode_view(self.h, C.byref(v)), "psgpu_decode_view")
        return v

    def tables(self, u, res):
        """utterance u's back-pointer table [n][10], score stack and frame marks on the host"""
        nb, nh, nfr = int(res[u, 0]), int(res[u, 1]), int(res[u, 2])
        bp = np.zeros((10, max(nb, 1)), np.int32); bss = np.zeros(max(nh, 1), np.int32); idx = np.zeros(nfr + 1, np.int32)
        capi.check(capi.lib().psgpu_decode_fetch_tables(self.h, int(u), nb, nh, nfr + 1, bp.ctypes.data_as(C.c_void_p),
                                                        bss.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), self._stream),
                   "psgpu_decode_fetch_tables")
        return dict(bp=bp[:, :nb].T.copy(), bscore_stack=bss[:nh].copy(), bp_table_idx=idx, n_frame=nfr, status=int(res[u, 3]))


def dedicated_stream():
    """A raw hipStream_t with a hardware queue of its own (psgpu_stream_create_dedicated); free with free_stream."""
    h = C.c_void_p()
    capi.check(capi.lib().psgpu_stream_create_dedicated(C.byref(h)), "psgpu_stream_create_dedicated")
    return h.value


def free_stream(h):
    capi.check(capi.lib().psgpu_stream_destroy(C.c_void_p(h)), "psgpu_stream_destroy")
