"""TEST INFRASTRUCTURE: builds and drives tests/hostsim -- the search and language-model kernel SOURCES of
pocketsphinx_amd/csrc compiled with g++ against a workgroup simulator (one fiber per work-item) -- so that
the kernels' logic (barrier structure, prefix sums, list orders, table contents) is checked against the
reference goldens on machines without a GPU.  "Device" buffers are numpy arrays.  Not a product path:
pocketsphinx_amd never loads this library."""
import ctypes as C
import os
import subprocess

import numpy as np

from pocketsphinx_amd.lm import NGramTrieLM
from pocketsphinx_amd.search import _DT, _NAMES, _Tables

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SIM_DIR = os.path.join(HERE, "hostsim")
CSRC = os.path.join(ROOT, "pocketsphinx_amd", "csrc")
LIB = os.path.join(SIM_DIR, "_build", "libpsgpu_hostsim.so")
SOURCES = [os.path.join(SIM_DIR, "hipsim.cc"), os.path.join(CSRC, "psgpu_search.hip"), os.path.join(CSRC, "psgpu_lm.hip"), os.path.join(CSRC, "psgpu_flat.hip")]
DEPS = SOURCES + [os.path.join(SIM_DIR, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "psgpu.h")] + \
    [os.path.join(CSRC, h) for h in ("psgpu_internal.h", "psgpu_hmm_dev.h", "psgpu_lm_dev.h")]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    def stale():
        return not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS)
    if stale():
        import fcntl
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        with open(LIB + ".lock", "w") as lk:            # parallel test workers: one builds, the others wait
            fcntl.flock(lk, fcntl.LOCK_EX)
            if stale():
                tmp = "%s.%d.tmp" % (LIB, os.getpid())
                subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-ffp-contract=off",
                                       "-Wall", "-Wno-unused-variable", "-Wno-unknown-pragmas", "-Wno-int-in-bool-context",
                                       "-I" + SIM_DIR, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", tmp] + SOURCES)
                os.replace(tmp, LIB)
    L = C.CDLL(LIB)
    L.psgpu_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib().psgpu_last_error().decode()))


class SimLm(NGramTrieLM):
    """NGramTrieLM on the simulator (same table marshalling, the other library)."""

    def __init__(self, g, lw=None, log_wip=None):
        import pocketsphinx_amd.lm as lm_mod
        real = lm_mod.capi
        lm_mod.capi = _SimCapi
        try:
            NGramTrieLM.__init__(self, g, lw, log_wip)
        finally:
            lm_mod.capi = real

    def close(self):
        if self.h:
            lib().psgpu_lm_free(self.h)
            self.h = C.c_void_p()

    def tg_score(self, queries):
        q = np.ascontiguousarray(queries, np.int32)
        w3, w2, w1 = (np.ascontiguousarray(q[:, i]) for i in range(3))
        n = q.shape[0]
        sc = np.empty(n, np.int32); nu = np.empty(n, np.int32)
        p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        check(lib().psgpu_lm_tg_score_dev(self.h, p(w3), p(w2), p(w1), C.c_int64(n), p(sc), p(nu), None), "psgpu_lm_tg_score_dev")
        return sc, nu


class _SimCapi:
    lib = staticmethod(lib)
    check = staticmethod(check)


class SimFwdtreeSearch:
    """pocketsphinx_amd.search.FwdtreeSearch on the simulator."""

    def __init__(self, static, par, lm=None):
        src = dict(static); src["par"] = par
        self._keep = {n: np.ascontiguousarray(src[n], _DT.get(n, np.int32)) for n in _NAMES if not (n == "lm" and lm is not None)}
        t = _Tables(*[self._keep[n].ctypes.data if n in self._keep else None for n in _NAMES],
                    int(self._keep["tp"].shape[0]), int(self._keep["sseq"].shape[0]))
        self.h = C.c_void_p()
        check(lib().psgpu_fwdtree_create(C.byref(self.h), C.byref(t)), "psgpu_fwdtree_create")
        self.lm = lm
        if lm is not None:
            check(lib().psgpu_fwdtree_set_lm(self.h, lm.h), "psgpu_fwdtree_set_lm")
        self.n_sen = int(par[2]); self.n_ci = int(par[0]); self.n1 = int(par[6]); self.n_emit = int(par[1])
        self.searched = []

    def close(self):
        if self.h:
            lib().psgpu_fwdtree_free(self.h)
            self.h = C.c_void_p()

    def search(self, senscr, penalties, utt_lens, bp_cap=16384, bss_cap=1 << 19, raw_scores=False, pl_window=0, handover=None,
               mpx_in=None, mpx_out=None, cuts=None, lag=0):
        """handover: a dict that receives the buffers a second pass takes over (bp [n][10][cap], result [n][8], w1_ssid);
        mpx_in / mpx_out: the session carry-over, as FwdtreeSearch.search"""
        off = np.zeros(len(utt_lens) + 1, np.int32); off[1:] = np.cumsum(utt_lens)
        n = len(utt_lens); mf = int(max(utt_lens)) if n else 0
        d_s = np.ascontiguousarray(senscr, np.int16); d_p = np.ascontiguousarray(penalties, np.int32)
        assert d_s.shape == (int(off[-1]), self.n_sen) and d_p.shape == (int(off[-1]), self.n_ci)
        bp = np.zeros((n, 10, bp_cap), np.int32); bss = np.zeros((n, bss_cap), np.int32)
        idx = np.zeros((n, mf + 2), np.int32); step = np.zeros((n, max(mf, 1), 4), np.int32); res = np.zeros((n, 8), np.int32)
        p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        w1 = None
        if handover is not None:
            w1 = np.zeros((n, self.n1, self.n_emit), np.int32)
            handover.update(bp=bp, result=res, w1_ssid=w1, bp_cap=bp_cap)
        lib().psgpu_fwdtree_n_mpx_channels.argtypes = [C.c_void_p]
        n_mpx = int(lib().psgpu_fwdtree_n_mpx_channels(self.h))
        mi = None if mpx_in is None else np.ascontiguousarray(mpx_in, np.int32).reshape(n, n_mpx, self.n_emit)
        mo = None if mpx_out is None else np.zeros((n, n_mpx, self.n_emit), np.int32)
        # cuts: one utterance searched in several calls (psgpu_fwdtree_search_resume) -- up to each cut's frame count minus `lag`, then
        # to the end: same buffers, same tables
        calls = [(off, 0, 0)] if cuts is None else \
            [(np.array([0, c], np.int32), lag, (1 if i == 0 else 3)) for i, c in enumerate(cuts)] + [(off, 0, 2)]
        for o, lg, mode in calls:
            if cuts is not None:
                check(lib().psgpu_fwdtree_search_lag(self.h, int(lg)), "psgpu_fwdtree_search_lag")
                check(lib().psgpu_fwdtree_search_resume(self.h, mode), "psgpu_fwdtree_search_resume")
            check(lib().psgpu_fwdtree_search_session_dev(self.h, p(d_s), C.c_int64(self.n_sen), p(d_p), p(o), n, mf, bp_cap, bss_cap,
                                                         p(bp), p(bss), p(idx), p(step), p(res), int(raw_scores), int(pl_window),
                                                         p(w1) if w1 is not None else None, p(mi) if mi is not None else None,
                                                         p(mo) if mo is not None else None, None),
                  "psgpu_fwdtree_search_session_dev")
            if cuts is not None:
                self.searched.append(int(res[0, 2]))
        if mpx_out is not None:
            mpx_out["mpx"] = mo
        self.last = dict(bp=bp, idx=idx, res=res, mf=mf, bp_cap=bp_cap)
        out = []
        for u in range(n):
            nb, nh, nfr, status = [int(v) for v in res[u, :4]]
            out.append(dict(bp=bp[u, :, :nb].T.copy(), bscore_stack=bss[u, :nh].copy(), bp_table_idx=idx[u, :nfr + 1].copy(),
                            step=step[u, :nfr].copy(), n_frame=nfr, status=status))
        return out


def search_windows(s, senscr, penalties, cuts, lag, bp_cap=16384, bss_cap=1 << 19, raw_scores=False, pl_window=0):
    """ONE utterance as a live stream keeps it (psgpu_fwdtree_search_streams + _resume): every call is handed a buffer that holds ONLY
    the score rows and penalties from the frame its search resumes at -- the rows of frames already searched are gone -- placed by a
    start BEFORE the buffer (utt_off = -first frame kept), and {frames scored, frame to search to}.  cuts: frames scored at each call
    (the last = all); the search runs `lag` behind them and to the end in the last call.  Returns the result dict and the frames
    searched after every call.  (Under AddressSanitizer a read of a dropped row is a report.)"""
    T = int(cuts[-1])
    d_s = np.ascontiguousarray(senscr, np.int16); d_p = np.ascontiguousarray(penalties, np.int32)
    bp = np.zeros((1, 10, bp_cap), np.int32); bss = np.zeros((1, bss_cap), np.int32)
    idx = np.zeros((1, T + 2), np.int32); step = np.zeros((1, max(T, 1), 4), np.int32); res = np.zeros((1, 8), np.int32)
    p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    S = 0; searched = []
    for k, c in enumerate(cuts):
        last = k == len(cuts) - 1
        to = c if last else max(c - lag, S)
        win_s = np.ascontiguousarray(d_s[S:c]); win_p = np.ascontiguousarray(d_p[S:c])       # what a stream still holds
        off = np.array([-S, 0], np.int32); ext = np.array([c, to, 0], np.int32)
        check(lib().psgpu_fwdtree_search_streams(s.h, p(ext)), "psgpu_fwdtree_search_streams")
        check(lib().psgpu_fwdtree_search_resume(s.h, 1 if k == 0 else 3), "psgpu_fwdtree_search_resume")
        check(lib().psgpu_fwdtree_search_session_dev(s.h, p(win_s), C.c_int64(s.n_sen), p(win_p), p(off), 1, T, bp_cap, bss_cap,
                                                     p(bp), p(bss), p(idx), p(step), p(res), int(raw_scores), int(pl_window), None, None, None, None),
              "psgpu_fwdtree_search_session_dev")
        S = to
        searched.append(int(res[0, 2]))
    nb, nh, nfr, status = [int(v) for v in res[0, :4]]
    return dict(bp=bp[0, :, :nb].T.copy(), bscore_stack=bss[0, :nh].copy(), bp_table_idx=idx[0, :nfr + 1].copy(), step=step[0, :nfr].copy(),
                n_frame=nfr, status=status), searched


def search_streams(s, utts, schedule, lag, bp_cap=16384, bss_cap=1 << 19):
    """SEVERAL utterances in progress on one handle, as psgpu_decode_streams_step drives them (psgpu_fwdtree_search_streams + _resume +
    _restart).  utts: list of (senscr [T][n_sen], penalties [T][n_ci]) per stream; schedule: a list of steps, each a list per stream of
    (frames scored so far, final?, restart_with or None) -- restart_with = (senscr, penalties) of the utterance the stream begins anew
    BEFORE this step.  Every call is handed ONE buffer of the streams' windows (rows from the frame each search resumes at) back to
    back.  Returns the per-stream result dicts after the last step and the frames searched after every step."""
    n = len(utts)
    cur = [(np.ascontiguousarray(a, np.int16), np.ascontiguousarray(b, np.int32)) for a, b in utts]
    mf = max(max(a.shape[0] for a, _ in cur), max((r[2][0].shape[0] for st in schedule for r in st if r[2] is not None), default=0))
    bp = np.zeros((n, 10, bp_cap), np.int32); bss = np.zeros((n, bss_cap), np.int32)
    idx = np.zeros((n, mf + 2), np.int32); step = np.zeros((n, max(mf, 1), 4), np.int32); res = np.zeros((n, 8), np.int32)
    p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    S = [0] * n; log = []
    for k, st in enumerate(schedule):
        wins, pens, off, ext = [], [], [], []
        at = 0
        for u, (c, fin, again) in enumerate(st):
            if again is not None:
                cur[u] = (np.ascontiguousarray(again[0], np.int16), np.ascontiguousarray(again[1], np.int32)); S[u] = 0
                if k > 0:
                    check(lib().psgpu_fwdtree_search_restart(s.h, u, None), "psgpu_fwdtree_search_restart")
            to = c if fin else max(c - lag, S[u])
            wins.append(cur[u][0][S[u]:c]); pens.append(cur[u][1][S[u]:c])
            off.append(at - S[u]); ext += [c, to, 0]
            at += c - S[u]; S[u] = to
        win_s = np.ascontiguousarray(np.concatenate(wins)) if at else np.zeros((1, s.n_sen), np.int16)
        win_p = np.ascontiguousarray(np.concatenate(pens)) if at else np.zeros((1, s.n_ci), np.int32)
        off_a = np.array(off + [0], np.int32); ext_a = np.array(ext, np.int32)
        check(lib().psgpu_fwdtree_search_streams(s.h, p(ext_a)), "psgpu_fwdtree_search_streams")
        check(lib().psgpu_fwdtree_search_resume(s.h, 1 if k == 0 else 3), "psgpu_fwdtree_search_resume")
        check(lib().psgpu_fwdtree_search_session_dev(s.h, p(win_s), C.c_int64(s.n_sen), p(win_p), p(off_a), n, mf, bp_cap, bss_cap,
                                                     p(bp), p(bss), p(idx), p(step), p(res), 0, 0, None, None, None, None),
              "psgpu_fwdtree_search_session_dev")
        log.append([int(v) for v in res[:, 2]])
    out = []
    for u in range(n):
        nb, nh, nfr, status = [int(v) for v in res[u, :4]]
        out.append(dict(bp=bp[u, :, :nb].T.copy(), bscore_stack=bss[u, :nh].copy(), bp_table_idx=idx[u, :nfr + 1].copy(),
                        step=step[u, :nfr].copy(), n_frame=nfr, status=status))
    return out, log


class PtmView(C.Structure):
    """psgpu_ptm_view_t"""
    _fields_ = [("mean", C.c_void_p), ("var", C.c_void_p), ("det", C.c_void_p), ("mixw", C.c_void_p), ("sen2cb", C.c_void_p),
                ("logadd8", C.c_void_p), ("n_mgau", C.c_int32), ("n_feat", C.c_int32), ("n_density", C.c_int32), ("n_sen", C.c_int32),
                ("veclen", C.c_int32), ("topn", C.c_int32), ("logadd8_size", C.c_int32), ("featlen", C.c_int32 * 16),
                ("featoff", C.c_int32 * 16), ("mixw_sen", C.c_void_p)]


def search_lists(s, tables, topn_raw, topn_cw, penalties, utt_lens, pl_window=0, bp_cap=16384, bss_cap=1 << 19):
    """psgpu_fwdtree_search_lists_dev on the simulator: `s` a SimFwdtreeSearch, `tables` the scorer's tables (mixw, sen2cb,
    logadd8), topn_raw / topn_cw [T][n_mgau][n_feat][4] the scorer's lists for the utterances back to back"""
    T = int(sum(utt_lens)); n = len(utt_lens); mf = int(max(utt_lens)) if n else 0
    n_chain = topn_raw.shape[1] * topn_raw.shape[2]
    # chain-major, as psgpu_ptm_score_batch_dev leaves them
    tsc = np.ascontiguousarray(np.asarray(topn_raw, np.int32).reshape(T, n_chain, 4).transpose(1, 0, 2))
    tcw = np.ascontiguousarray(np.asarray(topn_cw, np.uint8).reshape(T, n_chain, 4).transpose(1, 0, 2))
    mw = np.ascontiguousarray(tables["mixw"], np.uint8)                     # [n_feat][n_density][n_sen]
    ds = (mw.shape[1] + 63) // 64 * 64
    by_sen = np.zeros((mw.shape[2], mw.shape[0], ds), np.uint8)             # senone-major, as the device model keeps it
    by_sen[:, :, :mw.shape[1]] = mw.transpose(2, 0, 1)
    keep = dict(mixw=mw, by_sen=by_sen, sen2cb=np.ascontiguousarray(tables["sen2cb"], np.uint8),
                la=np.ascontiguousarray(tables["logadd8"], np.uint8))
    v = PtmView()
    v.mixw = keep["mixw"].ctypes.data; v.mixw_sen = keep["by_sen"].ctypes.data; v.sen2cb = keep["sen2cb"].ctypes.data; v.logadd8 = keep["la"].ctypes.data
    v.n_mgau = int(tables["n_mgau"][0]); v.n_feat = int(tables["n_feat"][0]); v.n_density = int(tables["n_density"][0])
    v.n_sen = int(tables["n_sen"][0]); v.topn = 4; v.logadd8_size = int(keep["la"].size)
    off = np.zeros(n + 1, np.int32); off[1:] = np.cumsum(utt_lens)
    d_p = np.ascontiguousarray(penalties, np.int32)
    bp = np.zeros((n, 10, bp_cap), np.int32); bss = np.zeros((n, bss_cap), np.int32)
    idx = np.zeros((n, mf + 2), np.int32); step = np.zeros((n, max(mf, 1), 4), np.int32); res = np.zeros((n, 8), np.int32)
    p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    L = lib()
    L.psgpu_fwdtree_can_score_lists.argtypes = [C.c_void_p, C.c_void_p]
    assert L.psgpu_fwdtree_can_score_lists(s.h, C.byref(v)) == 1
    check(L.psgpu_fwdtree_search_lists_dev(s.h, C.byref(v), p(tsc), p(tcw), T, p(d_p), p(off), n, mf, bp_cap, bss_cap, p(bp), p(bss),
                                           p(idx), p(step), p(res), int(pl_window), None, None, None, None),
          "psgpu_fwdtree_search_lists_dev")
    out = []
    for u in range(n):
        nb, nh, nfr, status = [int(x) for x in res[u, :4]]
        out.append(dict(bp=bp[u, :, :nb].T.copy(), bscore_stack=bss[u, :nh].copy(), bp_table_idx=idx[u, :nfr + 1].copy(),
                        step=step[u, :nfr].copy(), n_frame=nfr, status=status, listed=int(res[u, 7])))
    return out


class SimFwdflatSearch:
    """pocketsphinx_amd.flat.FwdflatSearch on the simulator."""

    def __init__(self, static, fstatic, par, flat_par, lwf, lm=None):
        from pocketsphinx_amd.flat import marshal
        self._keep, self._ft, t = marshal(static, fstatic, par, flat_par, lwf, lm)
        self.h = C.c_void_p()
        check(lib().psgpu_fwdflat_create(C.byref(self.h), C.byref(t)), "psgpu_fwdflat_create")
        self.lm = lm
        if lm is not None:
            check(lib().psgpu_fwdflat_set_lm(self.h, lm.h), "psgpu_fwdflat_set_lm")
        self.n_sen = int(par[2]); self.n1 = int(par[6]); self.n_emit = int(par[1])

    def close(self):
        if self.h:
            lib().psgpu_fwdflat_free(self.h)
            self.h = C.c_void_p()

    def search(self, senscr, utt_lens, bp1, w1_ssid=None, bp_cap=16384, bss_cap=1 << 19, ptm_tables=None, topn_seed=None, lists=None):
        """bp1: per utterance the first pass's table [n][10], or the `handover` dict of SimFwdtreeSearch.search.
        lists: (topn_score [n_chain][T][4] int32, topn_cw [n_chain][T][4] uint8, open [n_chain][T] uint8) -- the batch scorer's
        lists of the same frames (psgpu_fwdflat_search_feats_lists_dev).
        ptm_tables (a tests/golden/*_ptm_tables.npz) + topn_seed: senscr holds the FEATURE rows and the kernel scores
        its own senones (psgpu_fwdflat_search_feats_dev)."""
        off = np.zeros(len(utt_lens) + 1, np.int32); off[1:] = np.cumsum(utt_lens)
        n = len(utt_lens); mf = int(max(utt_lens)) if n else 0
        view = None
        if ptm_tables is not None:
            from pocketsphinx_amd.flat import PtmView
            t = ptm_tables
            k = dict(mean=np.ascontiguousarray(t["mean"], np.float32), var=np.ascontiguousarray(t["var"], np.float32),
                     det=np.ascontiguousarray(t["det"], np.float32), mixw=np.ascontiguousarray(t["mixw"], np.uint8),
                     sen2cb=np.ascontiguousarray(t["sen2cb"], np.uint8), logadd8=np.ascontiguousarray(t["logadd8"], np.uint8))
            fl = [int(v) for v in t["featlen"]]
            view = PtmView(*[k[x].ctypes.data for x in ("mean", "var", "det", "mixw", "sen2cb", "logadd8")],
                           int(t["n_mgau"][0]), int(t["n_feat"][0]), int(t["n_density"][0]), int(t["n_sen"][0]), sum(fl),
                           int(t["max_topn"][0]), int(k["logadd8"].size))
            for i, v in enumerate(fl):
                view.featlen[i] = v; view.featoff[i] = sum(fl[:i])
            if lists is not None:                        # the weights senone-major as well (psgpu_ptm_view_t.mixw_sen), as the product's model has them
                nd = int(t["n_density"][0]); ds = (nd + 63) & ~63
                k["mixw_sen"] = np.zeros((int(t["n_sen"][0]), int(t["n_feat"][0]), ds), np.uint8)
                k["mixw_sen"][:, :, :nd] = np.transpose(k["mixw"].reshape(int(t["n_feat"][0]), nd, int(t["n_sen"][0])), (2, 0, 1))
                view.mixw_sen = k["mixw_sen"].ctypes.data
            d_s = np.ascontiguousarray(senscr, np.float32)
            d_seed = np.ascontiguousarray(topn_seed, np.int32)
            assert d_s.shape == (int(off[-1]), view.veclen) and d_seed.size == n * view.n_mgau * view.n_feat * view.topn
        else:
            d_s = np.ascontiguousarray(senscr, np.int16)
            assert d_s.shape == (int(off[-1]), self.n_sen)
        if isinstance(bp1, dict):
            h_bp1, h_res1, cap1, d_w1 = bp1["bp"], bp1["result"], bp1["bp_cap"], bp1["w1_ssid"]
        else:
            cap1 = max(1, max(int(b.shape[0]) for b in bp1))
            h_bp1 = np.zeros((n, 10, cap1), np.int32); h_res1 = np.zeros((n, 8), np.int32)
            for u, b in enumerate(bp1):
                h_bp1[u, :, :b.shape[0]] = np.asarray(b, np.int32).T
                h_res1[u, 0] = b.shape[0]; h_res1[u, 2] = utt_lens[u]
            d_w1 = None if w1_ssid is None else np.ascontiguousarray(np.stack([np.asarray(w, np.int32) for w in w1_ssid]), np.int32)
        bp = np.zeros((n, 10, bp_cap), np.int32); bss = np.zeros((n, bss_cap), np.int32)
        idx = np.zeros((n, mf + 2), np.int32); step = np.zeros((n, max(mf, 1), 4), np.int32); res = np.zeros((n, 8), np.int32)
        p = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None  # noqa: E731
        if view is not None and lists is not None:
            l_sc = np.ascontiguousarray(lists[0], np.int32); l_cw = np.ascontiguousarray(lists[1], np.uint8)
            l_op = np.ascontiguousarray(lists[2], np.uint8)
            nch = view.n_mgau * view.n_feat
            assert l_sc.shape == (nch, int(off[-1]), 4) and l_cw.shape == l_sc.shape and l_op.shape == l_sc.shape[:2]
            check(lib().psgpu_fwdflat_search_feats_lists_dev(self.h, C.byref(view), p(d_s), p(d_seed), p(l_sc), p(l_cw), p(l_op), int(off[-1]),
                                                             p(off), n, mf, cap1, p(h_bp1), p(h_res1), p(d_w1), bp_cap, bss_cap, p(bp),
                                                             p(bss), p(idx), p(step), p(res), None),
                  "psgpu_fwdflat_search_feats_lists_dev")
        elif view is not None:
            check(lib().psgpu_fwdflat_search_feats_dev(self.h, C.byref(view), p(d_s), p(d_seed), p(off), n, mf, cap1, p(h_bp1), p(h_res1),
                                                       p(d_w1), bp_cap, bss_cap, p(bp), p(bss), p(idx), p(step), p(res), None),
                  "psgpu_fwdflat_search_feats_dev")
        else:
            check(lib().psgpu_fwdflat_search_dev(self.h, p(d_s), C.c_int64(self.n_sen), p(off), n, mf, cap1, p(h_bp1), p(h_res1), p(d_w1),
                                                 bp_cap, bss_cap, p(bp), p(bss), p(idx), p(step), p(res), None), "psgpu_fwdflat_search_dev")
        out = []
        for u in range(n):
            nb, nh, nfr, status = [int(v) for v in res[u, :4]]
            out.append(dict(bp=bp[u, :, :nb].T.copy(), bscore_stack=bss[u, :nh].copy(), bp_table_idx=idx[u, :nfr + 1].copy(),
                            step=step[u, :nfr].copy(), n_frame=nfr, status=status))
        return out
