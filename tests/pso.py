"""ctypes binding of oracle/libpsoracle.so (the CPU checker).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

_lib = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libpsoracle.so"])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "libpsoracle.so")
        src = os.path.join(ORACLE_DIR, "ps_oracle.c")
        if (not os.path.exists(path)) or os.path.getmtime(path) < os.path.getmtime(src):
            build_oracle()
        L = C.CDLL(path)
        vp, i32 = C.c_void_p, C.c_int32
        L.pso_ptm_new.restype = vp
        L.pso_ptm_new.argtypes = [i32, i32, i32, vp, i32, i32, i32, i32,
                                  vp, vp, vp, vp, vp, vp, vp, i32]
        L.pso_ptm_free.argtypes = [vp]
        L.pso_ptm_reset_hist.argtypes = [vp]
        L.pso_ptm_set_frame_idx.argtypes = [vp, i32]
        L.pso_ptm_get_frame_idx.argtypes = [vp]
        L.pso_ptm_get_frame_idx.restype = i32
        L.pso_ptm_frame_eval.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp]
        L.pso_ptm_frame_eval.restype = i32
        L.pso_ptm_cur_topn.argtypes = [vp]
        L.pso_ptm_cur_topn.restype = vp
        L.pso_ptm_score_utt.argtypes = [vp, vp, i32, i32, vp, vp, vp]
        L.pso_flags2list.argtypes = [vp, i32, vp]
        L.pso_flags2list.restype = i32
        L.pso_hmm_vit_eval.argtypes = [vp, vp]
        L.pso_hmm_vit_eval.restype = i32
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def load_tables(name="en_us_ptm_tables.npz"):
    z = np.load(os.path.join(GOLDEN_DIR, name))
    return {k: z[k] for k in z.files}


class OraclePTM:
    """Stateful wrapper around pso_ptm_t (restates ptm_mgau.c)."""

    def __init__(self, t, n_fast_hist=None, topn=None, ds_ratio=None):
        L = lib()
        self.t = t
        self.n_mgau = int(t["n_mgau"][0]); self.n_feat = int(t["n_feat"][0])
        self.n_density = int(t["n_density"][0]); self.n_sen = int(t["n_sen"][0])
        self.topn = int(topn if topn is not None else t["max_topn"][0])
        self.ds = int(ds_ratio if ds_ratio is not None else t["ds_ratio"][0])
        self.n_hist = int(n_fast_hist if n_fast_hist is not None else t["n_fast_hist"][0])
        self.veclen = int(t["featlen"].sum())
        # keep contiguous copies alive for the lifetime of the C object
        self._keep = dict(
            featlen=np.ascontiguousarray(t["featlen"], np.int32),
            mean=np.ascontiguousarray(t["mean"], np.float32),
            var=np.ascontiguousarray(t["var"], np.float32),
            det=np.ascontiguousarray(t["det"], np.float32),
            mixw=np.ascontiguousarray(t["mixw"], np.uint8),
            sen2cb=np.ascontiguousarray(t["sen2cb"], np.uint8),
            logadd8=np.ascontiguousarray(t["logadd8"], np.uint8),
            mixw_cb=(np.ascontiguousarray(t["mixw_cb"], np.uint8) if "mixw_cb" in t else None),
        )
        k = self._keep
        self.h = L.pso_ptm_new(self.n_mgau, self.n_feat, self.n_density, _p(k["featlen"]),
                               self.n_sen, self.topn, self.ds, self.n_hist,
                               _p(k["mean"]), _p(k["var"]), _p(k["det"]), _p(k["mixw"]),
                               _p(k["mixw_cb"]), _p(k["sen2cb"]), _p(k["logadd8"]),
                               int(k["logadd8"].size))

    def __del__(self):
        try:
            lib().pso_ptm_free(self.h)
        except Exception:
            pass

    def reset_hist(self):
        lib().pso_ptm_reset_hist(self.h)

    def set_frame_idx(self, v):
        lib().pso_ptm_set_frame_idx(self.h, int(v))

    def frame_eval(self, feat, frame, active=None, compallsen=True, want_raw=False):
        feat = np.ascontiguousarray(feat, np.float32)
        scr = np.empty(self.n_sen, np.int16)
        raw = np.empty((self.n_mgau, self.n_feat, self.topn, 2), np.int32) if want_raw else None
        act = np.ascontiguousarray(active, np.uint8) if active is not None else None
        ev = lib().pso_ptm_frame_eval(self.h, _p(scr), _p(act),
                                      0 if act is None else act.size, _p(feat),
                                      int(frame), int(bool(compallsen)), _p(raw))
        if want_raw:
            return scr, raw, ev
        return scr

    def cur_topn(self):
        p = lib().pso_ptm_cur_topn(self.h)
        n = self.n_mgau * self.n_feat * self.topn * 2
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(n,)).copy()
        return a.reshape(self.n_mgau, self.n_feat, self.topn, 2)

    def score_utt(self, feats, reset_hist=True, want_topn=True):
        feats = np.ascontiguousarray(feats, np.float32)
        T = feats.shape[0]
        scr = np.empty((T, self.n_sen), np.int16)
        cw = np.empty((T, self.n_mgau, self.n_feat, self.topn), np.uint8) if want_topn else None
        raw = np.empty((T, self.n_mgau, self.n_feat, self.topn), np.int32) if want_topn else None
        lib().pso_ptm_score_utt(self.h, _p(feats), T, int(bool(reset_hist)), _p(scr), _p(cw), _p(raw))
        return scr, cw, raw


def flags2list(flags):
    flags = np.ascontiguousarray(flags, np.uint8)
    out = np.empty(flags.size + flags.size // 255 + 8, np.uint8)
    n = lib().pso_flags2list(_p(flags), flags.size, _p(out))
    return out[:n].copy()


# ---- HMM step ---------------------------------------------------------
class PsoHmm(C.Structure):
    _fields_ = [("score", C.c_int32 * 5), ("history", C.c_int32 * 5),
                ("out_score", C.c_int32), ("out_history", C.c_int32),
                ("ssid", C.c_uint16), ("senid", C.c_uint16 * 5),
                ("bestscore", C.c_int32), ("tmatid", C.c_int16),
                ("frame", C.c_int32), ("mpx", C.c_uint8), ("n_emit_state", C.c_uint8)]


class PsoHmmCtx(C.Structure):
    _fields_ = [("n_emit_state", C.c_int), ("tp", C.c_void_p),
                ("senscore", C.c_void_p), ("sseq", C.c_void_p)]


def row_hash(a):
    """64-bit FNV-1a over each row of a 2-D (or flattened-per-first-axis) array."""
    a = np.ascontiguousarray(a)
    b = a.reshape(a.shape[0], -1).view(np.uint8)
    h = np.full(b.shape[0], 0xcbf29ce484222325, np.uint64)
    prime = np.uint64(0x100000001b3)
    with np.errstate(over="ignore"):
        for j in range(b.shape[1]):
            h ^= b[:, j].astype(np.uint64)
            h *= prime
    return h


def clustered_tables(tables, g):
    """the en-us tables with the 4-bit clustered mixture weights of a ptm_4bit_* golden"""
    t = dict(tables)
    t["mixw"] = np.ascontiguousarray(g["mixw4"], np.uint8)
    t["mixw_cb"] = np.ascontiguousarray(g["mixw_cb"], np.uint8)
    t["mixw_is_4bit"] = np.array([1], np.int32)
    return t


def write_clustered_model_dir(dst, src_model, g):
    """a copy (symlinks) of an acoustic model directory whose sendump holds the 4-bit clustered weights of a
    ptm_4bit_* golden, in the layout read_sendump (ptm_mgau.c:457-654) reads"""
    import struct
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src_model):
        if f != "sendump":
            os.symlink(os.path.join(src_model, f), os.path.join(dst, f))
    packed, cb = np.ascontiguousarray(g["mixw4"], np.uint8), np.ascontiguousarray(g["mixw_cb"], np.uint8)
    n_feat, n_den, _ = packed.shape

    def lstr(txt):
        b = txt.encode() + b"\0"
        return struct.pack("<i", len(b)) + b
    with open(os.path.join(dst, "sendump"), "wb") as fh:
        fh.write(lstr("V6 Senone Probs, Smoothed, Normalized"))
        fh.write(lstr("(HMM file format)"))
        for h in ("feature_count %d" % n_feat, "mixture_count %d" % n_den, "model_count %d" % int(g["senscr_sample"].shape[1]),
                  "cluster_count %d" % cb.size, "cluster_bits 4", "logbase 1.0001", "mixw_shift 10"):
            fh.write(lstr(h))
        fh.write(struct.pack("<i", 0))
        fh.write(cb.tobytes())
        fh.write(packed.tobytes())
    return dst


HMM_FIELDS = 19   # score[5] history[5] out_score out_history senid[5] bestscore tmatid (ref_dump.c hmm_pack)


def hmm_step_oracle(g, t):
    """Run pso_hmm_vit_eval over every HMM of step t of an hmm_*.npz fixture
    (state `before` -> returns (after [n_hmm][19], ret [n_hmm]))."""
    L = lib()
    n_emit = int(g["n_emit"][0])
    tp = np.ascontiguousarray(g["tp"], np.uint8)
    sseq = np.ascontiguousarray(g["sseq"], np.uint16)
    scr = np.ascontiguousarray(g["senscr"][t], np.int16)
    ctx = PsoHmmCtx(n_emit, tp.ctypes.data, scr.ctypes.data, sseq.ctypes.data)
    before = g["before"][t]
    n = before.shape[0]
    after = np.empty_like(before)
    ret = np.empty(n, np.int32)
    h = PsoHmm()
    for i in range(n):
        b = before[i]
        for s in range(5):
            h.score[s] = int(b[s]); h.history[s] = int(b[5 + s]); h.senid[s] = int(b[12 + s])
        h.out_score = int(b[10]); h.out_history = int(b[11])
        h.bestscore = int(b[17]); h.tmatid = int(b[18])
        h.mpx = int(g["mpx"][i]); h.n_emit_state = n_emit; h.ssid = int(b[12])
        ret[i] = L.pso_hmm_vit_eval(C.byref(ctx), C.byref(h))
        a = after[i]
        for s in range(5):
            a[s] = h.score[s]; a[5 + s] = h.history[s]; a[12 + s] = h.senid[s]
        a[10] = h.out_score; a[11] = h.out_history; a[17] = h.bestscore; a[18] = h.tmatid
    return after, ret


class OracleSemi:
    """Stateful wrapper around pso_semi_t (restates s2_semi_mgau.c)."""

    def __init__(self, t, topn=None, ds_ratio=None, topn_beam=None):
        L = lib()
        vp, i32 = C.c_void_p, C.c_int32
        L.pso_semi_new.restype = vp
        L.pso_semi_new.argtypes = [i32, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32]
        L.pso_semi_free.argtypes = [vp]
        L.pso_semi_reset_hist.argtypes = [vp]
        L.pso_semi_set_frame_idx.argtypes = [vp, i32]
        L.pso_semi_frame_eval.argtypes = [vp, vp, vp, i32, vp, i32, i32]
        L.pso_semi_cur_topn.argtypes = [vp, vp]
        L.pso_semi_cur_topn.restype = vp
        self.n_feat = int(t["n_feat"][0]); self.n_density = int(t["n_density"][0])
        self.n_sen = int(t["n_sen"][0])
        self.topn = int(topn if topn is not None else t["max_topn"][0])
        self.ds = int(ds_ratio if ds_ratio is not None else t["ds_ratio"][0])
        self.n_hist = int(t["n_fast_hist"][0])
        self.veclen = int(t["featlen"].sum())
        beam = np.asarray(topn_beam if topn_beam is not None else t["topn_beam"], np.uint8)
        self._keep = dict(
            featlen=np.ascontiguousarray(t["featlen"], np.int32), beam=np.ascontiguousarray(beam),
            mean=np.ascontiguousarray(t["mean"], np.float32), var=np.ascontiguousarray(t["var"], np.float32),
            det=np.ascontiguousarray(t["det"], np.float32), mixw=np.ascontiguousarray(t["mixw"], np.uint8),
            mixw_cb=(np.ascontiguousarray(t["mixw_cb"], np.uint8) if "mixw_cb" in t else None),
            logadd8=np.ascontiguousarray(t["logadd8"], np.uint8))
        k = self._keep
        self.h = L.pso_semi_new(self.n_feat, self.n_density, _p(k["featlen"]), self.n_sen, self.topn,
                                self.ds, self.n_hist, _p(k["beam"]), _p(k["mean"]), _p(k["var"]),
                                _p(k["det"]), _p(k["mixw"]), _p(k["mixw_cb"]), _p(k["logadd8"]),
                                int(k["logadd8"].size))

    def __del__(self):
        try:
            lib().pso_semi_free(self.h)
        except Exception:
            pass

    def set_frame_idx(self, v):
        lib().pso_semi_set_frame_idx(self.h, int(v))

    def frame_eval(self, feat, frame, active=None, compallsen=True):
        feat = np.ascontiguousarray(feat, np.float32)
        scr = np.empty(self.n_sen, np.int16)
        act = np.ascontiguousarray(active, np.uint8) if active is not None else None
        lib().pso_semi_frame_eval(self.h, _p(scr), _p(act), 0 if act is None else act.size,
                                  _p(feat), int(frame), int(bool(compallsen)))
        return scr

    def cur_topn(self):
        n = np.empty(self.n_feat, np.uint8)
        p = lib().pso_semi_cur_topn(self.h, _p(n))
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(self.n_feat * self.topn * 2,)).copy()
        return a.reshape(self.n_feat, self.topn, 2), n


def senlog_params(g):
    """topn / ds / topn_beam / ... overrides a senlog fixture was recorded with."""
    e = [str(x) for x in g["extra"]]
    return dict(zip(e[0::2], e[1::2]))


class OracleMs:
    """Wrapper around pso_ms_t (restates ms_mgau.c / ms_gauden.c / ms_senone.c).
    `senscr` is a persistent in/out buffer like acmod->senone_scores."""

    def __init__(self, t, topn=None, aw=None):
        L = lib()
        vp, i32 = C.c_void_p, C.c_int32
        L.pso_ms_new.restype = vp
        L.pso_ms_new.argtypes = [i32, i32, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32]
        L.pso_ms_free.argtypes = [vp]
        L.pso_ms_frame_eval.argtypes = [vp, vp, vp, i32, vp, i32]
        self.n_mgau = int(t["n_mgau"][0]); self.n_feat = int(t["n_feat"][0])
        self.n_density = int(t["n_density"][0]); self.n_sen = int(t["n_sen"][0])
        self.topn = min(int(topn if topn is not None else t["max_topn"][0]), self.n_density)
        self.aw = int(aw if aw is not None else t["aw"][0])
        self._keep = dict(
            featlen=np.ascontiguousarray(t["featlen"], np.int32),
            mean=np.ascontiguousarray(t["mean"], np.float32), var=np.ascontiguousarray(t["var"], np.float32),
            det=np.ascontiguousarray(t["det"], np.float32), pdf=np.ascontiguousarray(t["pdf"], np.uint8),
            map=np.ascontiguousarray(t["sen2mgau"], np.uint32), logadd=np.ascontiguousarray(t["logadd"], np.uint8))
        k = self._keep
        self.h = L.pso_ms_new(self.n_mgau, self.n_feat, self.n_density, _p(k["featlen"]), self.n_sen,
                              self.topn, self.aw, _p(k["mean"]), _p(k["var"]), _p(k["det"]), _p(k["pdf"]),
                              _p(k["map"]), _p(k["logadd"]), int(t["logadd_size"][0]),
                              int(t["logadd_width"][0]), int(t["log_zero"][0]))
        self.senscr = np.zeros(self.n_sen, np.int16)

    def __del__(self):
        try:
            lib().pso_ms_free(self.h)
        except Exception:
            pass

    def frame_eval(self, feat, active=None, compallsen=True):
        feat = np.ascontiguousarray(feat, np.float32)
        act = np.ascontiguousarray(active, np.uint8) if active is not None else None
        lib().pso_ms_frame_eval(self.h, _p(self.senscr), _p(act), 0 if act is None else act.size,
                                _p(feat), int(bool(compallsen)))
        return self.senscr.copy()


class OracleFe:
    """Wrapper around pso_fe_t / pso_fe_process_utt (restates fe_sigproc.c,
    fe_noise.c, fe_interface.c).  `t` = the arrays of an mfcc_*.npz fixture
    (the reference's own precomputed front-end tables).  The noise tracker is
    kept across process() calls like the reference's noise_stats_t; reset()
    = fe_reset_noisestats."""

    class _S(C.Structure):
        _fields_ = [(n, C.c_int32) for n in ("frame_size", "frame_shift", "fft_size", "fft_order", "n_filt",
                                             "num_cepstra", "out_dim", "transform", "log_spec", "remove_dc",
                                             "remove_noise", "has_lifter")] + \
                   [(n, C.c_float) for n in ("alpha", "sqrt_inv_n", "sqrt_inv_2n")] + \
                   [(n, C.c_void_p) for n in ("hamming", "ccc", "sss", "spec_start", "filt_start", "filt_width",
                                              "filt_coeffs", "mel_cosine", "lifter")]

    def __init__(self, t):
        par = [int(v) for v in t["par"]]
        self._keep = dict(
            hamming=np.ascontiguousarray(t["hamming"], np.float64), ccc=np.ascontiguousarray(t["ccc"], np.float64),
            sss=np.ascontiguousarray(t["sss"], np.float64), spec_start=np.ascontiguousarray(t["spec_start"], np.int16),
            filt_start=np.ascontiguousarray(t["filt_start"], np.int16),
            filt_width=np.ascontiguousarray(t["filt_width"], np.int16),
            filt_coeffs=np.ascontiguousarray(t["filt_coeffs"], np.float32),
            mel_cosine=np.ascontiguousarray(t["mel_cosine"], np.float32))
        if "lifter" in t:
            self._keep["lifter"] = np.ascontiguousarray(t["lifter"], np.float32)
        k = self._keep
        s = self._S()
        (s.frame_size, s.frame_shift, s.fft_size, s.fft_order, s.n_filt, s.num_cepstra, s.out_dim, s.transform,
         s.log_spec, s.remove_dc, s.remove_noise) = par[:11]
        s.has_lifter = int("lifter" in k)
        s.alpha = float(t["alpha"][0]); s.sqrt_inv_n = float(t["sqrt_inv_n"][0]); s.sqrt_inv_2n = float(t["sqrt_inv_2n"][0])
        for n in ("hamming", "ccc", "sss", "spec_start", "filt_start", "filt_width", "filt_coeffs", "mel_cosine", "lifter"):
            setattr(s, n, _p(k[n]) if n in k else None)
        self.s = s
        self.out_dim = s.out_dim
        self.noise = np.zeros((4, s.n_filt), np.float64)
        self.undefined = C.c_int32(1)
        L = lib()
        L.pso_fe_n_frames.argtypes = [C.c_void_p, C.c_long]
        L.pso_fe_process_utt.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]

    def reset(self):
        self.undefined = C.c_int32(1)

    def n_frames(self, n):
        return int(lib().pso_fe_n_frames(C.byref(self.s), n))

    def process(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        nfr = self.n_frames(pcm.size)
        cep = np.empty((nfr, self.out_dim), np.float32)
        got = lib().pso_fe_process_utt(C.byref(self.s), _p(pcm), pcm.size, _p(cep), _p(self.noise),
                                       C.byref(self.undefined))
        assert got == nfr
        return cep


class OracleFwdtree:
    """Wrapper around pso_ft_t (oracle/ps_oracle_search.c: restates ngram_search_fwdtree.c and the
    back-pointer helpers of ngram_search.c on flat tables).  `static` = a fwdtree_static_*.npz,
    `par` = the trace's parameter vector (beams, penalties, word ids)."""

    NAMES = ["par", "node_ci", "node_ci2", "node_ssid", "node_tmat", "node_child", "node_sib", "node_penult_wid",
             "homophone_set", "w1_wid", "w1_ci", "w1_ci2", "w1_ssid", "w1_tmat", "w1_mpx", "dict_pronlen", "dict_first",
             "dict_last", "dict_last2", "dict_basewid", "dict_filler", "rssid_n", "rssid_ssid", "rssid_cimap", "ldiph_lc",
             "tp", "sseq", "ci_tmat", "lm"]
    DT = {"tp": np.uint8, "sseq": np.uint16}

    def __init__(self, static, par, lm=None):
        """lm: an OracleLm -- language scores from the trie oracle instead of the dense table static["lm"]"""
        L = lib()
        src = dict(static); src["par"] = par
        self._keep = {n: np.ascontiguousarray(src[n], self.DT.get(n, np.int32)) for n in self.NAMES
                      if not (n == "lm" and lm is not None)}

        class T(C.Structure):
            _fields_ = [(n, C.c_void_p) for n in self.NAMES]
        self._t = T(*[self._keep[n].ctypes.data if n in self._keep else None for n in self.NAMES])
        self._lm = lm
        vp = C.c_void_p
        L.pso_ft_new.restype = vp; L.pso_ft_new.argtypes = [vp]
        L.pso_ft_free.argtypes = [vp]; L.pso_ft_start.argtypes = [vp]
        L.pso_ft_active_list.argtypes = [vp, C.c_int, vp]
        L.pso_ft_step.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int16, vp]
        L.pso_ft_finish.argtypes = [vp, C.c_int]
        for f in ("pso_ft_best_score", "pso_ft_last_phone_best_score", "pso_ft_bpidx", "pso_ft_bss_head"):
            getattr(L, f).argtypes = [vp]; getattr(L, f).restype = C.c_int32
        for f in ("pso_ft_bp", "pso_ft_bss", "pso_ft_bp_table_idx"):
            getattr(L, f).argtypes = [vp]; getattr(L, f).restype = vp
        self.n_sen = int(par[2])
        self.h = L.pso_ft_new(C.byref(self._t))
        if lm is not None:
            L.pso_ft_set_lm.argtypes = [vp, vp]
            L.pso_ft_set_lm(self.h, lm.h)
        self._buf = np.zeros(self.n_sen, np.int32)

    def __del__(self):
        try:
            lib().pso_ft_free(self.h)
        except Exception:
            pass

    def start(self):
        lib().pso_ft_start(self.h)

    def set_mpx_ssids(self, ssid):
        a = np.ascontiguousarray(ssid, np.int32)
        lib().pso_ft_set_mpx_ssids.argtypes = [C.c_void_p, C.c_void_p]
        lib().pso_ft_set_mpx_ssids(self.h, _p(a))

    def get_mpx_ssids(self, shape):
        a = np.zeros(shape, np.int32)
        lib().pso_ft_get_mpx_ssids.argtypes = [C.c_void_p, C.c_void_p]
        lib().pso_ft_get_mpx_ssids(self.h, _p(a))
        return a

    def active_list(self, frame):
        n = lib().pso_ft_active_list(self.h, int(frame), _p(self._buf))
        return self._buf[:n].copy()

    def step(self, frame, ids, scr, rest, penalties):
        ids = np.ascontiguousarray(ids, np.int32); scr = np.ascontiguousarray(scr, np.int16)
        pen = np.ascontiguousarray(penalties, np.int32)
        return lib().pso_ft_step(self.h, int(frame), _p(ids), _p(scr), int(ids.size), int(rest), _p(pen))

    def finish(self, n_frames):
        lib().pso_ft_finish(self.h, int(n_frames))

    def best_score(self):
        return int(lib().pso_ft_best_score(self.h))

    def last_phone_best_score(self):
        return int(lib().pso_ft_last_phone_best_score(self.h))

    def bpidx(self):
        return int(lib().pso_ft_bpidx(self.h))

    def bp_table(self):
        L = lib()
        n = L.pso_ft_bpidx(self.h)
        return np.ctypeslib.as_array(C.cast(L.pso_ft_bp(self.h), C.POINTER(C.c_int32)), shape=(n, 10)).copy()

    def bscore_stack(self):
        L = lib()
        n = L.pso_ft_bss_head(self.h)
        return np.ctypeslib.as_array(C.cast(L.pso_ft_bss(self.h), C.POINTER(C.c_int32)), shape=(max(n, 1),)).copy()[:n]

    def bp_table_idx(self, n_frames):
        L = lib()
        return np.ctypeslib.as_array(C.cast(L.pso_ft_bp_table_idx(self.h), C.POINTER(C.c_int32)),
                                     shape=(n_frames + 1,)).copy()


class OracleLm:
    """Wrapper around pso_lm_t (oracle/ps_oracle_lm.c: restates ngram_tg_score through the model set,
    the trie and its quantisation tables).  `g` = an lm_*.npz fixture (`ref_dump lm`);
    lw / log_wip override the weights the fixture was dumped with."""

    def __init__(self, g, lw=None, log_wip=None):
        L = lib()
        vp = C.c_void_p
        L.pso_lm_new.restype = vp
        L.pso_lm_new.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_uint64, vp, vp, C.c_float, C.c_int32,
                                 C.c_int32, vp]
        L.pso_lm_free.argtypes = [vp]
        L.pso_lm_tg_score.restype = C.c_int32
        L.pso_lm_tg_score.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp]
        L.pso_lm_tg_score_batch.argtypes = [vp, vp, vp, vp, C.c_int64, vp, vp]
        self.order = int(g["order"][0]); self.n_words = int(g["n_words"][0])
        mem = np.concatenate([np.asarray(g["ngram_mem"], np.uint8), np.zeros(8, np.uint8)])
        quant = g["quant"] if "quant" in g else np.zeros((1, 65536), np.float32)
        self._keep = [np.ascontiguousarray(g["unigrams"]).view(np.uint32), mem,
                      np.ascontiguousarray(g["levels"]).view(np.uint32), np.ascontiguousarray(quant, np.float32),
                      np.ascontiguousarray(g["widmap"], np.int32)]
        self.lw = float(g["lw"][0]) if lw is None else lw
        self.log_wip = int(g["log_wip"][0]) if log_wip is None else log_wip
        self.h = L.pso_lm_new(self.order, int(g["n_unigrams"][0]), self.n_words, self._keep[0].ctypes.data,
                              mem.ctypes.data, len(g["ngram_mem"]), self._keep[2].ctypes.data, self._keep[3].ctypes.data,
                              self.lw, self.log_wip, int(g["log_zero"][0]), self._keep[4].ctypes.data)
        assert self.h

    def __del__(self):
        try:
            lib().pso_lm_free(self.h)
        except Exception:
            pass

    def tg_score(self, w3, w2=-1, w1=-1):
        nu = C.c_int32(0)
        s = lib().pso_lm_tg_score(self.h, w3, w2, w1, C.byref(nu))
        return s, nu.value

    def tg_score_batch(self, q):
        q = np.ascontiguousarray(q, np.int32)
        w3, w2, w1 = (np.ascontiguousarray(q[:, i]) for i in range(3))
        sc = np.zeros(len(q), np.int32); nu = np.zeros(len(q), np.int32)
        lib().pso_lm_tg_score_batch(self.h, w3.ctypes.data, w2.ctypes.data, w1.ctypes.data, len(q), sc.ctypes.data,
                                    nu.ctypes.data)
        return sc, nu


class OracleFwdflat:
    """Wrapper around pso_ff_t (oracle/ps_oracle_flat.c: restates ngram_search_fwdflat.c on flat tables).
    `static` = a fwdtree_static_*.npz, `fstatic` = the matching fwdflat_static_*.npz, `g` = a fwdflat trace
    (par, flat_par, flat_lwf)."""

    EXTRA = ["pron_off", "pron_ci", "pron_ssid", "ci_ssid", "lm_known", "flat_par"]

    def __init__(self, static, fstatic, g, lm=None):
        L = lib()
        names = OracleFwdtree.NAMES
        src = dict(static); src.update(dict(fstatic)); src["par"] = g["par"]; src["flat_par"] = g["flat_par"]
        self._keep = {n: np.ascontiguousarray(src[n], OracleFwdtree.DT.get(n, np.int32)) for n in names + self.EXTRA
                      if not (n == "lm" and lm is not None)}

        class T(C.Structure):
            _fields_ = [(n, C.c_void_p) for n in names + self.EXTRA] + [("lwf", C.c_float)]
        self._t = T(*([self._keep[n].ctypes.data if n in self._keep else None for n in names + self.EXTRA] +
                      [float(np.asarray(g["flat_lwf"], np.float32).ravel()[0])]))
        self._lm = lm
        vp = C.c_void_p
        L.pso_ff_new.restype = vp; L.pso_ff_new.argtypes = [vp]
        L.pso_ff_free.argtypes = [vp]
        L.pso_ff_set_lm.argtypes = [vp, vp]
        L.pso_ff_start.argtypes = [vp, vp, C.c_int, C.c_int, vp]
        L.pso_ff_active_list.argtypes = [vp, C.c_int, vp]
        L.pso_ff_step.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int16]
        L.pso_ff_finish.argtypes = [vp, C.c_int]
        for f in ("pso_ff_best_score", "pso_ff_bpidx", "pso_ff_bss_head", "pso_ff_n_words", "pso_ff_n_chan"):
            getattr(L, f).argtypes = [vp]; getattr(L, f).restype = C.c_int32
        for f in ("pso_ff_bp", "pso_ff_bss", "pso_ff_bp_table_idx", "pso_ff_wordlist"):
            getattr(L, f).argtypes = [vp]; getattr(L, f).restype = vp
        self.n_sen = int(g["par"][2])
        self.h = L.pso_ff_new(C.byref(self._t))
        if lm is not None:
            L.pso_ff_set_lm(self.h, lm.h)
        self._buf = np.zeros(self.n_sen, np.int32)

    def __del__(self):
        try:
            lib().pso_ff_free(self.h)
        except Exception:
            pass

    def start(self, bp1, n_frame, w1_ssid):
        bp1 = np.ascontiguousarray(bp1, np.int32); w1 = np.ascontiguousarray(w1_ssid, np.int32)
        lib().pso_ff_start(self.h, _p(bp1), int(bp1.shape[0]), int(n_frame), _p(w1))

    def active_list(self, frame):
        n = lib().pso_ff_active_list(self.h, int(frame), _p(self._buf))
        return self._buf[:n].copy()

    def step(self, frame, ids, scr, rest):
        ids = np.ascontiguousarray(ids, np.int32); scr = np.ascontiguousarray(scr, np.int16)
        return lib().pso_ff_step(self.h, int(frame), _p(ids), _p(scr), int(ids.size), int(rest))

    def finish(self, n_frames):
        lib().pso_ff_finish(self.h, int(n_frames))

    def best_score(self):
        return int(lib().pso_ff_best_score(self.h))

    def bpidx(self):
        return int(lib().pso_ff_bpidx(self.h))

    def wordlist(self):
        L = lib()
        n = L.pso_ff_n_words(self.h)
        return np.ctypeslib.as_array(C.cast(L.pso_ff_wordlist(self.h), C.POINTER(C.c_int32)), shape=(max(n, 1),)).copy()[:n]

    def n_chan(self):
        return int(lib().pso_ff_n_chan(self.h))

    def bp_table(self):
        L = lib()
        n = L.pso_ff_bpidx(self.h)
        return np.ctypeslib.as_array(C.cast(L.pso_ff_bp(self.h), C.POINTER(C.c_int32)), shape=(max(n, 1), 10)).copy()[:n]

    def bscore_stack(self):
        L = lib()
        n = L.pso_ff_bss_head(self.h)
        return np.ctypeslib.as_array(C.cast(L.pso_ff_bss(self.h), C.POINTER(C.c_int32)), shape=(max(n, 1),)).copy()[:n]

    def bp_table_idx(self, n_frames):
        L = lib()
        return np.ctypeslib.as_array(C.cast(L.pso_ff_bp_table_idx(self.h), C.POINTER(C.c_int32)),
                                     shape=(n_frames + 1,)).copy()
