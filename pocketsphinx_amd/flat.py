"""Host-side mirror of the flat-lexicon second pass (ngram_search_t's fwdflat half, reference
src/ngram_search_fwdflat.c); arithmetic in csrc/psgpu_flat.hip."""
import ctypes as C

import numpy as np

from . import capi
from .search import _DT, _NAMES, _Tables


class _FlatTables(C.Structure):
    _fields_ = [("ft", C.c_void_p), ("pron_off", C.c_void_p), ("pron_ci", C.c_void_p), ("pron_ssid", C.c_void_p),
                ("ci_ssid", C.c_void_p), ("lm_known", C.c_void_p), ("fwdflatbeam", C.c_int32), ("fwdflatwbeam", C.c_int32),
                ("min_ef_width", C.c_int32), ("max_sf_win", C.c_int32), ("lwf", C.c_float)]


def marshal(static, fstatic, par, flat_par, lwf, lm):
    """(keep-alive dict, _Tables, _FlatTables) for psgpu_fwdflat_create"""
    src = dict(static); src["par"] = par
    keep = {n: np.ascontiguousarray(src[n], _DT.get(n, np.int32)) for n in _NAMES if n in src and not (n == "lm" and lm is not None)}
    ft = _Tables(*[keep[n].ctypes.data if n in keep else None for n in _NAMES], int(keep["tp"].shape[0]), int(keep["sseq"].shape[0]))
    for n in ("pron_off", "pron_ci", "pron_ssid", "ci_ssid", "lm_known"):
        keep["f_" + n] = np.ascontiguousarray(fstatic[n], np.int32)
    fp = [int(v) for v in np.asarray(flat_par).ravel()[:4]]
    t = _FlatTables(C.addressof(ft), keep["f_pron_off"].ctypes.data, keep["f_pron_ci"].ctypes.data, keep["f_pron_ssid"].ctypes.data,
                    keep["f_ci_ssid"].ctypes.data, keep["f_lm_known"].ctypes.data, fp[0], fp[1], fp[2], fp[3],
                    float(np.asarray(lwf, np.float32).ravel()[0]))
    return keep, ft, t


class PtmView(C.Structure):
    """psgpu_ptm_view_t (include/psgpu.h)"""
    _fields_ = [("mean", C.c_void_p), ("var", C.c_void_p), ("det", C.c_void_p), ("mixw", C.c_void_p), ("sen2cb", C.c_void_p),
                ("logadd8", C.c_void_p), ("n_mgau", C.c_int32), ("n_feat", C.c_int32), ("n_density", C.c_int32),
                ("n_sen", C.c_int32), ("veclen", C.c_int32), ("topn", C.c_int32), ("logadd8_size", C.c_int32),
                ("featlen", C.c_int32 * 16), ("featoff", C.c_int32 * 16), ("mixw_sen", C.c_void_p)]


class FwdflatSearch:
    """`static` = the first pass's flattened tables (FwdtreeSearch), `fstatic` = what the second pass adds
    (pron_off, pron_ci, pron_ssid, ci_ssid, lm_known), `par` the first pass's parameter vector, `flat_par` =
    (fwdflatbeam, fwdflatwbeam, min_ef_width, max_sf_win), `lwf` = fwdflat_fwdtree_lw_ratio."""

    def __init__(self, static, fstatic, par, flat_par, lwf, lm=None):
        self._keep, self._ft, t = marshal(static, fstatic, par, flat_par, lwf, lm)
        self.h = C.c_void_p()
        capi.check(capi.lib().psgpu_fwdflat_create(C.byref(self.h), C.byref(t)), "psgpu_fwdflat_create")
        self.lm = lm
        if lm is not None:
            capi.check(capi.lib().psgpu_fwdflat_set_lm(self.h, lm.h), "psgpu_fwdflat_set_lm")
        self.n_sen = int(par[2]); self.n1 = int(par[6]); self.n_emit = int(par[1])

    def close(self):
        if self.h:
            capi.lib().psgpu_fwdflat_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search(self, senscr, utt_lens, bp1, w1_ssid=None, bp_cap=16384, bss_cap=1 << 19, ptm=None, topn_seed=None, lists=None):
        """senscr [T][n_sen] int16 for utterances back to back (or, with ptm = a PtmModel and topn_seed
        [n_utt][n_chain][topn] codewords, the FEATURE rows [T][veclen] float32: the kernel then scores its own senones,
        psgpu_fwdflat_search_feats_dev); bp1: per utterance the first pass's back-pointer
        table [n][10] (numpy), or the `handover` dict FwdtreeSearch.search filled (device buffers as
        psgpu_fwdtree_search_dev left them); w1_ssid: per utterance [n_1ph][n_emit] or None.
        lists = (topn_score, topn_cw): the batch scorer's chain-major device lists of the same frames (the scorer's last
        call on the current stream; its open-entry flags are fetched here): psgpu_fwdflat_search_feats_lists_dev.
        Returns a list of dicts like FwdtreeSearch.search."""
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        off = np.zeros(len(utt_lens) + 1, np.int32); off[1:] = np.cumsum(utt_lens)
        T = int(off[-1]); n = len(utt_lens); mf = int(max(utt_lens)) if n else 0
        if ptm is not None:
            if not torch.is_tensor(senscr):
                senscr = torch.from_numpy(np.ascontiguousarray(senscr, np.float32)).to(dev)
            assert senscr.dtype == torch.float32 and senscr.dim() == 2 and senscr.shape[0] == T
            view = PtmView()
            capi.check(capi.lib().psgpu_ptm_model_view(ptm.h, C.byref(view)), "psgpu_ptm_model_view")
            assert senscr.shape[1] == view.veclen
            d_seed = topn_seed if torch.is_tensor(topn_seed) else torch.from_numpy(np.ascontiguousarray(topn_seed, np.int32)).to(dev)
            d_seed = d_seed.contiguous()              # (kept in a name: the launch below reads it)
            assert d_seed.dtype == torch.int32 and d_seed.numel() == n * view.n_mgau * view.n_feat * view.topn
        else:
            if not torch.is_tensor(senscr):
                senscr = torch.from_numpy(np.ascontiguousarray(senscr, np.int16)).to(dev)
            assert tuple(senscr.shape) == (T, self.n_sen) and senscr.dtype == torch.int16
        d_w1 = None
        if isinstance(bp1, dict):                 # FwdtreeSearch.search(handover=...): everything stays on the device
            d_bp1, d_res1, d_w1 = bp1["bp"], bp1["result"], bp1["w1_ssid"]
            cap1 = int(d_bp1.shape[2])
        else:
            cap1 = max(1, max(int(b.shape[0]) for b in bp1))
            h_bp1 = np.zeros((n, 10, cap1), np.int32); h_res1 = np.zeros((n, 8), np.int32)
            for u, b in enumerate(bp1):
                h_bp1[u, :, :b.shape[0]] = np.asarray(b, np.int32).T
                h_res1[u, 0] = b.shape[0]; h_res1[u, 2] = utt_lens[u]
            d_bp1 = torch.from_numpy(h_bp1).to(dev); d_res1 = torch.from_numpy(h_res1).to(dev)
        if w1_ssid is not None:
            d_w1 = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(w, np.int32) for w in w1_ssid]), np.int32)).to(dev)
            assert tuple(d_w1.shape) == (n, self.n1, self.n_emit)
        d_s, d_o = senscr.contiguous(), torch.from_numpy(off).to(dev)
        bp = torch.zeros((n, 10, bp_cap), dtype=torch.int32, device=dev)
        bss = torch.zeros((n, bss_cap), dtype=torch.int32, device=dev)
        idx = torch.zeros((n, mf + 2), dtype=torch.int32, device=dev)
        step = torch.zeros((n, max(mf, 1), 4), dtype=torch.int32, device=dev)
        res = torch.zeros((n, 8), dtype=torch.int32, device=dev)
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None  # noqa: E731
        sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if ptm is not None and lists is not None:
            tsc, tcw = lists
            fl = C.c_void_p()
            capi.check(capi.lib().psgpu_ptm_batch_open_flags(ptm.h, sp, C.byref(fl)), "psgpu_ptm_batch_open_flags")
            capi.check(capi.lib().psgpu_fwdflat_search_feats_lists_dev(self.h, C.byref(view), p(d_s), p(d_seed), p(tsc), p(tcw), fl, T, p(d_o), n,
                                                                       mf, cap1, p(d_bp1), p(d_res1), p(d_w1), bp_cap, bss_cap, p(bp), p(bss),
                                                                       p(idx), p(step), p(res), sp), "psgpu_fwdflat_search_feats_lists_dev")
        elif ptm is not None:
            capi.check(capi.lib().psgpu_fwdflat_search_feats_dev(self.h, C.byref(view), p(d_s), p(d_seed), p(d_o), n, mf, cap1,
                                                                 p(d_bp1), p(d_res1), p(d_w1), bp_cap, bss_cap, p(bp), p(bss), p(idx),
                                                                 p(step), p(res), sp), "psgpu_fwdflat_search_feats_dev")
        else:
            capi.check(capi.lib().psgpu_fwdflat_search_dev(self.h, p(d_s), C.c_int64(self.n_sen), p(d_o), n, mf, cap1, p(d_bp1), p(d_res1),
                                                           p(d_w1), bp_cap, bss_cap, p(bp), p(bss), p(idx), p(step), p(res), sp),
                       "psgpu_fwdflat_search_dev")
        out = []
        if n == 0:
            return out
        res_h = res.cpu().numpy()
        # one transfer per table for the whole batch (cut to the longest utterance's entries), sliced on the host
        mb, mh = max(1, int(res_h[:, 0].max())), max(1, int(res_h[:, 1].max()))
        bp_h = bp[:, :, :mb].contiguous().cpu().numpy(); bss_h = bss[:, :mh].contiguous().cpu().numpy()
        idx_h = idx.cpu().numpy(); step_h = step.cpu().numpy()
        for u in range(n):
            nb, nh, nfr, status = [int(v) for v in res_h[u, :4]]
            out.append(dict(bp=bp_h[u, :, :nb].T.copy(), bscore_stack=bss_h[u, :nh].copy(),
                            bp_table_idx=idx_h[u, :nfr + 1].copy(), step=step_h[u, :nfr].copy(),
                            n_frame=nfr, status=status))
        return out
