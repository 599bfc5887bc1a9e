"""Host-side mirror of the lexicon-tree search object (ngram_search_t's fwdtree half, reference
src/ngram_search_fwdtree.c); arithmetic in csrc/psgpu_search.hip."""
import ctypes as C

import numpy as np

from . import capi

_NAMES = ["par", "node_ci", "node_ci2", "node_ssid", "node_tmat", "node_child", "node_sib", "node_penult_wid",
          "homophone_set", "w1_wid", "w1_ci", "w1_ci2", "w1_ssid", "w1_tmat", "w1_mpx", "dict_pronlen", "dict_first",
          "dict_last", "dict_last2", "dict_basewid", "dict_filler", "rssid_n", "rssid_ssid", "rssid_cimap", "ldiph_lc",
          "tp", "sseq", "ci_tmat", "lm"]
_DT = {"tp": np.uint8, "sseq": np.uint16}


class _Tables(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _NAMES] + [("n_tmat", C.c_int32), ("n_sseq", C.c_int32)]


class FwdtreeSearch:
    """`static` = the flattened search tables (what `ref_dump fwdtree` writes / an integration reads out
    of its ngram_search_t), `par` = sizes, beams, penalties and special word ids."""

    BP_COLS = ("frame", "valid", "wid", "bp", "score", "s_idx", "real_wid", "prev_real_wid", "last_phone", "last2_phone")

    def __init__(self, static, par, lm=None):
        """lm: an NGramTrieLM over the same dictionary -- language scores are then looked up in the trie on
        the device and the dense table static["lm"] is not needed (any vocabulary the tree fits)."""
        src = dict(static); src["par"] = par
        self._keep = {n: np.ascontiguousarray(src[n], _DT.get(n, np.int32)) for n in _NAMES if not (n == "lm" and lm is not None)}
        t = _Tables(*[self._keep[n].ctypes.data if n in self._keep else None for n in _NAMES],
                    int(self._keep["tp"].shape[0]), int(self._keep["sseq"].shape[0]))
        self.h = C.c_void_p()
        capi.check(capi.lib().psgpu_fwdtree_create(C.byref(self.h), C.byref(t)), "psgpu_fwdtree_create")
        self.lm = lm
        if lm is not None:
            capi.check(capi.lib().psgpu_fwdtree_set_lm(self.h, lm.h), "psgpu_fwdtree_set_lm")
        self.n_sen = int(par[2]); self.n_ci = int(par[0]); self.finish_wid = int(par[20])

    def lds_layout(self):
        """True when the tree-level state of this search lives in LDS (psgpu_fwdtree_layout)."""
        v = C.c_int32()
        capi.check(capi.lib().psgpu_fwdtree_layout(self.h, C.byref(v), None), "psgpu_fwdtree_layout")
        return bool(v.value)

    def close(self):
        if self.h:
            capi.lib().psgpu_fwdtree_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def backtrace_dev(self, bp, idx, result, max_frames, max_words=512):
        """psgpu_fwdtree_backtrace_dev on the device tensors a search left (handover["bp"], ["idx"], ["result"]):
        returns (hyp [n][max_words][4] = wid, sf, ef, path score; hyp_n [n][4] = n words, exit score, exit bp, 0) as numpy."""
        import torch
        n = int(result.shape[0])
        hyp = torch.zeros((n, max_words, 4), dtype=torch.int32, device=bp.device)
        hn = torch.zeros((n, 4), dtype=torch.int32, device=bp.device)
        p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
        capi.check(capi.lib().psgpu_fwdtree_backtrace_dev(self.h, p(bp), p(idx), p(result), n, int(max_frames), int(bp.shape[2]),
                                                          int(max_words), p(hyp), p(hn),
                                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "psgpu_fwdtree_backtrace_dev")
        return hyp.cpu().numpy(), hn.cpu().numpy()

    def search(self, senscr, penalties, utt_lens, bp_cap=16384, bss_cap=1 << 19, raw_scores=False, pl_window=0, handover=None,
               mpx_in=None, mpx_out=None, cuts=None, lag=0):
        """mpx_in: [n][n_mpx][n_emit] int32 per-state ssids the permanent multiplexed channels start with (a session's carry-over,
        psgpu_fwdtree_search_session_dev) or None = a fresh decoder; mpx_out: a dict that receives {"mpx": the ssids they end with}.
        senscr [T][n_sen] int16 and penalties [T][n_ci] int32 for utterances back to back (numpy arrays, or
        torch tensors already on the device).  raw_scores: un-normalised rows + phone-loop output, see psgpu.h.
        handover: a dict that receives the device buffers a second pass takes over (FwdflatSearch.search(bp1=...)).
        Returns a list of dicts (bp [n][10], bscore_stack, bp_table_idx, step [frames][4], status) per utterance."""
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        off = np.zeros(len(utt_lens) + 1, np.int32); off[1:] = np.cumsum(utt_lens)
        T = int(off[-1]); n = len(utt_lens); mf = int(max(utt_lens)) if n else 0
        if not torch.is_tensor(senscr):
            senscr = torch.from_numpy(np.ascontiguousarray(senscr, np.int16)).to(dev)
        if not torch.is_tensor(penalties):
            penalties = torch.from_numpy(np.ascontiguousarray(penalties, np.int32)).to(dev)
        assert tuple(senscr.shape) == (T, self.n_sen) and tuple(penalties.shape) == (T, self.n_ci)
        assert senscr.dtype == torch.int16 and penalties.dtype == torch.int32
        d_s, d_p, d_o = senscr.contiguous(), penalties.contiguous(), torch.from_numpy(off).to(dev)
        bp = torch.zeros((n, 10, bp_cap), dtype=torch.int32, device=dev)
        bss = torch.zeros((n, bss_cap), dtype=torch.int32, device=dev)
        idx = torch.zeros((n, mf + 2), dtype=torch.int32, device=dev)
        step = torch.zeros((n, max(mf, 1), 4), dtype=torch.int32, device=dev)
        res = torch.zeros((n, 8), dtype=torch.int32, device=dev)
        p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
        w1 = None
        if handover is not None:
            w1 = torch.zeros((n, int(self._keep["par"][6]), int(self._keep["par"][1])), dtype=torch.int32, device=dev)
            handover.update(bp=bp, result=res, w1_ssid=w1, idx=idx, bp_cap=bp_cap, max_frames=mf)
        n_mpx = int(capi.lib().psgpu_fwdtree_n_mpx_channels(self.h)); ne = int(self._keep["par"][1])
        d_mi = d_mo = None
        if mpx_in is not None:
            d_mi = torch.from_numpy(np.ascontiguousarray(mpx_in, np.int32).reshape(n, n_mpx, ne)).to(dev)
        if mpx_out is not None:
            d_mo = torch.zeros((n, n_mpx, ne), dtype=torch.int32, device=dev)
        # cuts: ONE utterance searched in several calls (psgpu_fwdtree_search_resume): up to each cut's frame count minus `lag`, then
        # to the end -- same buffers, same tables as one call; self.searched receives the frames searched after every call
        calls = [(d_o, 0, 0)] if cuts is None else \
            [(torch.tensor([0, int(c)], dtype=torch.int32, device=dev), lag, (1 if i == 0 else 3)) for i, c in enumerate(cuts)] + [(d_o, 0, 2)]
        self.searched = []
        self.grown = []
        for attempt in range(40):
          for o, lg, mode in calls:
              if cuts is not None:
                  capi.check(capi.lib().psgpu_fwdtree_search_lag(self.h, int(lg)), "psgpu_fwdtree_search_lag")
                  capi.check(capi.lib().psgpu_fwdtree_search_resume(self.h, mode), "psgpu_fwdtree_search_resume")
              capi.check(capi.lib().psgpu_fwdtree_search_session_dev(
                  self.h, p(d_s), C.c_int64(self.n_sen), p(d_p), p(o), n, mf, bp_cap, bss_cap, p(bp), p(bss), p(idx), p(step), p(res),
                  int(raw_scores), int(pl_window), p(w1) if w1 is not None else None, p(d_mi) if d_mi is not None else None,
                  p(d_mo) if d_mo is not None else None, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                  "psgpu_fwdtree_search_session_dev")
              torch.cuda.current_stream().synchronize()   # the entry is asynchronous; d_s / d_p / d_o must outlive the kernel
              if cuts is not None:
                  self.searched.append(int(res[0, 2].item()))
          # status 4 / 5 (slab layouts): a frame listed more tree nodes than the compact channels hold / needed more blocks of the
          # right-context channels' pool than there are -- the capacity is doubled (psgpu_fwdtree_grow) and the search repeated, as
          # psgpu_decode_fetch_hyps does for the pipeline
          # (a search in several calls -- cuts -- that meets a capacity starts over from its first cut with the larger arrays;
          #  nothing left to grow: the status stays in the result records, as the pipeline leaves it)
          st = res[:, 3].cpu().numpy() if n else np.zeros(0, np.int32)
          need = [int(v) for v in st if int(v) in (4, 5, 6)]
          if not need:
              break
          if capi.lib().psgpu_fwdtree_grow(self.h, need[0]) != 0:
              break
          self.grown.append(need[0])
          if cuts is not None:
              self.searched = []
        if mpx_out is not None:
            mpx_out["mpx"] = d_mo.cpu().numpy()
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


def backtrace(result, finish_wid):
    """Best exit and its word sequence from a back-pointer table, as ngram_search_find_exit
    (reference src/ngram_search.c:500-544: the </s> entry of the last frame that has entries, else the
    best-scoring one) and ngram_search_bp_hyp / the segment iterator (:546-581, 903-1010) walk it.
    Returns (path score, [(wid, start_frame, end_frame), ...])."""
    bp, idx = result["bp"], result["bp_table_idx"]
    n_frame = result["n_frame"]
    if n_frame == 0 or bp.shape[0] == 0:
        return None, []
    f = n_frame - 1
    end = int(idx[f])
    while f >= 0 and int(idx[f]) == end:
        f -= 1
    if f < 0:
        return None, []
    best, best_score = -1, -(1 << 29)
    for b in range(int(idx[f]), end):
        if int(bp[b, 2]) == finish_wid or int(bp[b, 4]) > best_score:
            best, best_score = b, int(bp[b, 4])
        if int(bp[b, 2]) == finish_wid:
            break
    words = []
    b = best
    while b != -1:
        prev = int(bp[b, 3])
        sf = 0 if prev == -1 else int(bp[prev, 0]) + 1
        words.append((int(bp[b, 2]), sf, int(bp[b, 0])))
        b = prev
    return best_score, words[::-1]
