"""Host-side mirror of the n-gram model as the search uses it: ngram_tg_score on a model set holding
one trie model (reference src/lm/ngram_model_set.c, ngram_model_trie.c, lm_trie.c); arithmetic in
csrc/psgpu_lm_dev.h."""
import ctypes as C

import numpy as np

from . import capi

MAX_LEVELS = 4


class _LmTables(C.Structure):
    _fields_ = [("order", C.c_int32), ("n_unigrams", C.c_int32), ("n_words", C.c_int32),
                ("unigrams", C.c_void_p), ("ngram_mem", C.c_void_p), ("ngram_mem_size", C.c_uint64),
                ("level_offset", C.c_uint32 * MAX_LEVELS), ("total_bits", C.c_uint32 * MAX_LEVELS),
                ("word_bits", C.c_uint32 * MAX_LEVELS), ("word_mask", C.c_uint32 * MAX_LEVELS),
                ("max_vocab", C.c_uint32 * MAX_LEVELS), ("next_bits", C.c_uint32 * MAX_LEVELS),
                ("next_mask", C.c_uint32 * MAX_LEVELS),
                ("quant", C.c_void_p), ("lw", C.c_float), ("log_wip", C.c_int32), ("log_zero", C.c_int32),
                ("widmap", C.c_void_p), ("class_weight", C.c_void_p), ("histmap", C.c_void_p)]


class NGramTrieLM:
    """`g`: the model's tables under the names integration/psgpu_lm_tables.c / `ref_dump lm` give them
    (order, n_unigrams, n_words, unigrams [n+1][3], ngram_mem, levels [order-1][7], quant, lw, log_wip,
    log_zero, widmap).  lw / log_wip override the weights (ngram_model_apply_weights)."""

    def __init__(self, g, lw=None, log_wip=None):
        order = int(np.asarray(g["order"]).ravel()[0])
        t = _LmTables()
        t.order = order; t.n_unigrams = int(np.asarray(g["n_unigrams"]).ravel()[0]); t.n_words = int(np.asarray(g["n_words"]).ravel()[0])
        self._keep = dict(unigrams=np.ascontiguousarray(g["unigrams"]).view(np.uint32),
                          ngram_mem=np.ascontiguousarray(g["ngram_mem"], np.uint8),
                          widmap=np.ascontiguousarray(g["widmap"], np.int32))
        t.unigrams = self._keep["unigrams"].ctypes.data
        t.ngram_mem = self._keep["ngram_mem"].ctypes.data; t.ngram_mem_size = self._keep["ngram_mem"].size
        lev = np.ascontiguousarray(g["levels"]).view(np.uint32).reshape(-1, 7) if order > 1 else np.zeros((0, 7), np.uint32)
        for l in range(min(order - 1, lev.shape[0], MAX_LEVELS)):      # (an impossible order is psgpu_lm_create's to refuse)
            (t.level_offset[l], t.total_bits[l], t.word_bits[l], t.word_mask[l], t.max_vocab[l], t.next_bits[l],
             t.next_mask[l]) = (int(v) for v in lev[l])
        if order > 1:
            self._keep["quant"] = np.ascontiguousarray(g["quant"], np.float32)
            t.quant = self._keep["quant"].ctypes.data
        t.lw = float(np.asarray(g["lw"]).ravel()[0]) if lw is None else float(lw)
        t.log_wip = int(np.asarray(g["log_wip"]).ravel()[0]) if log_wip is None else int(log_wip)
        t.log_zero = int(np.asarray(g["log_zero"]).ravel()[0])
        t.widmap = self._keep["widmap"].ctypes.data
        if "class_weight" in g and g["class_weight"] is not None:      # word classes: psgpu_lm_tables_t.class_weight / .histmap
            self._keep["class_weight"] = np.ascontiguousarray(g["class_weight"], np.int32)
            self._keep["histmap"] = np.ascontiguousarray(g["histmap"], np.int32)
            t.class_weight = self._keep["class_weight"].ctypes.data; t.histmap = self._keep["histmap"].ctypes.data
        self.n_words = t.n_words; self.order = order
        self.h = C.c_void_p()
        capi.check(capi.lib().psgpu_lm_create(C.byref(self.h), C.byref(t)), "psgpu_lm_create")

    def close(self):
        if self.h:
            capi.lib().psgpu_lm_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tg_score(self, queries):
        """queries [n][3] = (w3, w2, w1) dictionary word ids, -1 for an absent history word.  Returns
        (scores [n] int32 = ngram_tg_score(lmset, w3, w2, w1, &n_used), n_used [n])."""
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        q = queries if torch.is_tensor(queries) else torch.from_numpy(np.ascontiguousarray(queries, np.int32)).to(dev)
        assert q.dtype == torch.int32 and q.dim() == 2 and q.shape[1] == 3
        w3, w2, w1 = (q[:, i].contiguous() for i in range(3))
        n = int(q.shape[0])
        sc = torch.empty(n, dtype=torch.int32, device=dev); nu = torch.empty(n, dtype=torch.int32, device=dev)
        p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
        capi.check(capi.lib().psgpu_lm_tg_score_dev(self.h, p(w3), p(w2), p(w1), C.c_int64(n), p(sc), p(nu),
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "psgpu_lm_tg_score_dev")
        if torch.is_tensor(queries):
            return sc, nu
        return sc.cpu().numpy(), nu.cpu().numpy()


class NGramSetLM(NGramTrieLM):
    """A model set looked up WITHOUT a current model (ngram_model_set_score with cur == -1, reference src/lm/ngram_model_set.c:685-727):
    the log-sum of lweights[i] + member i's look-up through the set's log-add table.  members: NGramTrieLM objects over the set's word
    ids (they stay alive with this object); addtab: logadd_t.table as integers; add_zero: logmath_get_zero; log_zero: the set's."""

    def __init__(self, members, lweights, addtab, add_zero, log_zero):
        self.members = list(members)
        hs = (C.c_void_p * len(self.members))(*[m.h for m in self.members])
        lw = np.ascontiguousarray(lweights, np.int32)
        tab = np.ascontiguousarray(addtab, np.uint32)
        self.n_words = self.members[0].n_words; self.order = max(m.order for m in self.members)
        self.h = C.c_void_p()
        capi.check(self._lib().psgpu_lm_create_interp(C.byref(self.h), hs, lw.ctypes.data_as(C.c_void_p), len(self.members),
                                                      tab.ctypes.data_as(C.c_void_p), 4, int(tab.size), int(add_zero), int(log_zero)),
                   "psgpu_lm_create_interp")

    @staticmethod
    def _lib():
        return capi.lib()
