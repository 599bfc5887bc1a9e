"""The Python wrappers the GPU tests and the bench drive (pocketsphinx_amd.search.FwdtreeSearch, .flat.FwdflatSearch, incl.
the device hand-over between them, the bulk result read-back and the scoring entry) executed on the CPU: `torch.device` is
redirected to the CPU and `capi.lib()` to the workgroup simulator's library, so every line of the wrappers runs -- with
CPU tensors standing in for device memory -- and the results are compared with the goldens.  This pins the host-side
plumbing (shapes, strides, argument order, pointer lifetimes) of code paths that otherwise only execute on an MI355X."""
import ctypes as C

import numpy as np
import pytest

import simlib
from test_flat_hostsim import check_flat, flat_rows
from test_oracle_flat import load_flat
from test_oracle_golden import _load
from test_search_gpu import _check, _inputs


class _Stream:
    cuda_stream = 0

    def synchronize(self):
        pass


class _SimLibWithPtmView:
    """the simulator's library + psgpu_ptm_model_view (which lives in psgpu_ptm.hip, not compiled by the simulator)"""

    def __init__(self):
        self._l = simlib.lib()

    def __getattr__(self, name):
        return getattr(self._l, name)

    @staticmethod
    def psgpu_ptm_model_view(h, view_ref):
        t, keep = h                                    # (tables, arrays kept alive)
        v = view_ref._obj
        for n in ("mean", "var", "det", "mixw", "sen2cb", "logadd8"):
            setattr(v, n, keep[n].ctypes.data)
        fl = [int(x) for x in t["featlen"]]
        v.n_mgau, v.n_feat, v.n_density, v.n_sen = int(t["n_mgau"][0]), int(t["n_feat"][0]), int(t["n_density"][0]), int(t["n_sen"][0])
        v.veclen, v.topn, v.logadd8_size = sum(fl), int(t["max_topn"][0]), int(keep["logadd8"].size)
        for i, x in enumerate(fl):
            v.featlen[i] = x; v.featoff[i] = sum(fl[:i])
        return 0


class _FakePtmModel:
    def __init__(self, t):
        keep = dict(mean=np.ascontiguousarray(t["mean"], np.float32), var=np.ascontiguousarray(t["var"], np.float32),
                    det=np.ascontiguousarray(t["det"], np.float32), mixw=np.ascontiguousarray(t["mixw"], np.uint8),
                    sen2cb=np.ascontiguousarray(t["sen2cb"], np.uint8), logadd8=np.ascontiguousarray(t["logadd8"], np.uint8))
        self.h = (t, keep)


@pytest.fixture
def on_simulator(monkeypatch):
    import torch
    import pocketsphinx_amd.capi as capi
    lib = _SimLibWithPtmView()
    cpu = torch.device("cpu")
    monkeypatch.setattr(capi, "lib", lambda: lib)
    monkeypatch.setattr(capi, "check", simlib.check)
    monkeypatch.setattr(torch, "device", lambda *a, **k: cpu)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    return torch


@pytest.mark.parametrize("layout", ["lds", "slab"])
def test_tree_search_wrapper_and_handover(on_simulator, layout, monkeypatch):
    import pocketsphinx_amd as P
    g1 = _load("fwdtree_trace_goforward.npz")
    g, st, fst = load_flat("goforward")
    monkeypatch.setenv("PSGPU_FWDTREE_LAYOUT", layout)
    s1 = P.FwdtreeSearch(st, g1["par"])
    assert s1.lds_layout() == (layout == "lds")
    rows1, pen1 = _inputs(g1, s1.n_sen)
    # a batch of two, so that the bulk read-back slices per utterance
    h = {}
    out = s1.search(np.concatenate([rows1, rows1]), np.concatenate([pen1, pen1]), [rows1.shape[0]] * 2, handover=h)
    for r in out:
        _check(r, g1, "wrapper")
    assert np.array_equal(h["w1_ssid"][1].numpy(), g["flat_w1_ssid"])
    s2 = P.FwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"])
    rows2 = flat_rows(g, s2.n_sen)
    for r in s2.search(np.concatenate([rows2, rows2]), [rows2.shape[0]] * 2, h):
        check_flat(r, g, "wrapper, hand-over")
    # the other ways in: host tables, and the kernel scoring its own senones from tensors already "on the device"
    check_flat(s2.search(rows2, [rows2.shape[0]], [g["bp1"]], [g["flat_w1_ssid"]])[0], g, "wrapper, host tables")
    import pso
    torch = on_simulator
    feats = torch.from_numpy(np.ascontiguousarray(g["flat_feat"], np.float32))
    seed = torch.from_numpy(np.ascontiguousarray(g["flat_ptm_seed"], np.int32).reshape(1, -1, g["flat_ptm_seed"].shape[-1]))
    r = s2.search(feats, [feats.shape[0]], [g["bp1"]], [g["flat_w1_ssid"]], ptm=_FakePtmModel(pso.load_tables()), topn_seed=seed)[0]
    check_flat(r, g, "wrapper, scoring")
    s1.close(); s2.close()


def test_trie_lm_wrapper(on_simulator):
    import pocketsphinx_amd as P
    g, st, fst = load_flat("medium_numbers")
    lm = P.NGramTrieLM(fst)
    s2 = P.FwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"], lm=lm)
    check_flat(s2.search(flat_rows(g, s2.n_sen), [int(g["flat_n_steps"][0])], [g["bp1"]], [g["flat_w1_ssid"]])[0], g, "wrapper, trie")
    s2.close(); lm.close()
