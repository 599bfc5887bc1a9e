"""GPU parity of the lexicon-tree search kernel in its ACTIVE_LIST formulation (psgpu_fwdtree_set_mode,
include/psgpu.h: per-frame work proportional to the active channels, word-level positions by workgroup prefix
sums, 1024 work-items per utterance on large trees -- DESIGN.md 7.2) against the same reference dumps as the
default formulation (tests/test_search_gpu.py, tests/test_lm_gpu.py): back-pointer table, score stack, frame
marks and per-frame best scores bit for bit.  The kernel source in this mode is also run on the CPU by
tests/test_search_hostsim.py; this file is its check on the MI355X (it sorts last on purpose: the mode is
opt-in and was written in a session that had no GPU minutes left)."""
import numpy as np
import pytest

from conftest import run_isolated
from test_oracle_golden import _load
from test_oracle_lm import load as lm_load
from test_oracle_search import CASES, MEDIUM_CASES, make_big_trace
from test_search_gpu import _check, _inputs

pytestmark = pytest.mark.gpu
ME = "test_zz_search_active_list_gpu"      # every test body runs in a child process (conftest.run_isolated)


@pytest.mark.parametrize("case", CASES + MEDIUM_CASES)
def test_fwdtree_kernel_active_list_matches_reference(case):
    run_isolated(ME, "impl_matches_reference", case)


def impl_matches_reference(case):
    import pocketsphinx_amd as P
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    lm = P.NGramTrieLM(st) if "lm" not in st else None
    s = P.FwdtreeSearch(st, g["par"], lm=lm, mode=P.FwdtreeSearch.ACTIVE_LIST)
    rows, pen = _inputs(g, s.n_sen)
    _check(s.search(rows, pen, [rows.shape[0]])[0], g, case)
    s.close()


def test_fwdtree_kernel_active_list_batch_and_trie():
    """several utterances per launch, language scores from the device trie"""
    run_isolated(ME, "impl_batch_and_trie")


def impl_batch_and_trie():
    import pocketsphinx_amd as P
    names = ["goforward", "numbers", "goforward"]
    gs = [_load("fwdtree_trace_%s.npz" % n) for n in names]
    st = _load("fwdtree_static_en_us_turtle.npz")
    s = P.FwdtreeSearch(st, gs[0]["par"], lm=P.NGramTrieLM(lm_load("turtle_decoder")), mode=P.FwdtreeSearch.ACTIVE_LIST)
    ins = [_inputs(g, s.n_sen) for g in gs]
    out = s.search(np.concatenate([i[0] for i in ins]), np.concatenate([i[1] for i in ins]), [i[0].shape[0] for i in ins])
    for r, g, n in zip(out, gs, names):
        _check(r, g, n)
    s.close()


def test_fwdtree_kernel_active_list_full_cmudict_vocabulary(tmp_path):
    """134,865 words, 248 k tree channels, ~8 k active channels per frame: the task the mode exists for"""
    import pso
    import os
    if not os.path.exists(os.path.join(pso.REF_DIR, "ref_dump")):
        pytest.skip("oracle/_ref (compiled reference + staged data) not built")
    run_isolated(ME, "impl_full_cmudict", str(tmp_path), timeout=1200)


def impl_full_cmudict(tmp):
    import pocketsphinx_amd as P
    g = make_big_trace(tmp)
    s = P.FwdtreeSearch(g, g["par"], lm=P.NGramTrieLM(g), mode=P.FwdtreeSearch.ACTIVE_LIST)
    rows, pen = _inputs(g, s.n_sen)
    _check(s.search(rows, pen, [rows.shape[0]])[0], g, "cmudict")
    _check(s.search(np.concatenate([rows, rows]), np.concatenate([pen, pen]), [rows.shape[0]] * 2)[1], g, "cmudict x2")
    s.close()


def test_fwdtree_set_mode_rejects_unknown_modes():
    run_isolated(ME, "impl_rejects_unknown_modes")


def impl_rejects_unknown_modes():
    import pocketsphinx_amd as P
    g = _load("fwdtree_trace_goforward.npz")
    with pytest.raises(P.PsgpuError):
        P.FwdtreeSearch(_load("fwdtree_static_en_us_turtle.npz"), g["par"], mode=7)
