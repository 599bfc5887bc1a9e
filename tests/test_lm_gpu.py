"""GPU parity of the device language model (psgpu_lm_*, SURVEY 8f-3: ngram_tg_score through the model
set, the bit-packed trie with its interpolation search, and the quantisation tables) against the
compiled reference's answers (tests/golden/lm_*.npz, `ref_dump lm`), against the reference's complete
dense table of the search fixtures, and inside the lexicon-tree search kernel."""
import os

import numpy as np
import pytest

from test_oracle_golden import _load
from test_oracle_lm import CASES, load
from test_oracle_search import CASES as SEARCH_CASES, MEDIUM_CASES, big_trace  # noqa: F401 (fixture)
from test_search_gpu import _check, _inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CASES)
def test_device_lm_equals_reference_on_every_query(name):
    import pocketsphinx_amd as P
    g = load(name)
    lm = P.NGramTrieLM(g)
    sc, nu = lm.tg_score(g["queries"])
    bad = np.flatnonzero(sc != g["scores"])
    assert bad.size == 0, (bad[:5], g["queries"][bad[:5]], sc[bad[:5]], g["scores"][bad[:5]])
    assert np.array_equal(nu, g["n_used"])
    lm.close()


def test_device_lm_known_answers_of_the_reference_unit_test():
    """test/unit/test_ngram/test_lm_score.c:15-66 on 100.lm.bin (tolerance of its TEST_EQUAL_LOG: 200)"""
    import pocketsphinx_amd as P
    g = load("100")
    w = {s: i for i, s in enumerate(bytes(g["words"]).decode().split("\n")[:-1])}
    q = np.array([[w["daines"], w["huggins"], w["david"]], [w["huggins"], w["david"], -1],
                  [w["daines"], w["huggins"], w["huggins"]], [w["david"], w["david"], w["david"]]], np.int32)
    raw = P.NGramTrieLM(g, lw=1.0, log_wip=0)
    sc, nu = raw.tg_score(q)
    assert abs(int(sc[0]) + 9452) < 200 and abs(int(sc[1]) + 831) < 200 and nu.tolist() == [3, 2, 2, 1]
    sc, _ = P.NGramTrieLM(g).tg_score(q)          # weights 7.5 / 0.5
    assert abs(int(sc[0]) + 77821) < 200


@pytest.mark.parametrize("static,name", [("en_us_turtle", "turtle_decoder"), ("tidigits", "tidigits_decoder")])
def test_device_lm_reproduces_the_whole_dense_table(static, name):
    """every (w3, w2, w1) the search can ask, -1 histories included: 1.5 M look-ups for the turtle task"""
    import pocketsphinx_amd as P
    st = _load("fwdtree_static_%s.npz" % static)
    dense = st["lm"]
    n_w = dense.shape[0]
    lm = P.NGramTrieLM(load(name))
    w3 = np.flatnonzero((st["dict_filler"] == 0) & (st["dict_basewid"] == np.arange(n_w)))
    grid = np.stack(np.meshgrid(w3, np.arange(-1, n_w), np.arange(-1, n_w), indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    sc, _ = lm.tg_score(grid)
    assert np.array_equal(sc >> 10, dense[grid[:, 0], grid[:, 1] + 1, grid[:, 2] + 1])


def test_device_lm_edge_cases():
    import torch
    import pocketsphinx_amd as P
    g = load("turtle_decoder")
    lm = P.NGramTrieLM(g)
    sc, nu = lm.tg_score(np.zeros((0, 3), np.int32))
    assert sc.shape == (0,)
    # dictionary words the model does not know (fillers) score log_zero whatever the history
    out = np.flatnonzero(g["widmap"] < 0)
    q = np.array([[out[0], 3, 4], [out[-1], -1, -1]], np.int32)
    sc, nu = lm.tg_score(q)
    assert sc.tolist() == [int(g["log_zero"][0])] * 2 and nu.tolist() == [0, 0]
    # device-resident queries stay on the device
    sc, nu = lm.tg_score(torch.from_numpy(g["queries"][:100]).cuda())
    assert torch.is_tensor(sc) and np.array_equal(sc.cpu().numpy(), g["scores"][:100])
    # a history word unknown to the model truncates the history there (ngram_model_trie.c:724-731)
    known = int(np.flatnonzero(g["widmap"] >= 0)[5])
    a, _ = lm.tg_score(np.array([[known, out[0], 7]], np.int32))
    b, _ = lm.tg_score(np.array([[known, -1, -1]], np.int32))
    assert a[0] == b[0]


def test_device_lm_create_rejects_inconsistent_tables():
    import pocketsphinx_amd as P
    g = dict(load("turtle_decoder"))
    bad = dict(g); bad["widmap"] = g["widmap"].copy(); bad["widmap"][3] = int(g["n_unigrams"][0]) + 5
    with pytest.raises(P.PsgpuError):
        P.NGramTrieLM(bad)
    bad = dict(g); bad["order"] = np.array([7], np.int32)
    with pytest.raises(P.PsgpuError):
        P.NGramTrieLM(bad)


@pytest.mark.parametrize("case", SEARCH_CASES + MEDIUM_CASES)
def test_fwdtree_kernel_with_the_trie_lm_matches_reference(case):
    """The lexicon-tree search with its language scores looked up in the trie on the device (no dense table
    uploaded): back-pointer table, score stack and per-frame scores are still the reference's.  The medium
    cases (715 words, 2820 tree nodes, forced word-exit pruning) cannot run any other way."""
    import pocketsphinx_amd as P
    g = _load("fwdtree_trace_%s.npz" % case)
    static = bytes(g["static"]).decode()
    st = _load("fwdtree_static_%s.npz" % static)
    lm = P.NGramTrieLM(st if "lm" not in st else load({"en_us_turtle": "turtle_decoder", "tidigits": "tidigits_decoder"}[static]))
    s = P.FwdtreeSearch(st, g["par"], lm=lm)
    rows, pen = _inputs(g, s.n_sen)
    _check(s.search(rows, pen, [rows.shape[0]])[0], g, case)
    s.close()


def test_fwdtree_set_lm_checks_the_vocabulary():
    import pocketsphinx_amd as P
    g = _load("fwdtree_trace_goforward.npz")
    st = _load("fwdtree_static_en_us_turtle.npz")
    with pytest.raises(P.PsgpuError):
        P.FwdtreeSearch(st, g["par"], lm=P.NGramTrieLM(load("tidigits_decoder")))


def test_fwdtree_kernel_full_cmudict_vocabulary(big_trace):  # noqa: F811
    """The search kernel at the scale of SURVEY 8d config 3's large-vocabulary decode: 134,865 dictionary entries,
    a lexicon tree of 248 k channels, 3.7 M (word, right context) last-phone slots, language scores from the
    device trie (126 k unigrams).  Beyond the LDS scratch (4096 tree nodes / 1024 words) the list and word scratch
    live in the utterance's slab; the algorithm is the same, so the back-pointer table, score stack, frame marks
    and per-frame best scores must again be the reference's (fixture made at test time by the compiled reference).
    This version still walks the whole tree every frame: correct, not yet fast."""
    import pocketsphinx_amd as P
    g = big_trace
    lm = P.NGramTrieLM(g)
    s = P.FwdtreeSearch(g, g["par"], lm=lm)
    rows, pen = _inputs(g, s.n_sen)
    _check(s.search(rows, pen, [rows.shape[0]])[0], g, "cmudict")
    s.close()
