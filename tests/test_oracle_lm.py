"""The language-model oracle (oracle/ps_oracle_lm.c) pinned against the reference:
(1) the known answers of the reference's own unit test test/unit/test_ngram/test_lm_score.c on its
100.lm.bin, (2) the compiled reference's ngram_tg_score on every query of the fixtures
tests/golden/lm_*.npz (`ref_dump lm`, oracle/make_golden.py lm)."""
import os

import numpy as np
import pytest

from pso import OracleLm

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["100", "turtle_decoder", "tidigits_decoder", "synthetic"]


def load(name):
    return dict(np.load(os.path.join(GOLD, "lm_%s.npz" % name)))


def words_of(g):
    return bytes(g["words"]).decode().split("\n")[:-1]


@pytest.mark.parametrize("name", CASES)
def test_lm_oracle_equals_reference_on_every_query(name):
    g = load(name)
    lm = OracleLm(g)
    sc, nu = lm.tg_score_batch(g["queries"])
    bad = np.flatnonzero(sc != g["scores"])
    assert bad.size == 0, (bad[:5], g["queries"][bad[:5]], sc[bad[:5]], g["scores"][bad[:5]])
    assert np.array_equal(nu, g["n_used"])
    # the fixture exercises every depth of the look-up
    assert set(np.unique(g["n_used"])) >= set(range(1, int(g["order"][0]) + 1))


def test_lm_oracle_known_answers_of_the_reference_unit_test():
    """test/unit/test_ngram/test_lm_score.c:15-66.  (TEST_EQUAL_LOG tolerates a difference below
    LOG_EPSILON; the compiled reference's exact answers are in the fixtures, these are the published numbers.)"""
    g = load("100")
    w = {s: i for i, s in enumerate(words_of(g))}
    raw = OracleLm(g, lw=1.0, log_wip=0)
    s, nu = raw.tg_score(w["daines"], w["huggins"], w["david"])
    assert nu == 3 and abs(s - (-9452)) <= 1, s
    s, nu = raw.tg_score(w["huggins"], w["david"], -1)
    assert nu == 2 and abs(s - (-831)) <= 1, s
    assert raw.tg_score(w["daines"], w["huggins"], w["huggins"])[1] == 2
    assert raw.tg_score(w["david"], w["david"], w["david"])[1] == 1
    assert raw.tg_score(w["david"], w["david"], -1)[1] == 1
    # weights 7.5 / 0.5 as the test applies them: -9452 * 7.5 + log(0.5) = -77821
    weighted = OracleLm(g)
    s, _ = weighted.tg_score(w["daines"], w["huggins"], w["david"])
    assert abs(s - (-77821)) <= 8, s


def test_lm_oracle_words_outside_the_model_score_log_zero():
    g = load("turtle_decoder")
    lm = OracleLm(g)
    out = np.flatnonzero(g["widmap"] < 0)
    assert out.size > 0            # filler words of the dictionary are not in the model
    s, nu = lm.tg_score(int(out[0]), 3, 4)
    assert s == int(g["log_zero"][0]) and nu == 0


@pytest.mark.parametrize("static,name", [("en_us_turtle", "turtle_decoder"), ("tidigits", "tidigits_decoder")])
def test_lm_oracle_reproduces_the_whole_dense_table_of_the_search_fixture(static, name):
    """fwdtree_static_*.npz holds ngram_tg_score >> 10 of the reference for EVERY (w3, w2, w1) the search can ask
    (w3 a non-filler base word; -1 = no history): the trie oracle must give the same number for each."""
    st = np.load(os.path.join(GOLD, "fwdtree_static_%s.npz" % static))
    dense = st["lm"]
    n_w = dense.shape[0]
    lm = OracleLm(load(name))
    assert lm.n_words == n_w
    w3 = np.flatnonzero((st["dict_filler"] == 0) & (st["dict_basewid"] == np.arange(n_w)))
    grid = np.stack(np.meshgrid(w3, np.arange(-1, n_w), np.arange(-1, n_w), indexing="ij"), -1).reshape(-1, 3)
    sc, _ = lm.tg_score_batch(grid)
    want = dense[grid[:, 0], grid[:, 1] + 1, grid[:, 2] + 1]
    assert np.array_equal(sc >> 10, want)
