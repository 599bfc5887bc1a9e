"""The flat-lexicon second-pass oracle (oracle/ps_oracle_flat.c: ngram_search_fwdflat.c restated on flat
tables) pinned against the unmodified reference.

`ref_dump fwdflat` recorded, for real two-pass decodes (-fwdtree yes -fwdflat yes -bestpath no), what the
first pass handed over (its back-pointer table; the ssids its permanent single-phone channels were left
with), the senone scores every frame of the second pass was handed, and what the second pass produced.
The oracle, given the same inputs, must reproduce per frame the active senone list
(compute_fwdflat_sen_active + acmod_flags2list), the best score and the back-pointer count, and at the end
the complete back-pointer table (ten columns), the right-context score stack and the frame marks -- bit for
bit, including the float-weighted language scores of fwdflat_word_transition.  Cases: en-us + turtle LM on
three recordings (one with a narrow start-frame window, end-point filter 2, other beams and another
fwdflatlw), tidigits (5-state HMMs, semi-continuous scores), the 715-word task with trie language scores."""
import numpy as np
import pytest

import pso
from test_oracle_golden import _load

FLAT_CASES = ["goforward", "numbers", "something_efwid2_sfwin8", "man_ah_2934za", "medium_numbers"]


def load_flat(case):
    g = _load("fwdflat_trace_%s.npz" % case)
    static = bytes(g["static"]).decode()
    return g, _load("fwdtree_static_%s.npz" % static), _load("fwdflat_static_%s.npz" % static)


@pytest.mark.parametrize("case", FLAT_CASES)
def test_fwdflat_oracle_matches_reference(case):
    g, st, fst = load_flat(case)
    o = pso.OracleFwdflat(st, fst, g, lm=pso.OracleLm(fst) if "lm" not in st else None)
    nfr = int(g["n_frame"][0])
    o.start(g["bp1"], nfr, g["flat_w1_ssid"])
    off, act, scr = g["flat_act_off"], g["flat_act"], g["flat_scr"]
    n = int(g["flat_n_steps"][0])
    assert n == nfr
    for i in range(n):
        a0, a1 = int(off[i]), int(off[i + 1])
        assert np.array_equal(o.active_list(i), act[a0:a1]), "frame %d: active senone list" % i
        o.step(i, act[a0:a1], scr[a0:a1], int(g["flat_rest"][i]))
        assert (o.best_score(), o.bpidx()) == (int(g["flat_best"][i]), int(g["flat_bpidx"][i])), "frame %d" % i
    o.finish(nfr)
    bp = o.bp_table()
    assert bp.shape == g["bp"].shape
    bad = np.nonzero((bp != g["bp"]).any(axis=1))[0]
    assert bad.size == 0, "first differing back-pointer %d: %r vs %r" % (bad[0], bp[bad[0]], g["bp"][bad[0]])
    assert np.array_equal(o.bscore_stack(), g["bscore_stack"])
    assert np.array_equal(o.bp_table_idx(nfr), g["bp_table_idx"])


def test_fwdflat_oracle_large_vocabulary(big_flat_trace):
    """The second-pass oracle at full scale: every base word of cmudict in the dictionary (134,865 entries), the
    synthetic large LM (trie language scores), the first pass's 2,613-entry table; the fixture is produced at test
    time by the compiled reference.  Vocabulary, per-frame active lists / best scores / back-pointer counts and the
    final tables must be the reference's."""
    g = big_flat_trace
    assert bytes(g["hyp"]).decode() == "go forward ten meters" and int(g["par"][3]) > 100000
    o = pso.OracleFwdflat(g, g, g, lm=pso.OracleLm(g))
    nfr = int(g["n_frame"][0])
    o.start(g["bp1"], nfr, g["flat_w1_ssid"])
    assert np.array_equal(o.wordlist(), g["flat_wordlist"])
    off, act, scr = g["flat_act_off"], g["flat_act"], g["flat_scr"]
    for i in range(nfr):
        a0, a1 = int(off[i]), int(off[i + 1])
        assert np.array_equal(o.active_list(i), act[a0:a1]), "frame %d: active senone list" % i
        o.step(i, act[a0:a1], scr[a0:a1], int(g["flat_rest"][i]))
        assert (o.best_score(), o.bpidx()) == (int(g["flat_best"][i]), int(g["flat_bpidx"][i])), "frame %d" % i
    o.finish(nfr)
    assert np.array_equal(o.bp_table(), g["bp"]) and np.array_equal(o.bscore_stack(), g["bscore_stack"])
    assert np.array_equal(o.bp_table_idx(nfr), g["bp_table_idx"])
