"""The lexicon-tree search oracle (oracle/ps_oracle_search.c: prune / transitions / back-pointer
table) pinned against the unmodified reference.

`ref_dump fwdtree` recorded, for real decodes, the static search tables the reference built, the
senone scores and phone-loop penalties its search was handed frame by frame, and what it
produced.  The oracle, fed the same inputs, must reproduce per frame the active senone list
(compute_sen_active + acmod_flags2list, bridging entries included), the best score, the
last-phone best score and the back-pointer count, and at the end the complete back-pointer
table (frame, valid, wid, bp, score, s_idx, real_wid, prev_real_wid, last_phone, last2_phone),
the right-context score stack and the per-frame marks -- bit for bit.  Cases: en-us + turtle LM
(3-state HMMs, PTM scores) on two recordings, with histogram pruning (maxhmmpf) and absolute
word-exit pruning (maxwpf) forced on, without the phone-loop look-ahead, and tidigits
(5-state HMMs, semi-continuous scores, its own LM and dictionary)."""
import os

import numpy as np
import pytest

import pso
from test_oracle_golden import _load

CASES = ["goforward", "numbers", "goforward_maxhmmpf60_maxwpf3", "something_plwindow0", "man_ah_2934za"]
# a 715-word task (oracle/make_medium_task.py): no dense LM table, the static fixture carries the model's trie
# tables and the language scores come from the trie oracle (ps_oracle_lm.c) / the device trie
MEDIUM_CASES = ["medium_goforward", "medium_numbers_maxwpf8"]


@pytest.mark.parametrize("parallel", [0, 1, 2])
@pytest.mark.parametrize("case", CASES + MEDIUM_CASES)
def test_fwdtree_oracle_matches_reference(case, parallel):
    """parallel = 1: the tree pruning in its data-parallel formulation (per-node decisions on a
    snapshot of the evaluated state + prefix sums for the list positions, prune_tree_parallel) --
    the form the device kernel uses -- must give the same tables as the sequential walk.
    parallel = 2: the same with work proportional to the active part of the tree (prune_tree_list)."""
    import ctypes as C
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    o = pso.OracleFwdtree(st, g["par"], lm=pso.OracleLm(st) if "lm" not in st else None)
    pso.lib().pso_ft_set_parallel.argtypes = [C.c_void_p, C.c_int]
    pso.lib().pso_ft_set_parallel(o.h, parallel)
    o.start()
    off, act, scr = g["step_act_off"], g["step_act"], g["step_scr"]
    n = int(g["n_steps"][0])
    for i in range(n):
        fr = int(g["step_frame"][i])
        a0, a1 = int(off[i]), int(off[i + 1])
        assert np.array_equal(o.active_list(fr), act[a0:a1]), "frame %d: active senone list" % fr
        o.step(fr, act[a0:a1], scr[a0:a1], int(g["step_rest"][i]), g["step_pen"][i])
        assert (o.best_score(), o.last_phone_best_score(), o.bpidx()) == \
            (int(g["step_best"][i]), int(g["step_lpbest"][i]), int(g["step_bpidx"][i])), "frame %d" % fr
    nfr = int(g["n_frame"][0])
    o.finish(nfr)
    bp = o.bp_table()
    assert bp.shape == g["bp"].shape
    bad = np.nonzero((bp != g["bp"]).any(axis=1))[0]
    assert bad.size == 0, "first differing back-pointer %d: %r vs %r" % (bad[0], bp[bad[0]], g["bp"][bad[0]])
    assert np.array_equal(o.bscore_stack(), g["bscore_stack"])
    assert np.array_equal(o.bp_table_idx(nfr), g["bp_table_idx"])


def _replay(o, g, check_lists=True):
    off, act, scr = g["step_act_off"], g["step_act"], g["step_scr"]
    for i in range(int(g["n_steps"][0])):
        fr = int(g["step_frame"][i])
        a0, a1 = int(off[i]), int(off[i + 1])
        if check_lists and not np.array_equal(o.active_list(fr), act[a0:a1]):
            return "frame %d: active senone list" % fr
        o.step(fr, act[a0:a1], scr[a0:a1], int(g["step_rest"][i]), g["step_pen"][i])
    nfr = int(g["n_frame"][0])
    o.finish(nfr)
    if not (np.array_equal(o.bp_table(), g["bp"]) and np.array_equal(o.bscore_stack(), g["bscore_stack"])):
        return "tables"
    return None


@pytest.mark.parametrize("case,first", [("goforward_after_numbers", "numbers"), ("numbers_after_something", None)])
def test_fwdtree_oracle_second_utterance_of_a_session(case, first):
    """The reference's decoder decoded another utterance first (oracle/make_golden.py session): its multiplexed permanent
    channels start with the per-state ssids that utterance left (hmm_clear, hmm.c:181-196, keeps them; `mpx_init` in the
    golden), which changes the senones listed per frame.  A new oracle object given those ssids reproduces the trace; a new
    one without them does not; one that decoded the first utterance itself carries them without being told."""
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    o = pso.OracleFwdtree(st, g["par"])
    o.start()
    o.set_mpx_ssids(g["mpx_init"])
    assert _replay(o, g) is None
    o2 = pso.OracleFwdtree(st, g["par"])
    o2.start()
    assert _replay(o2, g) is not None          # a fresh decoder lists other senones
    if first:
        g1 = _load("fwdtree_trace_%s.npz" % first)
        o3 = pso.OracleFwdtree(st, g["par"])
        o3.start()
        assert _replay(o3, g1) is None
        mpx = np.asarray(st["w1_mpx"]) != 0
        R = int(g["par"][4])
        got, want = o3.get_mpx_ssids(g["mpx_init"].shape), g["mpx_init"]
        assert np.array_equal(got[:R], want[:R]) and np.array_equal(got[R:][mpx], want[R:][mpx])
        o3.start()
        assert _replay(o3, g) is None


def make_big_trace(out_dir):
    """`ref_dump fwdtree` of the compiled reference on the large-vocabulary task (too large to commit, ~10 s to make);
    None when oracle/_ref is not built"""
    import subprocess
    import sys
    ref = pso.REF_DIR
    need = [os.path.join(ref, "ref_dump"), os.path.join(ref, "data", "big.arpa"), os.path.join(ref, "data", "cmudict-en-us.dict")]
    if not all(os.path.exists(p) for p in need):
        return None
    out = os.path.join(str(out_dir), "big.psgb")
    subprocess.check_call([need[0], "fwdtree", out, os.path.join(ref, "model", "en-us"), need[1], need[2],
                           os.path.join(ref, "data", "goforward.raw"), "--", "fwdflat", "no", "bestpath", "no"], timeout=600)
    sys.path.insert(0, os.path.join(os.path.dirname(pso.__file__), "..", "oracle"))
    from psgb import read_psgb
    return read_psgb(out)


@pytest.fixture(scope="module")
def big_trace(tmp_path_factory):
    g = make_big_trace(tmp_path_factory.mktemp("big"))
    if g is None:
        pytest.skip("oracle/_ref (compiled reference + staged data) not built")
    return g


@pytest.mark.parametrize("parallel", [0, 1, 2])
def test_fwdtree_oracle_large_vocabulary(parallel, big_trace):
    """The search oracle at full scale: every base word of cmudict (134,865 dictionary entries, a lexicon tree
    of 248 k channels, ~9,000 active HMMs per frame) with the synthetic large LM of SURVEY F9b (oracle/make_biglm.py).
    The fixture is produced at test time by the compiled reference, which travels with the repository as
    oracle/_ref; language scores come from the trie oracle.  parallel = 2 is the formulation whose work per frame
    is proportional to the active channels, not the tree."""
    import ctypes as C
    g = big_trace
    assert bytes(g["hyp"]).decode() == "go forward ten meters" and int(g["par"][3]) > 100000
    o = pso.OracleFwdtree(g, g["par"], lm=pso.OracleLm(g))
    pso.lib().pso_ft_set_parallel.argtypes = [C.c_void_p, C.c_int]
    pso.lib().pso_ft_set_parallel(o.h, parallel)
    o.start()
    off, act, scr = g["step_act_off"], g["step_act"], g["step_scr"]
    for i in range(int(g["n_steps"][0])):
        fr = int(g["step_frame"][i])
        a0, a1 = int(off[i]), int(off[i + 1])
        assert np.array_equal(o.active_list(fr), act[a0:a1]), "frame %d: active senone list" % fr
        o.step(fr, act[a0:a1], scr[a0:a1], int(g["step_rest"][i]), g["step_pen"][i])
        assert (o.best_score(), o.last_phone_best_score(), o.bpidx()) == \
            (int(g["step_best"][i]), int(g["step_lpbest"][i]), int(g["step_bpidx"][i])), "frame %d" % fr
    nfr = int(g["n_frame"][0])
    o.finish(nfr)
    assert np.array_equal(o.bp_table(), g["bp"])
    assert np.array_equal(o.bscore_stack(), g["bscore_stack"])
    assert np.array_equal(o.bp_table_idx(nfr), g["bp_table_idx"])
