"""GPU parity of the lexicon-tree search kernel (psgpu_fwdtree_*, the ngram_fwdtree_search
replacement) against dumps of the unmodified reference: fed the senone
scores and phone-loop penalties the reference's search was handed frame by frame, the kernel must
produce the reference's back-pointer table (all ten columns), right-context score stack, per-frame
marks, and per-frame best scores -- bit for bit.  Same goldens as the CPU oracle
(tests/test_oracle_search.py): en-us + turtle LM on two recordings, forced histogram and
word-exit pruning, no phone-loop look-ahead, tidigits (5-state HMMs)."""
import numpy as np
import pytest

from test_oracle_golden import _load
from test_oracle_search import CASES

pytestmark = pytest.mark.gpu


def _inputs(g, n_sen):
    n = int(g["n_steps"][0])
    off, act, scr = g["step_act_off"], g["step_act"], g["step_scr"]
    rows = np.empty((n, n_sen), np.int16)
    for i in range(n):
        rows[i] = g["step_rest"][i]
        rows[i, act[off[i]:off[i + 1]]] = scr[off[i]:off[i + 1]]
    return rows, np.ascontiguousarray(g["step_pen"], np.int32)


def _check(r, g, what):
    n = int(g["n_steps"][0])
    assert r["status"] == 0, what
    st = r["step"]
    ref = np.stack([g["step_best"], g["step_lpbest"], g["step_bpidx"]], axis=1)
    m = min(st.shape[0], n)
    bad = np.nonzero((st[:m, :3] != ref[:m]).any(axis=1))[0]
    assert bad.size == 0, "%s: first diverging frame %d: kernel %r reference %r" % (what, bad[0], st[bad[0]], ref[bad[0]])
    assert r["n_frame"] == int(g["n_frame"][0]), what
    assert r["bp"].shape == g["bp"].shape, what
    badbp = np.nonzero((r["bp"] != g["bp"]).any(axis=1))[0]
    assert badbp.size == 0, "%s: back-pointer %d: %r vs %r" % (what, badbp[0], r["bp"][badbp[0]], g["bp"][badbp[0]])
    assert np.array_equal(r["bscore_stack"], g["bscore_stack"]), what
    assert np.array_equal(r["bp_table_idx"], g["bp_table_idx"]), what


@pytest.mark.parametrize("case", CASES)
def test_fwdtree_kernel_matches_reference(case):
    import pocketsphinx_amd as P
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    s = P.FwdtreeSearch(st, g["par"])
    rows, pen = _inputs(g, s.n_sen)
    _check(s.search(rows, pen, [rows.shape[0]])[0], g, case)
    s.close()


def test_fwdtree_kernel_batch_of_utterances():
    """Several utterances in one launch (one workgroup each): every one equals its own golden."""
    import pocketsphinx_amd as P
    names = ["goforward", "numbers", "goforward"]
    gs = [_load("fwdtree_trace_%s.npz" % n) for n in names]
    st = _load("fwdtree_static_en_us_turtle.npz")
    s = P.FwdtreeSearch(st, gs[0]["par"])
    ins = [_inputs(g, s.n_sen) for g in gs]
    out = s.search(np.concatenate([i[0] for i in ins]), np.concatenate([i[1] for i in ins]), [i[0].shape[0] for i in ins])
    for r, g, n in zip(out, gs, names):
        _check(r, g, n)
    s.close()
