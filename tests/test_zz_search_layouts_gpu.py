"""GPU parity of the lexicon-tree search kernel in BOTH memory layouts (include/psgpu.h, psgpu_fwdtree_layout): the
tree-level state in LDS (what a small tree gets by default; tests/test_search_gpu.py, tests/test_lm_gpu.py run that) and
everything in the utterance's slab in device memory (what a large tree gets; forced here for the small ones with
PSGPU_FWDTREE_LAYOUT=slab), 256 and 1024 work-items per utterance, dense-table and trie language scores -- against
the same reference dumps: back-pointer table, score stack, frame marks and per-frame best scores bit for bit.  The
kernel source is also run on the CPU by tests/test_search_hostsim.py; every test body here runs in a child process
(conftest.run_isolated) so that a fault cannot take the other results with it."""
import os

import numpy as np
import pytest

from conftest import run_isolated
from test_oracle_golden import _load
from test_oracle_lm import load as lm_load
from test_oracle_search import CASES, MEDIUM_CASES, make_big_trace
from test_search_gpu import _check, _inputs

pytestmark = pytest.mark.gpu
ME = "test_zz_search_layouts_gpu"


@pytest.mark.parametrize("layout", ["slab", "lds"])
@pytest.mark.parametrize("case", CASES + MEDIUM_CASES)
def test_fwdtree_kernel_layouts_match_reference(case, layout):
    run_isolated(ME, "impl_matches_reference", case, layout)


def impl_matches_reference(case, layout):
    os.environ["PSGPU_FWDTREE_LAYOUT"] = layout
    import pocketsphinx_amd as P
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    lm = P.NGramTrieLM(st) if "lm" not in st else None
    s = P.FwdtreeSearch(st, g["par"], lm=lm)
    if layout == "slab":
        assert not s.lds_layout()
    elif case in CASES:
        assert s.lds_layout(), "the bundled small-vocabulary tasks are expected to fit the LDS pool"
    rows, pen = _inputs(g, s.n_sen)
    _check(s.search(rows, pen, [rows.shape[0]])[0], g, case)
    # ... and in three calls (psgpu_fwdtree_search_resume: the LDS layout's pool saved and restored, the slab layouts' state in place)
    T = rows.shape[0]
    _check(s.search(rows, pen, [T], cuts=[T // 3, T // 3 + 1, 2 * T // 3], lag=3)[0], g, case + ", resumed")
    assert s.searched == [T // 3 - 3, T // 3 - 2, 2 * T // 3 - 3, T]
    s.close()


@pytest.mark.parametrize("layout", ["slab", "lds"])
def test_fwdtree_kernel_batch_and_trie(layout):
    """several utterances per launch, language scores from the device trie"""
    run_isolated(ME, "impl_batch_and_trie", layout)


def impl_batch_and_trie(layout):
    os.environ["PSGPU_FWDTREE_LAYOUT"] = layout
    import pocketsphinx_amd as P
    names = ["goforward", "numbers", "goforward"]
    gs = [_load("fwdtree_trace_%s.npz" % n) for n in names]
    st = _load("fwdtree_static_en_us_turtle.npz")
    s = P.FwdtreeSearch(st, gs[0]["par"], lm=P.NGramTrieLM(lm_load("turtle_decoder")))
    ins = [_inputs(g, s.n_sen) for g in gs]
    out = s.search(np.concatenate([i[0] for i in ins]), np.concatenate([i[1] for i in ins]), [i[0].shape[0] for i in ins])
    for r, g, n in zip(out, gs, names):
        _check(r, g, n)
    # the on-device backtrace against the host walk of the same tables
    import torch
    h = {}
    out = s.search(ins[0][0], ins[0][1], [ins[0][0].shape[0]], handover=h)
    hyp, hn = s.backtrace_dev(h["bp"], h["idx"], h["result"], ins[0][0].shape[0])
    score, words = P.backtrace(out[0], s.finish_wid)
    assert int(hn[0, 0]) == len(words) and int(hn[0, 1]) == score
    assert [tuple(int(v) for v in hyp[0, i, :3]) for i in range(len(words))] == words
    torch.cuda.synchronize()
    s.close()


def test_fwdtree_kernel_full_cmudict_vocabulary(tmp_path):
    """134,865 words, 248 k tree channels, ~8 k active channels per frame: 1024 work-items per utterance"""
    import pso
    if not os.path.exists(os.path.join(pso.REF_DIR, "ref_dump")):
        pytest.fail("oracle/_ref (compiled reference + staged data) not built: run __graft_entry__.build() where /root/reference is present")
    run_isolated(ME, "impl_full_cmudict", str(tmp_path), timeout=1200)


def impl_full_cmudict(tmp):
    import pocketsphinx_amd as P
    g = make_big_trace(tmp)
    s = P.FwdtreeSearch(g, g["par"], lm=P.NGramTrieLM(g))
    assert not s.lds_layout()
    rows, pen = _inputs(g, s.n_sen)
    _check(s.search(rows, pen, [rows.shape[0]])[0], g, "cmudict")
    _check(s.search(np.concatenate([rows, rows]), np.concatenate([pen, pen]), [rows.shape[0]] * 2)[1], g, "cmudict x2")
    T = rows.shape[0]                                     # (1024 work-items an utterance, resumed twice)
    _check(s.search(rows, pen, [T], cuts=[T // 2, T // 2 + 9], lag=4)[0], g, "cmudict, resumed")
    assert s.searched == [T // 2 - 4, T // 2 + 5, T]
    s.close()


@pytest.mark.parametrize("layout", ["slab", "lds"])
def test_fwdtree_kernel_renormalises_as_the_oracle_does(layout):
    """renormalize_scores (ngram_search_fwdtree.c:566-603): unreachable with beams a configuration can name before an hour of audio, so
    held against the oracle with a beam that wide (tests/test_search_hostsim.py has the same case on the simulator)"""
    run_isolated(ME, "impl_renormalises", layout)


def impl_renormalises(layout):
    os.environ["PSGPU_FWDTREE_LAYOUT"] = layout
    import pocketsphinx_amd as P
    import pso
    g = _load("fwdtree_trace_goforward.npz")
    st = _load("fwdtree_static_en_us_turtle.npz")
    par = g["par"].copy()
    par[8] = -268434956                                   # beam: best + 2 beam < WORST_SCORE (-2^29) once best < -1,000
    o = pso.OracleFwdtree(st, par)
    o.start()
    off, act, scr = g["step_act_off"], g["step_act"], g["step_scr"]
    n = int(g["n_steps"][0])
    best = []
    for i in range(n):
        a0, a1 = int(off[i]), int(off[i + 1])
        o.step(int(g["step_frame"][i]), act[a0:a1], scr[a0:a1], 2000, g["step_pen"][i])
        best.append((o.best_score(), o.last_phone_best_score(), o.bpidx()))
    nfr = int(g["n_frame"][0])
    o.finish(nfr)
    assert sum(1 for i in range(1, n) if best[i][0] - best[i - 1][0] > 500) >= 3
    s = P.FwdtreeSearch(st, par)
    rows, pen = _inputs(g, s.n_sen)
    for i in range(n):
        listed = np.zeros(s.n_sen, bool); listed[act[int(off[i]):int(off[i + 1])]] = True
        rows[i, ~listed] = 2000
    r = s.search(rows, pen, [rows.shape[0]], bp_cap=1 << 16, bss_cap=1 << 21)[0]
    s.close()
    assert r["status"] == 0
    assert [tuple(int(v) for v in row[:3]) for row in r["step"][:n]] == best
    assert np.array_equal(r["bp"], o.bp_table()) and np.array_equal(r["bscore_stack"], o.bscore_stack())
    assert np.array_equal(r["bp_table_idx"], o.bp_table_idx(nfr))
