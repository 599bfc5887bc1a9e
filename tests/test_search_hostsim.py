"""CPU check of the lexicon-tree search KERNEL SOURCE (pocketsphinx_amd/csrc/psgpu_search.hip + the device
trie of psgpu_lm_dev.h) without a GPU: the unmodified sources are compiled with g++ against the workgroup
simulator of tests/hostsim (one fiber per work-item, cooperative switches at barriers and cross-lane
operations) and must reproduce the reference's back-pointer tables on the same goldens as the GPU parity
tests (tests/test_search_gpu.py, tests/test_lm_gpu.py).  The order in which the work-items of a workgroup
run between barriers is varied: a result that depends on it is a missing barrier.  This checks logic only --
the GPU tests remain the parity tests proper."""
import os

import numpy as np
import pytest

import simlib
from test_oracle_golden import _load
from test_oracle_lm import CASES as LM_CASES, load as lm_load
from test_oracle_search import CASES, MEDIUM_CASES, big_trace  # noqa: F401
from test_search_gpu import _check, _inputs


class _env:
    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = os.environ.get(self.name)
        os.environ[self.name] = self.value

    def __exit__(self, *a):
        if self.old is None:
            del os.environ[self.name]
        else:
            os.environ[self.name] = self.old


def _order(order):
    return _env("PSGPU_SIM_ORDER", order)


def _layout(layout):
    """"lds": the tree-level state in the workgroup's LDS pool where it fits (structure of arrays, copies of the static
    tree tables, score rows copied in a frame ahead); "slab": everything in the utterance's slab (psgpu_fwdtree_layout)."""
    return _env("PSGPU_FWDTREE_LAYOUT", layout)


def _run(case, order, trie=False, layout="lds"):
    g = _load("fwdtree_trace_%s.npz" % case)
    static = bytes(g["static"]).decode()
    st = _load("fwdtree_static_%s.npz" % static)
    lm = None
    if "lm" not in st:
        lm = simlib.SimLm(st)
    elif trie:
        lm = simlib.SimLm(lm_load({"en_us_turtle": "turtle_decoder", "tidigits": "tidigits_decoder"}[static]))
    with _order(order), _layout(layout):
        s = simlib.SimFwdtreeSearch(st, g["par"], lm=lm)
        rows, pen = _inputs(g, s.n_sen)
        _check(s.search(rows, pen, [rows.shape[0]])[0], g, "%s (%s)" % (case, order))
        s.close()


@pytest.mark.parametrize("layout,order", [("slab", "fwd"), ("slab", "rev"), ("lds", "fwd"), ("lds", "rev"), ("lds", "shuffle:7")])
@pytest.mark.parametrize("case", CASES + MEDIUM_CASES)
def test_search_kernel_source_on_the_simulator(case, layout, order):
    _run(case, order, layout=layout)


@pytest.mark.parametrize("layout", ["slab", "lds"])
@pytest.mark.parametrize("case", ["goforward", "man_ah_2934za"])
def test_search_kernel_source_with_the_trie_lm(case, layout):
    _run(case, "rev", trie=True, layout=layout)


@pytest.mark.parametrize("cap", ["0", "1", "3"])
@pytest.mark.parametrize("case", ["goforward", "numbers", "man_ah_2934za"])
def test_search_kernel_source_word_transitions_from_the_table(case, cap):
    """The LDS layout keeps a frame's exits in LDS for the word transitions (FtLay::xfr) and falls back to the back-pointer table
    itself on frames with more exits than that holds.  PSGPU_FWDTREE_XFR_CAP caps it at 0 / 1 / 3 entries: every frame, most
    frames, some frames take the table's path -- the tables must not change."""
    with _env("PSGPU_FWDTREE_XFR_CAP", cap):
        _run(case, "rev", layout="lds")


def test_search_kernel_source_batch_of_utterances():
    """several workgroups in one launch: every utterance equals its own golden"""
    names = ["goforward", "numbers", "goforward"]
    gs = [_load("fwdtree_trace_%s.npz" % n) for n in names]
    st = _load("fwdtree_static_en_us_turtle.npz")
    s = simlib.SimFwdtreeSearch(st, gs[0]["par"])
    ins = [_inputs(g, s.n_sen) for g in gs]
    out = s.search(np.concatenate([i[0] for i in ins]), np.concatenate([i[1] for i in ins]), [i[0].shape[0] for i in ins])
    for r, g, n in zip(out, gs, names):
        _check(r, g, n)
    s.close()


@pytest.mark.parametrize("order", ["fwd", "rev"])
def test_search_kernel_source_full_cmudict_vocabulary(big_trace, order):  # noqa: F811
    """The large-vocabulary form of the kernel (1024 work-items, everything in the utterance's slab) on the full cmudict task: 134,865 words, 248 k tree channels, ~8 k (up to 33 k) active
    channels per frame, language scores from the simulated device trie.  Tables identical to the reference's."""
    g = big_trace
    lm = simlib.SimLm(g)
    with _order(order):
        s = simlib.SimFwdtreeSearch(g, g["par"], lm=lm)
        rows, pen = _inputs(g, s.n_sen)
        _check(s.search(rows, pen, [rows.shape[0]])[0], g, "cmudict (%s)" % order)
        s.close()
    lm.close()


@pytest.mark.parametrize("cap", ["0"])
def test_search_kernel_source_rank_table_in_the_slab(big_trace, cap):  # noqa: F811
    """slab layouts: the listed nodes' index (bitmap -> rank -> position).  The rank -> position table is LDS while the frame's list fits
    what the pool has left and lies in the utterance's slab otherwise: PSGPU_FWDTREE_PERM_CAP cuts the LDS table down so that every frame
    takes the slab's -- on the full cmudict task and on small ones."""
    with _env("PSGPU_FWDTREE_PERM_CAP", cap):
        g = big_trace
        lm = simlib.SimLm(g)
        s = simlib.SimFwdtreeSearch(g, g["par"], lm=lm)
        rows, pen = _inputs(g, s.n_sen)
        _check(s.search(rows, pen, [rows.shape[0]])[0], g, "cmudict (perm cap %s)" % cap)
        s.close(); lm.close()
        _run("goforward", "rev", layout="slab")
        _run("man_ah_2934za", "fwd", layout="slab")             # (5-state HMMs)


@pytest.mark.parametrize("case,knob,status,start", [("goforward", "PSGPU_FWDTREE_LISTED_CAP", 4, "16"), ("goforward", "PSGPU_FWDTREE_RC_BLOCKS", 5, "1"),
                                               ("man_ah_2934za", "PSGPU_FWDTREE_RC_BLOCKS", 5, "1"), ("cmudict", "PSGPU_FWDTREE_RC_BLOCKS", 5, "256"),
                                               ("cmudict", "PSGPU_FWDTREE_LISTED_CAP", 4, "4096"), ("goforward", "PSGPU_FWDTREE_WL_CAP", 6, "16")])
def test_search_kernel_source_capacities_grow_on_demand(case, knob, status, start, big_trace):  # noqa: F811
    """slab layouts: the compact channels' capacity (tree nodes a frame may list) and the pool of the right-context channels' blocks start
    small and grow on demand -- a frame that needs more ends the utterance with status 4 / 5, psgpu_fwdtree_grow doubles the capacity and the
    search is repeated (status 6: the word level's scratch arrays leave LDS for the slab) (what psgpu_decode_fetch_hyps does by itself): the final tables are the reference's, and the first run did report the
    status (the knobs make the capacities that small)."""
    if case == "cmudict":
        g = big_trace; st = g; lm = simlib.SimLm(g)
    else:
        g = _load("fwdtree_trace_%s.npz" % case)
        st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
        lm = None if "lm" in st else simlib.SimLm(st)
    with _layout("slab"), _env(knob, start):
        s = simlib.SimFwdtreeSearch(st, g["par"], lm=lm)
    rows, pen = _inputs(g, s.n_sen)
    seen = []
    for _ in range(20):
        r = s.search(rows, pen, [rows.shape[0]])[0]
        seen.append(r["status"])
        if r["status"] not in (4, 5, 6):
            break
        simlib.check(simlib.lib().psgpu_fwdtree_grow(s.h, r["status"]), "psgpu_fwdtree_grow")
    assert seen[0] == status and seen[-1] == 0 and set(seen[:-1]) == {status}, seen
    _check(r, g, "%s after %d growths" % (case, len(seen) - 1))
    s.close()
    if lm is not None:
        lm.close()


@pytest.mark.parametrize("layout", ["slab", "lds"])
def test_search_kernel_source_renormalises_as_the_oracle_does(layout):
    """renormalize_scores (ngram_search_fwdtree.c:566-603, 1473-1480) runs when best_score + 2 beam falls below WORST_SCORE -- with the
    beams a configuration can name that is an hour into an utterance, so no reference dump holds it.  Here the beam is made wide enough
    (a parameter no configuration reaches) that it happens every few dozen frames of goforward: the kernel's tables against the oracle's
    (oracle/ps_oracle_search.c, whose renormalisation restates the reference's lines; the oracle is pinned to the reference on every dump
    without one).  The slab layouts normalise a listed node's channel where the frame's evaluation makes it."""
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
        o.step(int(g["step_frame"][i]), act[a0:a1], scr[a0:a1], 2000, g["step_pen"][i])      # (senones the dump did not list: a plain high cost)
        best.append((o.best_score(), o.last_phone_best_score(), o.bpidx()))
    nfr = int(g["n_frame"][0])
    o.finish(nfr)
    jumps = sum(1 for i in range(1, n) if best[i][0] - best[i - 1][0] > 500)
    assert jumps >= 3, jumps                                # (a renormalised frame's best score rises by what was taken out)
    with _layout(layout):
        s = simlib.SimFwdtreeSearch(st, par)
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


@pytest.mark.parametrize("name", LM_CASES)
def test_simulated_device_trie_equals_reference_look_ups(name):
    """psgpu_lm_dev.h through the simulator against the reference's recorded look-ups"""
    g = lm_load(name)
    lm = simlib.SimLm(g)
    q = g["queries"][:20000]
    sc, nu = lm.tg_score(q)
    assert np.array_equal(sc, g["scores"][:q.shape[0]]) and np.array_equal(nu, g["n_used"][:q.shape[0]])
    lm.close()


def test_search_kernel_source_scoring_its_own_senones(tables):
    """psgpu_fwdtree_search_lists_dev: no score rows -- the kernel gets the scorer's top-N lists (the reference's, golden
    ptm_goforward with full_topn) and evaluates ptm_mgau_codebook_norm + ptm_mgau_senone_eval for the senones it lists,
    as the reference's first pass asks its scorer to.  The tables it produces are the reference decoder's."""
    g = _load("fwdtree_trace_goforward.npz")
    gp = _load("ptm_goforward.npz")
    st = _load("fwdtree_static_en_us_turtle.npz")
    with _layout("lds"):
        s = simlib.SimFwdtreeSearch(st, g["par"])
    rows, pen = _inputs(g, s.n_sen)
    T = rows.shape[0]
    assert gp["topn_cw"].shape[0] >= T
    r = simlib.search_lists(s, tables, gp["topn_raw"][:T], gp["topn_cw"][:T], pen, [T])[0]
    _check(r, g, "lists")
    n_act = int(g["step_act_off"][-1])                                     # the reference's lists, bridging entries included
    assert 0.9 * n_act < r["listed"] <= n_act
    s.close()


def _raw_rows(g, rows, seed=5):
    """the reference's normalised scores plus an arbitrary per-frame offset in the listed senones, garbage elsewhere"""
    rng = np.random.default_rng(seed)
    off, act = g["step_act_off"], g["step_act"]
    raw = rng.integers(-30000, 30000, rows.shape).astype(np.int16)
    for i in range(rows.shape[0]):
        a = act[off[i]:off[i + 1]]
        raw[i, a] = (rows[i, a].astype(np.int32) + int(rng.integers(-2000, 2000))).astype(np.int16)
    return raw


@pytest.mark.parametrize("layout", ["slab", "lds"])
def test_search_kernel_source_second_utterance_of_a_session(layout):
    """psgpu_fwdtree_search_session_dev: the reference's decoder had decoded numbers.raw before goforward.raw (golden
    goforward_after_numbers; oracle/make_golden.py session).  In raw-score mode the kernel lists the senones itself, so it
    must start from the per-state ssids the first utterance left in the multiplexed permanent channels: given the golden's
    `mpx_init` it reproduces the trace, without it it does not; and the ssids the kernel itself ends numbers.raw with are
    those the reference's decoder had."""
    g = _load("fwdtree_trace_goforward_after_numbers.npz")
    g1 = _load("fwdtree_trace_numbers.npz")
    st = _load("fwdtree_static_en_us_turtle.npz")
    with _layout(layout):
        s = simlib.SimFwdtreeSearch(st, g["par"])
    rows, pen = _inputs(g, s.n_sen)
    raw = _raw_rows(g, rows)
    _check(s.search(raw, pen, [rows.shape[0]], raw_scores=True, pl_window=0, mpx_in=g["mpx_init"][None])[0], g, "session")
    fresh = s.search(raw, pen, [rows.shape[0]], raw_scores=True, pl_window=0)[0]
    assert fresh["bp"].shape != g["bp"].shape or not np.array_equal(fresh["bp"], g["bp"])
    rows1, pen1 = _inputs(g1, s.n_sen)
    out = {}
    _check(s.search(_raw_rows(g1, rows1), pen1, [rows1.shape[0]], raw_scores=True, pl_window=0, mpx_out=out)[0], g1, "first")
    R = int(g["par"][4]); mpx = np.asarray(st["w1_mpx"]) != 0
    assert np.array_equal(out["mpx"][0][:R], g["mpx_init"][:R]) and np.array_equal(out["mpx"][0][R:][mpx], g["mpx_init"][R:][mpx])
    # two utterances of one call are independent: each starts from its own input state
    both = s.search(np.concatenate([raw, raw]), np.concatenate([pen, pen]), [rows.shape[0]] * 2, raw_scores=True, pl_window=0,
                    mpx_in=np.stack([g["mpx_init"], out["mpx"][0]]))
    _check(both[0], g, "batch 0"); _check(both[1], g, "batch 1")
    s.close()


@pytest.mark.parametrize("layout", ["slab", "lds"])
@pytest.mark.parametrize("case", ["goforward", "numbers", "medium_goforward"])
def test_search_kernel_source_building_its_own_active_lists(case, layout):
    """raw-score mode (the kernel builds each frame's active senone list, bridging entries included, and subtracts the
    list's minimum itself, as the PTM scorer does): fed the reference's normalised scores plus an arbitrary per-frame
    offset -- the listed senones' minimum is 0 in the PTM scorer's rows, so the kernel must recover them exactly --
    and garbage in every senone the search does not ask for.  (PTM traces only: the semi-continuous scorer of the
    tidigits trace does not normalise by the minimum.)"""
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    lm = simlib.SimLm(st) if "lm" not in st else None
    with _layout(layout):
        s = simlib.SimFwdtreeSearch(st, g["par"], lm=lm)
    rows, pen = _inputs(g, s.n_sen)
    rng = np.random.default_rng(5)
    off, act = g["step_act_off"], g["step_act"]
    raw = rng.integers(-30000, 30000, rows.shape).astype(np.int16)          # what nobody should read
    for i in range(rows.shape[0]):
        a = act[off[i]:off[i + 1]]
        raw[i, a] = (rows[i, a].astype(np.int32) + int(rng.integers(-2000, 2000))).astype(np.int16)
    with _order("rev"):
        _check(s.search(raw, pen, [rows.shape[0]], raw_scores=True, pl_window=0)[0], g, case)
    s.close()


def test_search_kernel_source_full_cmudict_own_active_lists(big_trace):  # noqa: F811
    """the same at full scale: 1024 work-items, the tree search building its own active lists"""
    g = big_trace
    lm = simlib.SimLm(g)
    s = simlib.SimFwdtreeSearch(g, g["par"], lm=lm)
    rows, pen = _inputs(g, s.n_sen)
    off, act = g["step_act_off"], g["step_act"]
    raw = np.full(rows.shape, 12345, np.int16)
    for i in range(rows.shape[0]):
        a = act[off[i]:off[i + 1]]
        raw[i, a] = (rows[i, a].astype(np.int32) + 777 - 5 * (i % 50)).astype(np.int16)
    _check(s.search(raw, pen, [rows.shape[0]], raw_scores=True, pl_window=0)[0], g, "cmudict raw")
    s.close(); lm.close()


@pytest.mark.parametrize("layout", ["slab", "lds"])
@pytest.mark.parametrize("lag", [1, 7, 10 ** 6])
def test_search_kernel_source_stopping_short_of_the_last_frames(layout, lag):
    """psgpu_fwdtree_search_lag: an utterance in progress (ps_search_forward, pocketsphinx.c:1173-1197 -- the n-gram search runs
    behind the frames scored so far).  The tables are append-only and the golden trace records where each frame's entries
    begin, so the search stopped `lag` frames early must hold exactly the golden's first bp_table_idx[T - lag] back-pointers and
    the score-stack entries those own; a lag past the utterance's length searches nothing; and the request is one call's worth."""
    g = _load("fwdtree_trace_goforward.npz")
    st = _load("fwdtree_static_en_us_turtle.npz")
    with _layout(layout):
        s = simlib.SimFwdtreeSearch(st, g["par"])
    rows, pen = _inputs(g, s.n_sen)
    T = rows.shape[0]
    L = simlib.lib()
    L.psgpu_fwdtree_search_lag.argtypes = [simlib.C.c_void_p, simlib.C.c_int32]
    simlib.check(L.psgpu_fwdtree_search_lag(s.h, lag), "psgpu_fwdtree_search_lag")
    r = s.search(rows, pen, [T])[0]
    n = max(T - lag, 0)
    assert r["status"] == 0 and r["n_frame"] == n
    nbp = int(g["bp_table_idx"][n]) if n else 0
    assert r["bp"].shape[0] == nbp and np.array_equal(r["bp"], g["bp"][:nbp])
    assert np.array_equal(r["bp_table_idx"], g["bp_table_idx"][:n + 1])
    later = g["bp"][nbp:, 5]                              # (column 5: s_idx, the first stack entry a back-pointer owns; -1 for
    later = later[later >= 0]                             # single-phone words, which own none: ngram_search.c:476-480)
    nbss = int(later[0]) if later.size else g["bscore_stack"].size
    assert np.array_equal(r["bscore_stack"], g["bscore_stack"][:nbss if nbp else 0])
    assert np.array_equal(r["step"][:n, :3], np.stack([g["step_best"], g["step_lpbest"], g["step_bpidx"]], axis=1)[:n])
    _check(s.search(rows, pen, [T])[0], g, "the call after")
    s.close()


@pytest.mark.parametrize("layout,order", [("lds", "fwd"), ("lds", "rev"), ("slab", "rev")])
@pytest.mark.parametrize("case,cuts,lag", [("goforward", [1, 2, 40, 41, 150], 0), ("goforward", [30, 100, 200], 7),
                                           ("numbers", [97], 3), ("man_ah_2934za", [10, 11, 60], 2),
                                           ("goforward_maxhmmpf60_maxwpf3", [50, 51, 170], 5), ("medium_numbers_maxwpf8", [33, 120], 4)])
def test_search_kernel_source_resumed_between_calls(case, cuts, lag, layout, order):
    """psgpu_fwdtree_search_resume: one utterance searched in several calls, each going on where the one before stopped (the LDS
    pool, the counters and the frame loop's carried registers saved and restored -- the slab layouts' state stays in the slab; the
    reference's search keeps its state between
    ps_search_forward rounds, ngram_search_fwdtree.c:1454-1495).  Every frame is searched once -- the frames searched after each
    call are the cut minus the lag -- and the tables at the end are the golden's, as from one call."""
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    lm = simlib.SimLm(st) if "lm" not in st else None       # (the medium task: language scores from the device trie)
    with _order(order), _layout(layout):
        s = simlib.SimFwdtreeSearch(st, g["par"], lm=lm)
        rows, pen = _inputs(g, s.n_sen)
        T = rows.shape[0]
        _check(s.search(rows, pen, [T], cuts=cuts, lag=lag)[0], g, "%s resumed at %r" % (case, cuts))
        assert s.searched == [max(c - lag, 0) for c in cuts] + [T]
        # the handle's next plain call starts afresh; a resume without a kept state is refused
        _check(s.search(rows, pen, [T])[0], g, "the call after")
        L = simlib.lib()
        simlib.check(L.psgpu_fwdtree_search_resume(s.h, 2), "psgpu_fwdtree_search_resume")
        with pytest.raises(RuntimeError, match="nothing to resume"):
            s.search(rows, pen, [T])
        s.close()


@pytest.mark.parametrize("layout", ["lds", "slab"])
@pytest.mark.parametrize("case,cuts,lag", [("goforward", [9, 10, 47, 120, 121, 200], 5), ("numbers", [60, 61, 150], 8)])
def test_search_kernel_source_on_a_window_of_rows(case, cuts, lag, layout):
    """psgpu_fwdtree_search_streams: a live stream keeps only the score rows and penalties its search has not reached yet; every call gets
    that window, placed by a start before the buffer, and per-utterance {frames scored, frame to search to}.  The tables at the end
    are the golden's."""
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    with _order("rev"), _layout(layout):
        s = simlib.SimFwdtreeSearch(st, g["par"])
        rows, pen = _inputs(g, s.n_sen)
        T = rows.shape[0]
        r, searched = simlib.search_windows(s, rows, pen, cuts + [T], lag)
        _check(r, g, "%s in windows" % case)
        assert searched == [max(c - lag, 0) for c in cuts] + [T]
        s.close()


@pytest.mark.parametrize("layout", ["lds", "slab"])
def test_search_kernel_source_streams_with_a_restart(layout):
    """two utterances in progress on one handle, growing at different paces, each on its own window of rows; stream 0 ends early and is
    begun again with the other recording (psgpu_fwdtree_search_restart) while stream 1 goes on: every finished utterance's tables are
    its golden's"""
    ga, gb = _load("fwdtree_trace_goforward.npz"), _load("fwdtree_trace_numbers.npz")
    st = _load("fwdtree_static_en_us_turtle.npz")
    with _order("rev"), _layout(layout):
        s = simlib.SimFwdtreeSearch(st, ga["par"])
        a, b = _inputs(ga, s.n_sen), _inputs(gb, s.n_sen)
        Ta, Tb = a[0].shape[0], b[0].shape[0]
        lag = 6
        sched = [[(50, False, None), (20, False, None)],
                 [(51, False, None), (130, False, None)],
                 [(Ta, True, None), (131, False, None)]]
        out, log = simlib.search_streams(s, [a, b], sched, lag)
        _check(out[0], ga, "stream 0, first utterance")
        assert log == [[44, 14], [45, 124], [Ta, 125]]
        # stream 0 again, now with the other recording; stream 1 to its end
        sched2 = sched + [[(70, False, b), (200, False, None)], [(Tb, True, None), (Tb, True, None)]]
        out, log = simlib.search_streams(s, [a, b], sched2, lag)
        _check(out[0], gb, "stream 0, second utterance")
        _check(out[1], gb, "stream 1")
        assert log[3] == [64, 194] and log[4] == [Tb, Tb]
        s.close()


@pytest.mark.parametrize("layout", ["slab", "lds"])
def test_search_kernel_source_final_scores_mode(layout):
    """raw_scores = 3 (psgpu.h): the kernel builds each frame's active senone list but the rows are FINAL scores -- a scorer that
    does not normalise over that list (s2_semi_mgau_frame_eval, src/s2_semi_mgau.c:837-883: the tidigits trace).  The golden's
    rows go in unchanged where the search asks, garbage everywhere else; with raw_scores = 1 the same input must NOT give the
    trace (the listed senones' minimum is not 0 for this scorer), which is what separates the two modes."""
    g = _load("fwdtree_trace_man_ah_2934za.npz")
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    with _layout(layout):
        s = simlib.SimFwdtreeSearch(st, g["par"])
    rows, pen = _inputs(g, s.n_sen)
    off, act = g["step_act_off"], g["step_act"]
    raw = np.random.default_rng(11).integers(-30000, 30000, rows.shape).astype(np.int16)
    mins = []
    for i in range(rows.shape[0]):
        a = act[off[i]:off[i + 1]]
        raw[i, a] = rows[i, a]
        mins.append(int(rows[i, a].min()) if a.size else 0)
    assert any(m != 0 for m in mins)
    with _order("rev"):
        _check(s.search(raw, pen, [rows.shape[0]], raw_scores=3, pl_window=0)[0], g, "final scores")
    other = s.search(raw, pen, [rows.shape[0]], raw_scores=1, pl_window=0)[0]
    assert other["bp"].shape != g["bp"].shape or not np.array_equal(other["bp"], g["bp"])
    s.close()
