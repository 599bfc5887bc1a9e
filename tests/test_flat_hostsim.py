"""CPU check of the flat-lexicon second-pass KERNEL SOURCE (pocketsphinx_amd/csrc/psgpu_flat.hip) on the
workgroup simulator of tests/hostsim: given what the reference's first pass handed over and the senone scores
its second pass was handed, the kernel must reproduce the second pass's back-pointer table, score stack, frame
marks and per-frame best score / back-pointer count (goldens of tests/test_oracle_flat.py), in any work-item
order.  Logic only -- tests/test_zz_flat_gpu.py is the parity test proper."""
import numpy as np
import pytest

import simlib
from test_oracle_flat import FLAT_CASES, load_flat
from test_search_hostsim import _order


def flat_rows(g, n_sen):
    n = int(g["flat_n_steps"][0])
    off, act, scr = g["flat_act_off"], g["flat_act"], g["flat_scr"]
    rows = np.empty((n, n_sen), np.int16)
    for i in range(n):
        rows[i] = g["flat_rest"][i]
        rows[i, act[off[i]:off[i + 1]]] = scr[off[i]:off[i + 1]]
    return rows


def check_flat(r, g, what):
    n = int(g["flat_n_steps"][0])
    assert r["status"] == 0 and r["n_frame"] == n, what
    ref = np.stack([g["flat_best"], g["flat_bpidx"]], axis=1)
    got = r["step"][:, [0, 2]]
    bad = np.nonzero((got != ref).any(axis=1))[0]
    assert bad.size == 0, "%s: first diverging frame %d: kernel %r reference %r" % (what, bad[0], got[bad[0]], ref[bad[0]])
    assert r["bp"].shape == g["bp"].shape, what
    badbp = np.nonzero((r["bp"] != g["bp"]).any(axis=1))[0]
    assert badbp.size == 0, "%s: back-pointer %d: %r vs %r" % (what, badbp[0], r["bp"][badbp[0]], g["bp"][badbp[0]])
    assert np.array_equal(r["bscore_stack"], g["bscore_stack"]), what
    assert np.array_equal(r["bp_table_idx"], g["bp_table_idx"]), what


@pytest.mark.parametrize("order", ["fwd", "rev", "shuffle:11"])
@pytest.mark.parametrize("case", FLAT_CASES)
def test_flat_kernel_source_on_the_simulator(case, order):
    g, st, fst = load_flat(case)
    lm = simlib.SimLm(fst) if "lm" not in st else None
    with _order(order):
        s = simlib.SimFwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"], lm=lm)
        r = s.search(flat_rows(g, s.n_sen), [int(g["flat_n_steps"][0])], [g["bp1"]], [g["flat_w1_ssid"]])[0]
        check_flat(r, g, "%s (%s)" % (case, order))
        s.close()


def test_flat_kernel_source_batch_of_utterances():
    """several workgroups per launch, each with its own vocabulary and chain layout"""
    names = ["goforward", "numbers"]
    loaded = [load_flat(n) for n in names]
    g0, st, fst = loaded[0]
    s = simlib.SimFwdflatSearch(st, fst, g0["par"], g0["flat_par"], g0["flat_lwf"])
    gs = [loaded[0][0], loaded[1][0], loaded[0][0]]
    rows = [flat_rows(g, s.n_sen) for g in gs]
    out = s.search(np.concatenate(rows), [r.shape[0] for r in rows], [g["bp1"] for g in gs], [g["flat_w1_ssid"] for g in gs])
    for r, g in zip(out, gs):
        check_flat(r, g, "batch")
    s.close()


@pytest.mark.parametrize("case", ["goforward", "numbers", "man_ah_2934za"])
def test_two_passes_chained_on_the_simulator(case):
    """First-pass kernel -> second-pass kernel with nothing but the kernels' own buffers in between: the tree search
    on the first pass's trace leaves its back-pointer table, result record and the single-phone channels' ssids
    (psgpu_fwdtree_set_w1_ssid_out); the flat search takes them over and must end with the reference's pass-2
    tables.  The hand-over itself is checked against what the reference's first pass left."""
    from test_oracle_golden import _load
    from test_search_gpu import _inputs
    g, st, fst = load_flat(case)
    g1 = _load("fwdtree_trace_%s.npz" % case)
    s1 = simlib.SimFwdtreeSearch(st, g1["par"])
    rows1, pen1 = _inputs(g1, s1.n_sen)
    h = {}
    r1 = s1.search(rows1, pen1, [rows1.shape[0]], handover=h)[0]
    assert np.array_equal(r1["bp"], g["bp1"]) and np.array_equal(h["w1_ssid"][0], g["flat_w1_ssid"])
    s2 = simlib.SimFwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"])
    r2 = s2.search(flat_rows(g, s2.n_sen), [int(g["flat_n_steps"][0])], h)[0]
    check_flat(r2, g, case)
    s1.close(); s2.close()


@pytest.mark.parametrize("order", ["fwd", "rev"])
@pytest.mark.parametrize("case", ["goforward", "numbers", "something_efwid2_sfwin8"])
def test_flat_kernel_source_scoring_its_own_senones(case, order):
    """psgpu_fwdflat_search_feats_dev: handed the FEATURE rows and the PTM tables instead of scores, the kernel restates
    ptm_mgau_frame_eval as the second pass calls it (own active list, only the touched codebooks scanned and
    normalised) and must still end with the reference's pass-2 tables."""
    import pso
    g, st, fst = load_flat(case)
    with _order(order):
        s = simlib.SimFwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"])
        r = s.search(g["flat_feat"], [g["flat_feat"].shape[0]], [g["bp1"]], [g["flat_w1_ssid"]], ptm_tables=pso.load_tables(),
                     topn_seed=g["flat_ptm_seed"])[0]
        check_flat(r, g, "%s (%s)" % (case, order))
        s.close()


@pytest.mark.parametrize("case", ["goforward", "numbers", "something_efwid2_sfwin8"])
def test_second_pass_seed_is_a_first_pass_list(case):
    """What psgpu_fwdflat_search_feats_dev documents about topn_seed_dev: the history slot pass-2 frame 0 starts from
    holds the lists of the last first-pass frame t with t % n_fast_hist == n_fast_hist - 1 (every codebook is
    touched in pass 1, so these are the batch scorer's lists); and the oracle of the per-call scorer, rewound and
    called with the second pass's active lists, gives the scores the reference's second pass was handed."""
    import pso
    t = pso.load_tables()
    g, st, fst = load_flat(case)
    H = int(t["n_fast_hist"][0])
    o = pso.OraclePTM(t)
    feats = g["flat_feat"]
    _, cw, _ = o.score_utt(feats, reset_hist=True)
    ts = max(x for x in range(feats.shape[0]) if x % H == H - 1)
    assert np.array_equal(cw[ts].astype(np.int32), g["flat_ptm_seed"])
    off, act, scr = g["flat_act_off"], g["flat_act"], g["flat_scr"]
    for i in range(0, feats.shape[0]):
        a = act[off[i]:off[i + 1]].astype(np.int64)
        o.set_frame_idx(i)
        sc = o.frame_eval(feats[i], i, active=np.diff(np.concatenate([[0], a])).astype(np.uint8), compallsen=False)
        assert np.array_equal(sc[a], scr[off[i]:off[i + 1]]), "frame %d" % i


@pytest.mark.parametrize("scoring", [False, True])
def test_flat_kernel_source_full_cmudict_vocabulary(big_flat_trace, scoring):
    """the second-pass kernel at full scale (134,865-word dictionary, trie language scores, per-word arrays sized for the
    dictionary, an utterance vocabulary of ~150 words in ~4,400 chain channels), with the scores given and scoring its
    own senones from the feature rows"""
    import pso
    g = big_flat_trace
    lm = simlib.SimLm(g)
    s = simlib.SimFwdflatSearch(g, g, g["par"], g["flat_par"], g["flat_lwf"], lm=lm)
    nfr = int(g["n_frame"][0])
    if scoring:
        r = s.search(g["flat_feat"], [nfr], [g["bp1"]], [g["flat_w1_ssid"]], ptm_tables=pso.load_tables(), topn_seed=g["flat_ptm_seed"])[0]
    else:
        r = s.search(flat_rows(g, s.n_sen), [nfr], [g["bp1"]], [g["flat_w1_ssid"]])[0]
    check_flat(r, g, "cmudict")
    s.close(); lm.close()


def test_two_passes_chained_full_cmudict_vocabulary(big_flat_trace):
    """both kernels at full scale, chained through their own buffers: the tree search (ACTIVE_LIST, 1024 work-items) on the
    first pass's trace of the same two-pass decode, then the flat search on what it left"""
    from test_search_gpu import _inputs
    g = big_flat_trace
    lm = simlib.SimLm(g)
    s1 = simlib.SimFwdtreeSearch(g, g["par"], lm=lm)
    rows1, pen1 = _inputs(g, s1.n_sen)
    h = {}
    r1 = s1.search(rows1, pen1, [rows1.shape[0]], handover=h)[0]
    assert np.array_equal(r1["bp"], g["bp1"]) and np.array_equal(h["w1_ssid"][0], g["flat_w1_ssid"])
    s2 = simlib.SimFwdflatSearch(g, g, g["par"], g["flat_par"], g["flat_lwf"], lm=lm)
    check_flat(s2.search(flat_rows(g, s2.n_sen), [int(g["flat_n_steps"][0])], h)[0], g, "cmudict chained")
    s1.close(); s2.close(); lm.close()


def _all_density_lists(t, feats):
    """what the batch scorer hands over (psgpu_ptm_score_batch_dev): per (codebook, stream) chain and frame the four best of ALL
    128 densities -- fp32 distances with the reference's one-rounding-per-operation order (ptm_mgau.c:102-128), truncated,
    descending, the lower codeword first among equals -- and the `open` flag where that is not certainly the reference's list
    (a tie among the best five, a score at the clamp)"""
    n_cb, n_feat, n_den = int(t["n_mgau"][0]), int(t["n_feat"][0]), int(t["n_density"][0])
    fl = [int(v) for v in t["featlen"]]; veclen = sum(fl); T = feats.shape[0]
    mean = np.asarray(t["mean"], np.float32).reshape(n_cb, veclen * n_den); var = np.asarray(t["var"], np.float32).reshape(n_cb, veclen * n_den)
    det = np.asarray(t["det"], np.float32)
    sc = np.zeros((n_cb * n_feat, T, 4), np.int32); cw = np.zeros((n_cb * n_feat, T, 4), np.uint8); op = np.zeros((n_cb * n_feat, T), np.uint8)
    x = np.asarray(feats, np.float32)
    for cb in range(n_cb):
        for fs in range(n_feat):
            o = n_den * sum(fl[:fs]); ln = fl[fs]
            m = mean[cb, o:o + n_den * ln].reshape(n_den, ln); v = var[cb, o:o + n_den * ln].reshape(n_den, ln)
            xs = x[:, sum(fl[:fs]):sum(fl[:fs]) + ln]
            d = np.broadcast_to(det[cb, fs][None, :], (T, n_den)).astype(np.float32)
            for j in range(ln):
                diff = (xs[:, j][:, None] - m[:, j][None, :]).astype(np.float32)
                d = (d - ((diff * diff).astype(np.float32) * v[:, j][None, :]).astype(np.float32)).astype(np.float32)
            s = np.clip(np.trunc(d.astype(np.float64)), -(1 << 24), (1 << 24) - 1).astype(np.int64)
            key = (s << 7) | (127 - np.arange(n_den))[None, :]
            best = np.sort(key, axis=1)[:, ::-1][:, :5]
            b = best >> 7
            ch = cb * n_feat + fs
            sc[ch] = b[:, :4]; cw[ch] = (127 - (best[:, :4] & 127)).astype(np.uint8)
            op[ch] = ((np.diff(b, axis=1) == 0).any(axis=1) | (b[:, 0] >= (1 << 24) - 1) | (b[:, 3] <= -(1 << 24))).astype(np.uint8)
    return sc, cw, op


@pytest.mark.parametrize("extra_open", [0.0, 0.02, 0.5])
@pytest.mark.parametrize("case", ["goforward", "something_efwid2_sfwin8"])
def test_flat_kernel_source_taking_the_batch_scorers_lists(case, extra_open):
    """psgpu_fwdflat_search_feats_lists_dev: a touched codebook's list is taken from the batch scorer where its entry is closed, an
    untouched codebook's chain is left alone, and an OPEN entry of a touched codebook is scanned with the reference's sequential
    procedure -- on the list the reference would carry at that point, which the kernel has to replay (the re-orderings of the
    untouched frames in between).  Marking MORE entries open than the ties require is always allowed (they are then scanned
    exactly), and makes that replay the common case instead of a once-an-utterance one: the tables must not change."""
    import pso
    t = pso.load_tables()
    g, st, fst = load_flat(case)
    feats = g["flat_feat"]
    sc, cw, op = _all_density_lists(t, feats)
    if extra_open:
        op = op | (np.random.default_rng(17).random(op.shape) < extra_open).astype(np.uint8)
    with _order("rev"):
        s = simlib.SimFwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"])
        r = s.search(feats, [feats.shape[0]], [g["bp1"]], [g["flat_w1_ssid"]], ptm_tables=t, topn_seed=g["flat_ptm_seed"],
                     lists=(sc, cw, op))[0]
        check_flat(r, g, "%s (extra open %s)" % (case, extra_open))
        s.close()


@pytest.mark.parametrize("knob", ["psgpu_sim_ff_exit_cap:1", "psgpu_sim_ff_el_cap:3", "psgpu_sim_ff_awl_regs:0", "psgpu_sim_ff_pair_rows:0",
                                  "psgpu_sim_ff_pair_rows:2", "psgpu_sim_ff_slice_chunk:3", "psgpu_sim_ff_force_walk:1"])
@pytest.mark.parametrize("case", ["goforward", "something_efwid2_sfwin8"])
def test_flat_kernel_source_more_than_the_lds_queues_hold(case, knob):
    """a frame's word exits (kFfMaxExit) and its active-channel list (kFfMaxEl) are held in LDS; what does not fit goes through
    the slab -- flags per channel for the exits of such a frame, the list's tail in the slab's arrays.  The simulator's build
    reads both capacities from variables: with room for ONE exit / THREE list entries nearly every frame takes the other
    paths (third knob: the next active word list by a scan through the slab, as vocabularies beyond 1024 words get it; fourth: the word
    transitions of a frame with more new entries than the LDS rows hold -- none / two -- go through the table, exits in order; fifth:
    the window's words three a chunk; sixth: every exiting word's entry by the walk over its exits instead of the exits' maximum), scoring its own senones from the batch scorer's lists or given the scores, and the tables must be the same."""
    import ctypes
    import pso
    g, st, fst = load_flat(case)
    name, val = knob.split(":")
    cap = ctypes.c_int.in_dll(simlib.lib(), name)
    old = cap.value
    try:
        cap.value = int(val)
        with _order("rev"):
            s = simlib.SimFwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"])
            r = s.search(flat_rows(g, s.n_sen), [int(g["flat_n_steps"][0])], [g["bp1"]], [g["flat_w1_ssid"]])[0]
            check_flat(r, g, case)
            t = pso.load_tables()
            feats = g["flat_feat"]
            r = s.search(feats, [feats.shape[0]], [g["bp1"]], [g["flat_w1_ssid"]], ptm_tables=t, topn_seed=g["flat_ptm_seed"],
                         lists=_all_density_lists(t, feats))[0]
            check_flat(r, g, case + " (scoring)")
            s.close()
    finally:
        cap.value = old
