"""GPU parity of the stateful per-call scorer (psgpu_ptm_frame_eval, the
ps_mgau_t::frame_eval replacement) against (a) every frame_eval call the
unmodified reference made during real decodes (senlog fixtures: active lists,
history-slot reuse by the fwdtree search 5 frames behind the phone loop,
pass-2 codebook masking after acmod_rewind) and (b) the pinned oracle on
adversarial call sequences.  Bit-exact int16 scores and int32 top-N lists."""
import numpy as np
import pytest

import pso
from test_oracle_golden import _load, dup_tables

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_model(tables):
    import pocketsphinx_amd as P
    m = P.PtmModel(tables)
    yield m
    m.close()


@pytest.mark.parametrize("case", ["default", "fwdtree_only", "ptm_topn2", "ptm_topn6_ds2"])
def test_senlog_replay_gpu(tables, gpu_model, case):
    """topn 2 / topn 6 + ds 2 go through the any-shape kernels (exact sequential
    procedure only); the default shape through the specialised ones."""
    import pocketsphinx_amd as P
    g = _load("senlog_%s.npz" % case)
    pr = pso.senlog_params(g)
    own = None
    if "topn" in pr or "ds" in pr:
        own = gpu_model = P.PtmModel(tables, topn=int(pr.get("topn", 4)), ds_ratio=int(pr.get("ds", 1)))
    st = P.PtmState(gpu_model, int(tables["n_fast_hist"][0]))
    n = int(g["call_frame"].size)
    off = g["call_act_off"]
    scr = np.empty((n, gpu_model.n_sen), np.int16)
    for c in range(n):
        na = int(g["call_nact"][c])
        act = None if na < 0 else g["call_act"][off[c]:off[c] + na]
        scr[c] = st.frame_eval(g["call_feat"][c], int(g["call_frame"][c]), active=act,
                               compallsen=(na < 0), frame_idx=int(g["call_frame_idx"][c]))
    bad = np.nonzero(pso.row_hash(scr) != g["call_scr_hash"])[0]
    assert bad.size == 0, "first mismatching call %d (frame %d)" % (bad[0], g["call_frame"][bad[0]])
    assert np.array_equal(scr[g["sample_idx"]], g["call_scr_sample"])
    st.close()
    if own is not None:
        # the batched entry serves the shape too (round 6: ptm_batch_topn_generic): the utterance's frames in order -- the first call of
        # every frame of this decode is the phone loop's, all codebooks active -- against the pinned oracle with the same knobs
        fr = g["call_frame"]
        first = np.array([np.nonzero(fr == t)[0][0] for t in range(int(fr.max()) + 1)])
        feats = g["call_feat"][first]
        r = P.PtmMgau(own).score_utts(feats, [feats.shape[0]])
        o = pso.OraclePTM(tables, topn=int(pr.get("topn", 4)), ds_ratio=int(pr.get("ds", 1)))
        scr_o, cw_o, raw_o = o.score_utt(feats, reset_hist=True)
        assert np.array_equal(r["senscr"], scr_o)
        assert np.array_equal(r["topn_cw"].reshape(cw_o.shape), cw_o) and np.array_equal(r["topn_score"].reshape(raw_o.shape), raw_o)
        own.close()


def _random_list(rng, n_sen, sen2cb, mode):
    """uint8 delta list (acmod_flags2list) of a random active set."""
    flags = np.zeros(n_sen, np.uint8)
    if mode == 0:                                   # CI senones only (phone loop)
        flags[:126] = 1
    elif mode == 1:                                 # a few codebooks, sparse
        cbs = rng.choice(int(sen2cb.max()) + 1, size=rng.integers(1, 6), replace=False)
        cand = np.nonzero(np.isin(sen2cb, cbs))[0]
        flags[rng.choice(cand, size=min(cand.size, rng.integers(1, 200)), replace=False)] = 1
    elif mode == 2:                                 # wide random set with big gaps
        flags[rng.choice(n_sen, size=rng.integers(1, 600), replace=False)] = 1
    else:                                           # a single far senone: bridging entries
        flags[n_sen - 1 - rng.integers(0, 50)] = 1
    return pso.flags2list(flags)


@pytest.mark.parametrize("dup", [0, 1])
def test_vs_oracle_call_patterns(tables, dup):
    """Random mixes of the reference's three call patterns, HIP vs oracle, every
    call memcmp'd (scores + the slot's top-N lists).  dup=1 duplicates
    codewords so that exact ties and the sequential fallback are exercised
    with partially active codebooks."""
    import pocketsphinx_amd as P
    t = dup_tables(tables) if dup else tables
    rng = np.random.default_rng(11 + dup)
    base = _load("ptm_adversarial.npz" if dup else "ptm_goforward.npz")["feat"]
    H = int(tables["n_fast_hist"][0])
    m = P.PtmModel(t)
    st = P.PtmState(m, H)
    o = pso.OraclePTM(t, n_fast_hist=H)
    sen2cb = tables["sen2cb"]
    n_sen = m.n_sen

    def both(feat, frame, act, call, frame_idx):
        o.set_frame_idx(frame_idx)
        a = o.frame_eval(feat, frame, active=act, compallsen=call)
        b = st.frame_eval(feat, frame, active=act, compallsen=call, frame_idx=frame_idx)
        assert np.array_equal(a, b), "scores differ at frame %d (frame_idx %d)" % (frame, frame_idx)
        ocur = o.cur_topn().reshape(m.n_chain, m.topn, 2)
        cw, sc = st.cur_topn()
        assert np.array_equal(ocur[..., 0], cw), "top-N codewords differ at frame %d" % frame
        assert np.array_equal(ocur[..., 1], sc), "top-N slot scores differ at frame %d" % frame

    T = 60
    feats = base[rng.integers(0, base.shape[0], T)]
    # pass 1: phone loop at t (fresh), search at t-5 (slot reuse, arbitrary list)
    for t_ in range(T):
        both(feats[t_], t_, _random_list(rng, n_sen, sen2cb, 0), False, t_)
        if t_ >= 5:
            both(feats[t_ - 5], t_ - 5, _random_list(rng, n_sen, sen2cb, int(rng.integers(1, 4))), False, t_)
    # pass 2 after acmod_rewind: fresh evaluation with a codebook subset, no reset
    for t_ in range(T):
        mode = int(rng.integers(1, 4))
        both(feats[t_], t_, _random_list(rng, n_sen, sen2cb, mode), False, t_)
        if t_ % 7 == 3:                              # same frame again: reuse + 96-overwrite path
            both(feats[t_], t_, _random_list(rng, n_sen, sen2cb, 2), False, t_ + 1)
    # compallsen calls interleaved with an empty list
    for t_ in range(10):
        both(feats[t_], t_, None, True, t_)
        both(feats[t_], t_, np.zeros(0, np.uint8), False, t_ + 1)
    st.close()
    m.close()


def test_reset_hist(tables, gpu_model):
    """psgpu_ptm_state_reset == ptm_mgau_reset_fast_hist: scores after a reset
    equal those of a fresh state."""
    import pocketsphinx_amd as P
    g = _load("ptm_goforward.npz")
    H = int(tables["n_fast_hist"][0])
    a = P.PtmState(gpu_model, H)
    for t_ in range(20):
        a.frame_eval(g["feat"][t_ + 100], t_, compallsen=True, frame_idx=t_)
    a.reset_hist()
    b = P.PtmState(gpu_model, H)
    for t_ in range(12):
        x = a.frame_eval(g["feat"][t_], t_, compallsen=True, frame_idx=t_)
        y = b.frame_eval(g["feat"][t_], t_, compallsen=True, frame_idx=t_)
        assert np.array_equal(x, y)
    a.close(); b.close()


@pytest.mark.parametrize("case", ["default", "fwdtree_only"])
def test_lookahead_cache_is_invisible(tables, gpu_model, case):
    """psgpu_ptm_state_lookahead: announce every pass's frames up front (what the
    shim does from acmod's feature buffer).  Pass-1 calls (phone loop: all
    codebooks, search: slot reuse) must be answered from the batched pass,
    pass-2 calls (codebook subsets) must fall back to the per-call kernels on a
    ring brought up to date -- and every score vector must still equal the
    reference's, call for call."""
    import pocketsphinx_amd as P
    g = _load("senlog_%s.npz" % case)
    st = P.PtmState(gpu_model, int(tables["n_fast_hist"][0]))
    n = int(g["call_frame"].size)
    off = g["call_act_off"]
    fr, fi = g["call_frame"], g["call_frame_idx"]
    scr = np.empty((n, gpu_model.n_sen), np.int16)
    c = 0
    while c < n:
        if fr[c] == 0 and fi[c] == 0:
            # start of a pass: collect the feature vector of every frame up to the next pass start
            e = c + 1
            while e < n and not (fr[e] == 0 and fi[e] == 0):
                e += 1
            T = int(fr[c:e].max()) + 1
            feats = np.zeros((T, gpu_model.veclen), np.float32)
            for k in range(c, e):
                feats[fr[k]] = g["call_feat"][k]
            st.lookahead(feats, 0)
        na = int(g["call_nact"][c])
        act = None if na < 0 else g["call_act"][off[c]:off[c] + na]
        scr[c] = st.frame_eval(g["call_feat"][c], int(fr[c]), active=act, compallsen=(na < 0), frame_idx=int(fi[c]))
        c += 1
    bad = np.nonzero(pso.row_hash(scr) != g["call_scr_hash"])[0]
    assert bad.size == 0, "first mismatching call %d (frame %d)" % (bad[0], fr[bad[0]])
    served, batches = st.lookahead_stats()
    assert served > n // 3 and batches >= 1, (served, batches, n)
    st.close()
