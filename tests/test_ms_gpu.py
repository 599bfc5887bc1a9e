"""GPU parity of the multi-stream / continuous scorer (psgpu_ms_*, the
ms_cont_mgau_frame_eval replacement) against every call of four recorded
decodes of the unmodified reference (an4_ci_cont: 102 codebooks x 1 density x
39 dims; en-us forced through the ms scorer: 42 x 3 streams x 128 densities,
top-N scan) and, for the batched entry, against the pinned oracle."""
import numpy as np
import pytest

import pso
from test_oracle_golden import _load, MS_CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case,tab", MS_CASES)
def test_ms_senlog_replay_gpu(case, tab):
    import pocketsphinx_amd as P
    g = _load("senlog_%s.npz" % case)
    t = _load("%s.npz" % tab)
    p = pso.senlog_params(g)
    s = P.MsMgau(t, topn=int(p["topn"]) if "topn" in p else None, aw=int(p["aw"]) if "aw" in p else None)
    off = g["call_act_off"]
    n = int(g["call_frame"].size)
    scr = np.empty((n, s.n_sen), np.int16)
    for c in range(n):
        na = int(g["call_nact"][c])
        act = None if na < 0 else g["call_act"][off[c]:off[c] + na]
        scr[c] = s.frame_eval(g["call_feat"][c], active=act, compallsen=(na < 0))
    bad = np.nonzero(pso.row_hash(scr) != g["call_scr_hash"])[0]
    assert bad.size == 0, "first mismatching call %d (frame %d)" % (bad[0], g["call_frame"][bad[0]])
    assert np.array_equal(scr[g["sample_idx"]], g["call_scr_sample"])
    s.close()


@pytest.mark.parametrize("tab,topn,aw", [("ms_en_us_tables", 4, 1), ("ms_en_us_tables", 1, 3), ("ms_an4_tables", 1, 1)])
def test_ms_batch_vs_oracle(tab, topn, aw):
    """psgpu_ms_score_batch (frames x codebooks in two launches) == the oracle's
    compallsen scoring frame by frame; plus empty-list and ragged behaviour."""
    import pocketsphinx_amd as P
    t = _load("%s.npz" % tab)
    g = _load("senlog_ms_en_us_default.npz" if "en_us" in tab else "senlog_ms_an4_default.npz")
    feats = np.ascontiguousarray(g["call_feat"][::3][:200])
    s = P.MsMgau(t, topn=topn, aw=aw)
    o = pso.OracleMs(t, topn=topn, aw=aw)
    got = s.score_frames(feats)
    for i in range(feats.shape[0]):
        want = o.frame_eval(feats[i], compallsen=True)
        assert np.array_equal(got[i], want), "frame %d" % i
    assert s.score_frames(feats[:0]).shape == (0, s.n_sen)
    # per-call with an empty list leaves the score buffer untouched
    before = s.frame_eval(feats[0], compallsen=True)
    after = s.frame_eval(feats[1], active=np.zeros(0, np.uint8), compallsen=False)
    assert np.array_equal(before, after)
    s.close()


@pytest.mark.parametrize("case,tab", MS_CASES)
def test_ms_lookahead_cache_is_invisible(case, tab):
    """psgpu_ms_lookahead: announce an utterance's frames once; every call of every pass
    (the recorded decodes have fwdtree + fwdflat, i.e. each frame is asked for up to three
    times with different active lists) is answered from the one batched pass -- and
    still equals the reference's score vector call for call."""
    import pocketsphinx_amd as P
    g = _load("senlog_%s.npz" % case)
    t = _load("%s.npz" % tab)
    p = pso.senlog_params(g)
    s = P.MsMgau(t, topn=int(p["topn"]) if "topn" in p else None, aw=int(p["aw"]) if "aw" in p else None)
    off = g["call_act_off"]
    fr = g["call_frame"]
    n = int(fr.size)
    T = int(fr.max()) + 1
    feats = np.zeros((T, s.veclen), np.float32)
    for c in range(n):
        feats[fr[c]] = g["call_feat"][c]
    s.lookahead(feats, 0)
    scr = np.empty((n, s.n_sen), np.int16)
    for c in range(n):
        na = int(g["call_nact"][c])
        act = None if na < 0 else g["call_act"][off[c]:off[c] + na]
        scr[c] = s.frame_eval(g["call_feat"][c], active=act, compallsen=(na < 0), frame=int(fr[c]))
    bad = np.nonzero(pso.row_hash(scr) != g["call_scr_hash"])[0]
    assert bad.size == 0, "first mismatching call %d (frame %d)" % (bad[0], fr[bad[0]])
    served, batches = s.lookahead_stats()
    assert batches == 1 and served >= n - 2, (served, batches, n)
    s.close()


def test_ms_lookahead_keeps_the_stale_id_side_effect():
    """Served calls must leave the per-call lists as the reference's calls would: a later
    call whose lists stay unfilled (every density below WORST_DIST) reuses the ids of the
    last call in which its codebook was active (ms_gauden.c:438-440).  Mix served calls
    (different active sets per call), cache misses (a changed vector), and such
    degenerate frames; the stateful oracle is the judge."""
    import pocketsphinx_amd as P
    from test_random_models_gpu import _gauss, _logadd8
    rng = np.random.default_rng(77)
    n_mgau, n_feat, n_den, n_sen = 6, 2, 12, 90
    featlen = np.array([7, 5], np.int32)
    mean, var, det = _gauss(rng, n_mgau, n_feat, n_den, featlen)
    t = dict(n_mgau=np.array([n_mgau]), n_feat=np.array([n_feat]), n_density=np.array([n_den]),
             n_sen=np.array([n_sen]), max_topn=np.array([3]), aw=np.array([1]), featlen=featlen,
             mean=mean, var=var, det=det, pdf=rng.integers(0, 256, (n_sen, n_feat, n_den)).astype(np.uint8),
             sen2mgau=rng.integers(0, n_mgau, n_sen).astype(np.uint32), logadd=_logadd8(),
             logadd_size=np.array([256]), logadd_width=np.array([1]), log_zero=np.array([-524288]))
    g, o = P.MsMgau(t), pso.OracleMs(t)
    T = 40
    feats = rng.standard_normal((T, 12)).astype(np.float32)
    g.lookahead(feats, 0)

    def act_list():
        flags = np.zeros(n_sen, np.uint8)
        flags[rng.choice(n_sen, size=int(rng.integers(1, 30)), replace=False)] = 1
        return pso.flags2list(flags)

    def both(x, frame, act, call):
        a = o.frame_eval(x, active=act, compallsen=call)
        b = g.frame_eval(x, active=act, compallsen=call, frame=frame)
        assert np.array_equal(a, b), "frame %d" % frame

    for t_ in range(T):
        both(feats[t_], t_, act_list(), False)                     # served, partial active set
        if t_ % 5 == 4:
            both(feats[t_] * 1e5, t_, act_list(), False)           # miss: everything below WORST_DIST -> stale ids
        if t_ % 7 == 3:
            both(feats[t_], t_, None, True)                        # served, compallsen
        if t_ % 11 == 10:
            both(feats[t_] * 1e5, -1, None, True)                  # degenerate compallsen call on the per-call path
    served, batches = g.lookahead_stats()
    assert batches == 1 and served >= T
    g.close()
