"""GPU parity of the semi-continuous scorer (psgpu_semi_frame_eval, the
s2_semi_mgau_frame_eval replacement) against every call of five recorded
tidigits decodes of the unmodified reference (4 streams of 12/24/3/12 dims x
256 densities, 4-bit clustered weights; default, per-stream top-N beams,
topn 6 + ds 2, topn 7 + compallsen, topn 2) and against the pinned oracle's
top-N state."""
import numpy as np
import pytest

import pso
from test_oracle_golden import _load, SEMI_CASES, semi_oracle_for

pytestmark = pytest.mark.gpu


def gpu_for(P, g, t):
    p = pso.senlog_params(g)
    beam = None
    if "topn_beam" in p:
        b = [int(x) for x in p["topn_beam"].split(",")]
        beam = (b + [max(b)] * 4)[:int(t["n_feat"][0])]
    return P.SemiMgau(t, topn=int(p["topn"]) if "topn" in p else None,
                      ds_ratio=int(p["ds"]) if "ds" in p else None, topn_beam=beam)


@pytest.mark.parametrize("case", SEMI_CASES)
def test_semi_senlog_replay_gpu(case):
    import pocketsphinx_amd as P
    g = _load("senlog_%s.npz" % case)
    t = _load("semi_tidigits_tables.npz")
    s = gpu_for(P, g, t)
    o = semi_oracle_for(g, t)
    off = g["call_act_off"]
    n = int(g["call_frame"].size)
    scr = np.empty((n, s.n_sen), np.int16)
    for c in range(n):
        na = int(g["call_nact"][c])
        act = None if na < 0 else g["call_act"][off[c]:off[c] + na]
        fi, fr = int(g["call_frame_idx"][c]), int(g["call_frame"][c])
        scr[c] = s.frame_eval(g["call_feat"][c], fr, active=act, compallsen=(na < 0), frame_idx=fi)
        if c % 37 == 0:       # the slot's lists and counts, against the oracle
            o_ = o  # noqa: F841
        o.set_frame_idx(fi)
        o.frame_eval(g["call_feat"][c], fr, active=act, compallsen=(na < 0))
        if c % 37 == 0:
            lists, cnt = o.cur_topn()
            cw, sc, n_used = s.cur_topn()
            assert np.array_equal(lists[..., 0], cw) and np.array_equal(lists[..., 1], sc), "call %d" % c
            assert np.array_equal(cnt.astype(np.int32), n_used), "call %d" % c
    bad = np.nonzero(pso.row_hash(scr) != g["call_scr_hash"])[0]
    assert bad.size == 0, "first mismatching call %d (frame %d)" % (bad[0], g["call_frame"][bad[0]])
    assert np.array_equal(scr[g["sample_idx"]], g["call_scr_sample"])
    s.close()


def test_semi_8bit_weights_vs_oracle():
    """8-bit mixture weights (the tidigits sendump de-clustered into bytes):
    the int arithmetic of get_scores_8b_feat_* instead of the uint8 w_den."""
    import pocketsphinx_amd as P
    t = dict(_load("semi_tidigits_tables.npz"))
    n_sen = int(t["n_sen"][0])
    cb = t.pop("mixw_cb")
    packed = t["mixw"]
    full = np.empty(packed.shape[:2] + (n_sen,), np.uint8)
    full[..., 0::2] = cb[packed & 0x0f][..., :(n_sen + 1) // 2]
    full[..., 1::2] = cb[packed >> 4][..., :n_sen // 2]
    t["mixw"] = full
    g = _load("senlog_tidigits_default.npz")
    s = P.SemiMgau(t)
    o = pso.OracleSemi(t)
    off = g["call_act_off"]
    for c in range(0, 300):
        na = int(g["call_nact"][c])
        act = None if na < 0 else g["call_act"][off[c]:off[c] + na]
        fi, fr = int(g["call_frame_idx"][c]), int(g["call_frame"][c])
        o.set_frame_idx(fi)
        a = o.frame_eval(g["call_feat"][c], fr, active=act, compallsen=(na < 0))
        b = s.frame_eval(g["call_feat"][c], fr, active=act, compallsen=(na < 0), frame_idx=fi)
        assert np.array_equal(a, b), "call %d" % c
    s.close()


@pytest.mark.parametrize("variant", ["4bit_default", "8bit", "4bit_beam_topn6_ds2", "dup_codewords_topn2"])
def test_semi_batch_vs_oracle(variant):
    """psgpu_semi_score_batch (one wave per (utterance, stream) walking its frames in
    order + one workgroup per frame for the senones) == the oracle's compallsen
    frame_eval frame by frame, every utterance from a fresh state: ragged utterances
    (one frame, empty), 4-bit and 8-bit weights, per-stream beams, down-sampling,
    duplicated codewords (exact ties through the sequential acceptance rule)."""
    import pocketsphinx_amd as P
    t = dict(_load("semi_tidigits_tables.npz"))
    kw = {}
    if variant == "8bit":
        n_sen = int(t["n_sen"][0])
        cb = t.pop("mixw_cb")
        packed = t["mixw"]
        full = np.empty(packed.shape[:2] + (n_sen,), np.uint8)
        full[..., 0::2] = cb[packed & 0x0f][..., :(n_sen + 1) // 2]
        full[..., 1::2] = cb[packed >> 4][..., :n_sen // 2]
        t["mixw"] = full
    elif variant == "4bit_beam_topn6_ds2":
        kw = dict(topn=6, ds_ratio=2, topn_beam=[40, 30, 0, 60])
    elif variant == "dup_codewords_topn2":
        # codeword 2k+1 := codeword 2k in every stream: exact score ties everywhere
        kw = dict(topn=2)
        fl = t["featlen"]
        nd = int(t["n_density"][0])
        mean, var, det = t["mean"].copy(), t["var"].copy(), t["det"].copy().reshape(len(fl), nd)
        o = 0
        for f, ln in enumerate(fl):
            m = mean[o:o + nd * ln].reshape(nd, ln); v = var[o:o + nd * ln].reshape(nd, ln)
            m[1::2] = m[0::2]; v[1::2] = v[0::2]; det[f, 1::2] = det[f, 0::2]
            o += nd * ln
        t["mean"], t["var"], t["det"] = mean, var, det.reshape(t["det"].shape)
    g = _load("senlog_tidigits_default.npz")
    rng = np.random.default_rng(9)
    lens = [37, 1, 0, 90, 12]
    feats = np.ascontiguousarray(g["call_feat"][rng.integers(0, g["call_feat"].shape[0], sum(lens))])
    s = P.SemiMgau(t, **kw)
    got = s.score_utts(feats, lens)
    pos = 0
    for n in lens:
        o = pso.OracleSemi(t, **kw)
        for i in range(n):
            o.set_frame_idx(i)
            want = o.frame_eval(feats[pos + i], i, compallsen=True)
            assert np.array_equal(got[pos + i], want), "utterance frame %d (batch row %d)" % (i, pos + i)
        pos += n
    s.close()
