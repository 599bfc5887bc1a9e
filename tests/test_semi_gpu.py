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
