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
