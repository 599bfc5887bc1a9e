"""GPU parity of the dynamic feature kernel (psgpu_feat_1s_c_d_dd, the
feat_s2mfc2feat_live(begin, end) replacement) against the reference's output on
the bundled cepstra and against the pinned oracle on ragged batches."""
import ctypes as C

import numpy as np
import pytest

import pso
from test_oracle_golden import _load

pytestmark = pytest.mark.gpu


def _oracle(cep):
    out = np.empty((cep.shape[0], 3 * cep.shape[1]), np.float32)
    L = pso.lib()
    L.pso_dynfeat_1s_c_d_dd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    c = np.ascontiguousarray(cep, np.float32)
    L.pso_dynfeat_1s_c_d_dd(c.ctypes.data, c.shape[0], c.shape[1], out.ctypes.data)
    return out


def test_dynfeat_matches_reference():
    import pocketsphinx_amd as P
    g = _load("dynfeat_goforward.npz")
    out = P.dynfeat_1s_c_d_dd(g["cep"], [g["cep"].shape[0]])
    assert out.tobytes() == np.ascontiguousarray(g["feat"], np.float32).tobytes()


def test_dynfeat_ragged_batch_and_scoring(tables):
    """Ragged / one-frame / empty utterances, frames with negative c0 (skipped by the
    mean), and the features feeding the scorer: same senone scores as from the
    reference's features."""
    import pocketsphinx_amd as P
    g = _load("dynfeat_goforward.npz")
    rng = np.random.default_rng(3)
    base = g["cep"]
    lens = [1, 0, 2, 7, 130, 3, 40]
    cep = base[rng.integers(0, base.shape[0], sum(lens))].copy()
    cep[rng.random(cep.shape[0]) < 0.1, 0] *= -1.0
    out = P.dynfeat_1s_c_d_dd(cep, lens)
    o = 0
    for n in lens:
        if n:
            assert out[o:o + n].tobytes() == _oracle(cep[o:o + n]).tobytes(), "utterance at %d" % o
        o += n
    m = P.PtmModel(tables)
    a = P.PtmMgau(m).score_utts(P.dynfeat_1s_c_d_dd(g["cep"], [base.shape[0]]), [base.shape[0]], want_topn=False)
    b = P.PtmMgau(m).score_utts(g["feat"], [base.shape[0]], want_topn=False)
    assert np.array_equal(a["senscr"], b["senscr"])
    m.close()
