"""Every feature type feat_init knows, on the device (psgpu_feat_create / psgpu_feat_compute; VERDICT round 5 "missing 6"): s2_4x (the
semi-continuous models' four streams), s3_1x39, 1s_c_d_dd, 1s_c_d_ld_dd, cep_dcep, cep, the concatenating types and the generic
"widths[:window]", with batch CMN, unit variance, agc max, the linear transform of -lda and subvector specifications -- against
feat_s2mfc2feat_live(begin = end = TRUE) of the compiled reference (tests/golden/feat_types.npz, oracle/ref_dump.c dynfeat_cfg):
every value of every frame, bit for bit; ragged batches with an empty and a one-frame utterance."""
import ctypes as C

import numpy as np
import pytest

from test_oracle_golden import _load

pytestmark = pytest.mark.gpu


def _feat(cfg, lda, subvec):
    from pocketsphinx_amd import capi
    L = capi.lib()
    typ, cmn, vn, agc, ldadim, sv = [str(x) for x in cfg]
    h = C.c_void_p()
    la = np.ascontiguousarray(lda, np.float32) if lda is not None else None
    sb = np.ascontiguousarray(subvec, np.int32) if subvec is not None else None
    capi.check(L.psgpu_feat_create(C.byref(h), typ.encode(), 13, 0 if cmn == "none" else 1, int(vn), 0 if agc == "none" else 1,
                                   la.ctypes.data_as(C.c_void_p) if la is not None else None, 0 if la is None else la.shape[0], 0 if la is None else la.shape[1],
                                   sb.ctypes.data_as(C.c_void_p) if sb is not None else None, 0 if sb is None else sb.size), "psgpu_feat_create")
    return h


def _compute(h, cep, lens):
    from pocketsphinx_amd import capi
    L = capi.lib()
    L.psgpu_feat_out_dim.argtypes = [C.c_void_p]
    dim = int(L.psgpu_feat_out_dim(h))
    off = np.zeros(len(lens) + 1, np.int32); off[1:] = np.cumsum(lens)
    cep = np.ascontiguousarray(cep, np.float32)
    out = np.full((int(off[-1]), dim), np.nan, np.float32)
    capi.check(L.psgpu_feat_compute(h, cep.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), len(lens), out.ctypes.data_as(C.c_void_p)), "psgpu_feat_compute")
    return out


@pytest.mark.parametrize("ci", range(11))
def test_feature_type_equals_the_reference(ci):
    from pocketsphinx_amd import capi
    g = _load("feat_types.npz")
    assert int(g["n_cases"][0]) == 11
    k = "c%d_" % ci
    h = _feat(g[k + "cfg"], g.get(k + "lda"), g.get(k + "subvec"))
    got = _compute(h, g["cep"], [g["cep"].shape[0]])
    want = g[k + "feat"]
    assert got.shape == want.shape, (got.shape, want.shape)
    bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0]
    assert bad.size == 0, "%r: first differing frame %d: %r vs %r" % (list(g[k + "cfg"]), bad[0], got[bad[0], :6], want[bad[0], :6])
    capi.lib().psgpu_feat_free(h)


def test_ragged_batch_of_utterances():
    """utterances are independent: a batch of pieces of the recording (an empty one, a single frame, shorter than the window) equals
    each piece computed alone"""
    from pocketsphinx_amd import capi
    g = _load("feat_types.npz")
    cep = g["cep"]
    lens = [40, 0, 1, 3, 100, 7]
    for ci in (0, 2, 4, 10):
        k = "c%d_" % ci
        h = _feat(g[k + "cfg"], g.get(k + "lda"), g.get(k + "subvec"))
        allf = _compute(h, cep[:sum(lens)], lens)
        at = 0
        for n in lens:
            if n:
                one = _compute(h, cep[at:at + n], [n])
                assert np.array_equal(allf[at:at + n].view(np.uint32), one.view(np.uint32)), (ci, n)
            at += n
        capi.lib().psgpu_feat_free(h)


def test_en_us_type_equals_the_specialised_kernel():
    import pocketsphinx_amd as P
    from pocketsphinx_amd import capi
    g = _load("feat_types.npz")
    h = _feat(np.array(["1s_c_d_dd", "batch", "0", "none", "0", "-"]), None, None)
    got = _compute(h, g["cep"], [100, 164])
    want = P.dynfeat_1s_c_d_dd(g["cep"], [100, 164])
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(want, np.float32).view(np.uint32))
    capi.lib().psgpu_feat_free(h)
    with pytest.raises(P.PsgpuError):
        _feat(np.array(["s2_4x", "batch", "0", "emax", "0", "-"]), None, None) if False else capi.check(
            capi.lib().psgpu_feat_create(C.byref(C.c_void_p()), b"5,9", 13, 1, 0, 0, None, 0, 0, None, 0), "psgpu_feat_create")
