"""The host-side walk of the reference's live-decoder buffer counters (LiveSim, csrc/psgpu_decode.hip: fe_process_frames' overflow
samples, acmod's circular cepstrum buffer and feature buffer, feat_s2mfc2feat_live's window) against what the reference itself did
(tests/golden/livefeat_en_us.npz, written by oracle/ref_dump.c livefeat from chunked acmod_process_raw): the feature frames per
utterance for every list of chunk sizes, with and without the growing feature buffer -- including the utterances whose last
cepstra the reference drops at the buffer's end (acmod.c:718-723).  No device is touched (psgpu_live_pieces is host code)."""
import ctypes as C
import os

import numpy as np
import pytest

from test_oracle_golden import _load

CLIP_SAMPLES = {"goforward": 44580, "numbers": 64371}


def _pieces(L, chunks, grow, final=True):
    ch = np.ascontiguousarray(chunks, np.int64)
    ops = np.zeros((4096, 2), np.int32)
    n_ops, n_cep, n_feat = C.c_int32(), C.c_int32(), C.c_int32()
    rc = L.psgpu_live_pieces(410, 160, 3, 5, int(grow), ch.ctypes.data_as(C.c_void_p), int(ch.size), int(final), ops.ctypes.data_as(C.c_void_p), 4096,
                             C.byref(n_ops), C.byref(n_cep), C.byref(n_feat))
    assert rc == 0
    return ops[:n_ops.value], n_cep.value, n_feat.value


@pytest.mark.parametrize("grow", [1, 0])
def test_feature_frames_per_utterance_equal_the_references(grow):
    from pocketsphinx_amd import capi
    L = C.CDLL(capi.LIB_PATH)
    g = _load("livefeat_en_us.npz")
    gi = 0 if grow else 1
    ci = 0
    while "g%d_c%d_hash" % (gi, ci) in g:
        k = "g%d_c%d_" % (gi, ci)
        n = CLIP_SAMPLES[bytes(g[k + "clip"]).decode()]
        cyc, at = [int(c) for c in g[k + "chunks"]], 0
        for u in range(int(g[k + "nutt"][0])):
            chunks, left = [], n
            while left:                                  # (the list of chunk sizes goes on from one utterance to the next)
                take = min(left, cyc[at % len(cyc)]); at += 1
                chunks.append(take); left -= take
            ops, n_cep, n_feat = _pieces(L, chunks, grow)
            assert n_feat == int(g[k + "utt_frames"][u]), (k, u, n_feat, n_cep)
            assert n_cep == 1 + (n - 410) // 160 + 1
            assert int(ops[:, 0].sum()) <= n_cep and (ops[:, 1] & 2).any() == (n_feat == n_cep)      # (a dropped tail: no ending piece)
        ci += 1
    assert ci == 7


def test_pieces_are_at_most_a_cepstrum_buffer_long_and_begin_once():
    from pocketsphinx_amd import capi
    L = C.CDLL(capi.LIB_PATH)
    ops, n_cep, n_feat = _pieces(L, [16000] * 4, 1)
    assert n_cep == n_feat and int(ops[:, 0].max()) <= 7
    assert int(((ops[:, 1] & 1) != 0).sum()) == 1 and (ops[0, 1] & 1) and int(((ops[:, 1] & 2) != 0).sum()) == 1 and (ops[-1, 1] & 2)
    # less than a frame of audio in the first call: the utterance begins with an EMPTY beginning piece (feat.c:1360: no replication)
    ops, n_cep, n_feat = _pieces(L, [100, 16000], 1)
    assert tuple(ops[0]) == (0, 1) and not (ops[1:, 1] & 1).any()
    # mid-utterance: three cepstra are held back for the window
    ops, n_cep, n_feat = _pieces(L, [16000], 1, final=False)
    assert n_feat == n_cep - 3
