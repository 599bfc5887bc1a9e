"""GPU parity of the HMM Viterbi step (psgpu_hmm_vit_eval*, the hmm_vit_eval
replacement) against state dumps of the unmodified reference (hmm_*.npz:
3-state en-us and 5-state tidigits, multiplex and not; synthetic 4-, 2- and
1-state contexts for the any-topology form) and the pinned oracle.
Bit-exact int32 scores, history pointers, propagated ssids, best scores."""
import ctypes as C

import numpy as np
import pytest

import pso
from test_oracle_golden import _load

pytestmark = pytest.mark.gpu


def to_recs(P, before, mpx):
    r = np.zeros(before.shape[0], P.HMM_REC)
    r["score"] = before[:, 0:5]; r["history"] = before[:, 5:10]
    r["out_score"] = before[:, 10]; r["out_history"] = before[:, 11]
    r["senid"] = before[:, 12:17].astype(np.uint16); r["bestscore"] = before[:, 17]
    r["tmatid_mpx"] = (before[:, 18].astype(np.uint16) | np.where(mpx != 0, 0x8000, 0).astype(np.uint16))
    return r


def from_recs(r):
    out = np.empty((r.size, pso.HMM_FIELDS), np.int32)
    out[:, 0:5] = r["score"]; out[:, 5:10] = r["history"]
    out[:, 10] = r["out_score"]; out[:, 11] = r["out_history"]
    out[:, 12:17] = r["senid"]; out[:, 17] = r["bestscore"]
    out[:, 18] = r["tmatid_mpx"] & 0x7fff
    return out


@pytest.mark.parametrize("case", ["en_us_3st", "tidigits_5st", "syn_4st", "syn_2st", "syn_1st"])
def test_hmm_steps_match_reference(case):
    """(syn_*: synthetic contexts with 4, 2, 1 emitting states -- hmm_vit_eval_anytopo, hmm.c:710-784)"""
    import pocketsphinx_amd as P
    g = _load("hmm_%s.npz" % case)
    ne = int(g["n_emit"][0])
    ctx = P.HmmContext(g["tp"], g["sseq"], int(g["n_sen"][0]))
    for t in range(g["before"].shape[0]):
        recs = to_recs(P, g["before"][t], g["mpx"])
        best = ctx.vit_eval(recs, g["senscr"][t])
        got, want = from_recs(recs), g["after"][t].copy()
        # states beyond n_emit are not part of the HMM
        if ne < 5:
            for a in (got, want):
                a[:, ne:5] = 0; a[:, 5 + ne:10] = 0; a[:, 12 + ne:17] = 0
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, "step %d HMM %d (mpx %d)\nbefore %s\nref    %s\ngpu    %s" % (
            t, bad[0], g["mpx"][bad[0]], g["before"][t][bad[0]], want[bad[0]], got[bad[0]])
        assert np.array_equal(recs["bestscore"], g["ret"][t])
        assert best == max(int(g["ret"][t].max()), -0x20000000)
    ctx.close()


def test_active_list_multi_utt_dev():
    """Device entry point: sparse active list over an arena that holds HMMs of
    several utterances, each scored against its own senone-score row; inactive
    records must stay untouched and best[] is a per-utterance maximum."""
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import capi
    g = _load("hmm_en_us_3st.npz")
    n_sen = int(g["n_sen"][0])
    ctx = P.HmmContext(g["tp"], g["sseq"], n_sen)
    n_utt, T = 4, g["before"].shape[0]
    n = g["before"].shape[1]
    rng = np.random.default_rng(3)
    # arena = step-0 states; utterance u scored with senone row u
    recs = to_recs(P, g["before"][0], g["mpx"])
    utt = rng.integers(0, n_utt, n).astype(np.uint16)
    active = np.sort(rng.choice(n, n // 3, replace=False)).astype(np.int32)
    scr = np.ascontiguousarray(g["senscr"][:n_utt])
    dev = torch.device("cuda", 0)
    d_recs = torch.from_numpy(recs.view(np.uint8).reshape(n, 64).copy()).to(dev)
    d_act = torch.from_numpy(active).to(dev)
    d_utt = torch.from_numpy(utt.view(np.int16).copy()).to(dev)
    d_scr = torch.from_numpy(scr).to(dev)
    d_best = torch.full((n_utt,), -0x20000000, dtype=torch.int32, device=dev)
    L = capi.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(L.psgpu_hmm_vit_eval_dev(ctx.h, C.c_void_p(d_recs.data_ptr()), C.c_void_p(d_act.data_ptr()),
                                        int(active.size), C.c_void_p(d_utt.data_ptr()),
                                        C.c_void_p(d_scr.data_ptr()), n_sen,
                                        C.c_void_p(d_best.data_ptr()), st), "hmm_vit_eval_dev")
    torch.cuda.synchronize()
    got = d_recs.cpu().numpy().reshape(-1).view(P.HMM_REC)
    # oracle: same thing per utterance row
    want = recs.copy()
    best = np.full(n_utt, -0x20000000, np.int64)
    for u in range(n_utt):
        gg = dict(g); gg["senscr"] = scr[u][None, :].repeat(T, axis=0)
        after, ret = pso.hmm_step_oracle(gg, 0)
        sel = active[utt[active] == u]
        w = to_recs(P, after, g["mpx"])
        want[sel] = w[sel]
        want["bestscore"][sel] = ret[sel]
        if sel.size:
            best[u] = max(best[u], int(ret[sel].max()))
    for f in ("score", "history", "senid"):
        assert np.array_equal(got[f][:, :3], want[f][:, :3]), f
    for f in ("out_score", "out_history", "bestscore", "tmatid_mpx"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(d_best.cpu().numpy().astype(np.int64), best)
    ctx.close()


@pytest.mark.parametrize("n_rep", [1, 3000])
def test_dense_list_dev(n_rep):
    """Device entry point with a DENSE list (active == NULL): full wavefronts of consecutive records travel through LDS (1 KB a
    wave-instruction, quads swizzled) and, past 2^20 records, around the caches (the streaming instantiation: n_rep = 3000 tiles
    the golden's records to ~1.3 M); the tail that does not fill a wavefront takes the record-by-record path.  Every record equals
    the reference's hmm_vit_eval of the golden's step."""
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import capi
    g = _load("hmm_en_us_3st.npz")
    n_sen = int(g["n_sen"][0])
    ctx = P.HmmContext(g["tp"], g["sseq"], n_sen)
    n0 = g["before"].shape[1]
    recs0 = to_recs(P, g["before"][0], g["mpx"])
    want0 = to_recs(P, g["after"][0], g["mpx"]); want0["bestscore"] = g["ret"][0]
    n = n0 * n_rep - 17                                           # (not a multiple of 64: a tail)
    recs = np.tile(recs0, n_rep)[:n]; want = np.tile(want0, n_rep)[:n]
    if n_rep > 1:
        assert n > (1 << 20)
    dev = torch.device("cuda", 0)
    d_recs = torch.from_numpy(recs.view(np.uint8).reshape(n, 64).copy()).to(dev)
    d_scr = torch.from_numpy(np.ascontiguousarray(g["senscr"][0])).to(dev)
    d_best = torch.full((1,), -0x20000000, dtype=torch.int32, device=dev)
    L = capi.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(L.psgpu_hmm_vit_eval_dev(ctx.h, C.c_void_p(d_recs.data_ptr()), None, n, None, C.c_void_p(d_scr.data_ptr()), n_sen,
                                        C.c_void_p(d_best.data_ptr()), st), "hmm_vit_eval_dev")
    torch.cuda.synchronize()
    got = d_recs.cpu().numpy().reshape(-1).view(P.HMM_REC)
    for f in ("score", "history", "senid"):
        assert np.array_equal(got[f][:, :3], want[f][:, :3]), f
    for f in ("out_score", "out_history", "bestscore", "tmatid_mpx"):
        assert np.array_equal(got[f], want[f]), f
    assert int(d_best.item()) == max(int(g["ret"][0].max()), -0x20000000)
    ctx.close()


def test_empty_and_errors():
    import pocketsphinx_amd as P
    g = _load("hmm_en_us_3st.npz")
    ctx = P.HmmContext(g["tp"], g["sseq"], int(g["n_sen"][0]))
    assert ctx.vit_eval(np.zeros(0, P.HMM_REC), g["senscr"][0]) == -0x20000000
    with pytest.raises(P.PsgpuError):
        P.HmmContext(np.zeros((2, 6, 7), np.uint8), np.zeros((3, 6), np.uint16), 10)   # 6 states: beyond HMM_MAX_NSTATE
    ctx.close()
