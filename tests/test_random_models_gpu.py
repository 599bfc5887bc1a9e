"""Differential tests on synthetic models: shapes and table contents the
bundled fixtures never reach (odd stream lengths, tiny / non-power-of-two
codebooks, every top-N size, 8-bit and 4-bit weights, beams, transition
matrices with missing arcs, BAD_SSID states ...).  HIP vs the oracle (which is
pinned to the reference on the real models); every call memcmp'd."""
import numpy as np
import pytest

import pso

pytestmark = pytest.mark.gpu


def _logadd8():
    return np.ascontiguousarray(pso.load_tables()["logadd8"], np.uint8)


def _gauss(rng, n_cb, n_feat, n_den, featlen):
    tot = int(np.sum(featlen))
    mean = rng.standard_normal((n_cb, n_den * tot)).astype(np.float32).reshape(-1)
    # precomputed variances are integer-valued floats in [1, 5e7], dets integer-valued
    var = np.floor(np.exp(rng.uniform(0, 12, n_cb * n_den * tot))).astype(np.float32)
    det = np.floor(rng.uniform(-500000, 400000, (n_cb, n_feat, n_den))).astype(np.float32)
    return mean, var, det


@pytest.mark.parametrize("seed", range(8))
def test_semi_random_model(seed):
    import pocketsphinx_amd as P
    rng = np.random.default_rng(100 + seed)
    n_feat = int(rng.integers(1, 5))
    featlen = rng.integers(1, 17, n_feat).astype(np.int32)
    n_den = int(rng.choice([5, 17, 64, 100, 200, 256]))
    topn = int(rng.integers(1, min(8, n_den) + 1))
    n_sen = int(rng.integers(3, 400))
    four = bool(seed % 2)
    mean, var, det = _gauss(rng, 1, n_feat, n_den, featlen)
    t = dict(n_feat=np.array([n_feat]), n_density=np.array([n_den]), n_sen=np.array([n_sen]),
             max_topn=np.array([topn]), ds_ratio=np.array([int(rng.integers(1, 3))]),
             n_fast_hist=np.array([int(rng.integers(2, 8))]), featlen=featlen,
             topn_beam=rng.integers(0, 60, n_feat).astype(np.uint8) * (seed % 3 == 0),
             mean=mean, var=var, det=det.reshape(n_feat, n_den), logadd8=_logadd8())
    if four:
        t["mixw"] = rng.integers(0, 256, (n_feat, n_den, (n_sen + 1) // 2)).astype(np.uint8)
        t["mixw_cb"] = rng.integers(0, 200, 16).astype(np.uint8)
    else:
        t["mixw"] = rng.integers(0, 160, (n_feat, n_den, n_sen)).astype(np.uint8)
    g, o = P.SemiMgau(t), pso.OracleSemi(t)
    fi = 0
    for step in range(60):
        # near a codeword mean now and then, so that ties / near-ties happen
        x = rng.standard_normal(int(featlen.sum())).astype(np.float32)
        frame = fi if rng.random() < 0.8 else max(0, fi - int(rng.integers(1, 3)))
        call = rng.random() < 0.3
        flags = (rng.random(n_sen) < rng.uniform(0.05, 0.9)).astype(np.uint8)
        act = None if call else pso.flags2list(flags)
        o.set_frame_idx(fi)
        a = o.frame_eval(x, frame, active=act, compallsen=call)
        b = g.frame_eval(x, frame, active=act, compallsen=call, frame_idx=fi)
        assert np.array_equal(a, b), "seed %d step %d (frame %d/%d, topn %d, n_den %d, 4bit %s)" % (
            seed, step, frame, fi, topn, n_den, four)
        if frame == fi:
            fi += 1
    g.close()


@pytest.mark.parametrize("seed", range(8))
def test_ms_random_model(seed):
    import pocketsphinx_amd as P
    rng = np.random.default_rng(200 + seed)
    n_feat = int(rng.integers(1, 4))
    len_all = int(rng.choice([13, 39, 7]))                 # 13 / 39: the frames-on-lanes kernels; 7: generic
    featlen = np.full(n_feat, len_all, np.int32)
    n_mgau = int(rng.integers(1, 9))
    n_den = int(rng.choice([1, 2, 5, 16, 70, 256]))
    topn = int(rng.integers(1, min(8, n_den) + 1))
    n_sen = int(rng.integers(max(2, n_mgau), 300))
    mean, var, det = _gauss(rng, n_mgau, n_feat, n_den, featlen)
    if seed % 4 == 0:
        det[...] -= 3.0e9                                    # drive distances below WORST_DIST: unfilled slots
    t = dict(n_mgau=np.array([n_mgau]), n_feat=np.array([n_feat]), n_density=np.array([n_den]),
             n_sen=np.array([n_sen]), max_topn=np.array([topn]), aw=np.array([int(rng.integers(1, 4))]),
             featlen=featlen, mean=mean, var=var, det=det,
             pdf=rng.integers(0, 256, (n_sen, n_feat, n_den)).astype(np.uint8),
             sen2mgau=rng.integers(0, n_mgau, n_sen).astype(np.uint32),
             logadd=_logadd8(), logadd_size=np.array([256]), logadd_width=np.array([1]),
             log_zero=np.array([-524288]))
    g, o = P.MsMgau(t), pso.OracleMs(t)
    feats = rng.standard_normal((40, int(featlen.sum()))).astype(np.float32)
    for step in range(40):
        call = rng.random() < 0.4
        flags = (rng.random(n_sen) < rng.uniform(0.05, 0.9)).astype(np.uint8)
        act = None if call else pso.flags2list(flags)
        a = o.frame_eval(feats[step], active=act, compallsen=call)
        b = g.frame_eval(feats[step], active=act, compallsen=call)
        assert np.array_equal(a, b), "seed %d step %d" % (seed, step)
    if seed % 4 != 0:
        # batched entry == per-frame compallsen (stateless scorer); with unfilled slots it must refuse
        got = g.score_frames(feats)
        o2 = pso.OracleMs(t)
        for i in range(feats.shape[0]):
            assert np.array_equal(got[i], o2.frame_eval(feats[i], compallsen=True)), "batch frame %d" % i
    elif topn < n_den:
        with pytest.raises(P.PsgpuError):
            g.score_frames(feats)
    g.close()


@pytest.mark.parametrize("n_emit", [3, 5])
def test_hmm_random_tables(n_emit):
    """Random transition matrices (with the 255 = no-arc floor on every arc kind),
    random sseq, random mpx / BAD_SSID patterns and scores around WORST_SCORE."""
    import ctypes as C
    import pocketsphinx_amd as P
    from test_hmm_gpu import to_recs, from_recs
    rng = np.random.default_rng(7 + n_emit)
    n_tmat, n_sseq, n_sen, n = 9, 50, 120, 4096
    tp = rng.integers(0, 80, (n_tmat, n_emit, n_emit + 1)).astype(np.uint8)
    tp[rng.random(tp.shape) < 0.3] = 255
    sseq = rng.integers(0, n_sen, (n_sseq, n_emit)).astype(np.uint16)
    ctx = P.HmmContext(tp, sseq, n_sen)
    W = -0x20000000
    for rep in range(4):
        mpx = (rng.random(n) < 0.5).astype(np.uint8)
        before = np.zeros((n, pso.HMM_FIELDS), np.int32)
        sc = -rng.integers(0, 300000, (n, 5))
        sc[rng.random((n, 5)) < 0.2] = W
        sc[rng.random((n, 5)) < 0.05] = W + rng.integers(-100, 2000)
        before[:, 0:5] = sc
        before[:, 5:10] = rng.integers(-1, 5000, (n, 5))
        before[:, 10] = W; before[:, 11] = -1
        sen = rng.integers(0, n_sen, (n, 5)); ssid = rng.integers(0, n_sseq, (n, 5))
        ssid[rng.random((n, 5)) < 0.2] = 0xffff
        ssid[:, 0] = rng.integers(0, n_sseq, n)              # state 0 always has an ssid
        before[:, 12:17] = np.where(mpx[:, None] != 0, ssid, sen)
        before[:, 17] = W; before[:, 18] = rng.integers(0, n_tmat, n)
        scr = rng.integers(0, 5000, n_sen).astype(np.int16)
        g = dict(n_emit=np.array([n_emit]), tp=tp, sseq=sseq, senscr=scr[None, :], before=before[None], mpx=mpx)
        want, ret = pso.hmm_step_oracle(g, 0)
        recs = to_recs(P, before, mpx)
        best = ctx.vit_eval(recs, scr)
        got = from_recs(recs)
        if n_emit == 3:
            for a in (got, want):
                a[:, 3:5] = 0; a[:, 8:10] = 0; a[:, 15:17] = 0
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, "HMM %d (mpx %d)\nbefore %s\noracle %s\ngpu    %s" % (
            bad[0], mpx[bad[0]], before[bad[0]], want[bad[0]], got[bad[0]])
        assert best == max(int(ret.max()), W)
    ctx.close()


@pytest.mark.parametrize("seed", range(6))
def test_ptm_random_model(seed):
    """PTM models of other shapes than en-us through the any-shape per-call kernels."""
    import pocketsphinx_amd as P
    rng = np.random.default_rng(300 + seed)
    n_mgau = int(rng.integers(1, 7))
    n_feat = int(rng.integers(1, 4))
    featlen = rng.integers(1, 15, n_feat).astype(np.int32)
    n_den = int(rng.choice([4, 9, 64, 128, 200, 256]))
    topn = int(rng.integers(1, min(8, n_den) + 1))
    n_sen = int(rng.integers(max(3, n_mgau), 500))
    mean, var, det = _gauss(rng, n_mgau, n_feat, n_den, featlen)
    if seed == 0:                                            # duplicated codewords: exact ties
        tot = int(featlen.sum())
        mm = mean.reshape(n_mgau, -1); vv = var.reshape(n_mgau, -1)
        o_ = 0
        for f in range(n_feat):
            blk = n_den * int(featlen[f]); half = (n_den // 2) * int(featlen[f])
            mm[:, o_ + half:o_ + 2 * half] = mm[:, o_:o_ + half]
            vv[:, o_ + half:o_ + 2 * half] = vv[:, o_:o_ + half]
            det[:, f, n_den // 2:2 * (n_den // 2)] = det[:, f, :n_den // 2]
            o_ += blk
    t = dict(n_mgau=np.array([n_mgau]), n_feat=np.array([n_feat]), n_density=np.array([n_den]),
             n_sen=np.array([n_sen]), max_topn=np.array([topn]), ds_ratio=np.array([int(rng.integers(1, 3))]),
             n_fast_hist=np.array([int(rng.integers(2, 8))]), featlen=featlen, mean=mean, var=var, det=det,
             mixw=rng.integers(0, 160, (n_feat, n_den, n_sen)).astype(np.uint8),
             sen2cb=np.sort(rng.integers(0, n_mgau, n_sen)).astype(np.uint8), logadd8=_logadd8())
    H = int(t["n_fast_hist"][0])
    m = P.PtmModel(t)
    st = P.PtmState(m, H)
    o = pso.OraclePTM(t, n_fast_hist=H)
    fi = 0
    for step in range(50):
        x = rng.standard_normal(int(featlen.sum())).astype(np.float32)
        frame = fi if rng.random() < 0.8 else max(0, fi - int(rng.integers(1, 3)))
        call = rng.random() < 0.3
        flags = (rng.random(n_sen) < rng.uniform(0.02, 0.9)).astype(np.uint8)
        act = None if call else pso.flags2list(flags)
        o.set_frame_idx(fi)
        a = o.frame_eval(x, frame, active=act, compallsen=call)
        b = st.frame_eval(x, frame, active=act, compallsen=call, frame_idx=fi)
        assert np.array_equal(a, b), "seed %d step %d (frame %d/%d topn %d n_den %d)" % (seed, step, frame, fi, topn, n_den)
        if frame == fi:
            fi += 1
    st.close(); m.close()


@pytest.mark.parametrize("L,n_den,topn,aw,T", [(39, 8, 4, 1, 150), (13, 16, 2, 3, 70), (39, 4, 3, 1, 64)])
def test_ms_continuous_model_batch(L, n_den, topn, aw, T):
    """A fully continuous model (.cont. mapping: every senone its own codebook,
    ms_senone.c:305-315): 700 senones, one stream -- the fused frames-on-lanes kernel
    (densities -> top-N -> senone score without the lists in between), with the list
    buffers (host wrapper) and without them (device entry, NULL lists); senone and
    frame counts that are not multiples of the 64 x 64 tiles."""
    import ctypes as C
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import capi
    rng = np.random.default_rng(42 + L + topn)
    n_sen = 700
    featlen = np.array([L], np.int32)
    mean, var, det = _gauss(rng, n_sen, 1, n_den, featlen)
    t = dict(n_mgau=np.array([n_sen]), n_feat=np.array([1]), n_density=np.array([n_den]),
             n_sen=np.array([n_sen]), max_topn=np.array([topn]), aw=np.array([aw]), featlen=featlen,
             mean=mean, var=var, det=det, pdf=rng.integers(0, 256, (n_sen, 1, n_den)).astype(np.uint8),
             sen2mgau=np.arange(n_sen, dtype=np.uint32), logadd=_logadd8(), logadd_size=np.array([256]),
             logadd_width=np.array([1]), log_zero=np.array([-524288]))
    g, o = P.MsMgau(t), pso.OracleMs(t)
    feats = rng.standard_normal((T, L)).astype(np.float32)
    got = g.score_frames(feats)
    for i in range(feats.shape[0]):
        assert np.array_equal(got[i], o.frame_eval(feats[i], compallsen=True)), "frame %d" % i
    dev = torch.device("cuda", 0)
    f = torch.from_numpy(feats).to(dev)
    scr = torch.zeros((T, n_sen), dtype=torch.int16, device=dev)
    L_ = capi.lib()
    capi.check(L_.psgpu_ms_score_batch_dev(g.h, C.c_void_p(f.data_ptr()), T, None, None, C.c_void_p(scr.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), "no lists")
    capi.check(L_.psgpu_ms_batch_check(g.h, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "check")
    assert np.array_equal(scr.cpu().numpy(), got)
    g.close()


def test_ms_shared_codebooks_need_lists():
    """The list-less call is only for fully continuous models; any other shape says so."""
    import ctypes as C
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import capi
    rng = np.random.default_rng(5)
    featlen = np.array([13], np.int32)
    mean, var, det = _gauss(rng, 4, 1, 8, featlen)
    t = dict(n_mgau=np.array([4]), n_feat=np.array([1]), n_density=np.array([8]), n_sen=np.array([40]),
             max_topn=np.array([4]), aw=np.array([1]), featlen=featlen, mean=mean, var=var, det=det,
             pdf=rng.integers(0, 256, (40, 1, 8)).astype(np.uint8),
             sen2mgau=(np.arange(40) % 4).astype(np.uint32), logadd=_logadd8(), logadd_size=np.array([256]),
             logadd_width=np.array([1]), log_zero=np.array([-524288]))
    g = P.MsMgau(t)
    dev = torch.device("cuda", 0)
    f = torch.zeros((8, 13), dtype=torch.float32, device=dev)
    scr = torch.zeros((8, 40), dtype=torch.int16, device=dev)
    rc = capi.lib().psgpu_ms_score_batch_dev(g.h, C.c_void_p(f.data_ptr()), 8, None, None, C.c_void_p(scr.data_ptr()), None)
    assert rc == -2
    g.close()
