"""Pin oracle/ps_oracle.c (the CPU checker) to goldens produced by the
unmodified reference (oracle/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import pso

G = pso.GOLDEN_DIR


def _load(name):
    z = np.load(os.path.join(G, name))
    return {k: z[k] for k in z.files}


def dup_tables(t):
    """Same in-memory edit as ref_dump.c dup_codewords()."""
    t = dict(t)
    n_mgau, n_feat, n_den = int(t["n_mgau"][0]), int(t["n_feat"][0]), int(t["n_density"][0])
    fl = int(t["featlen"][0])
    half = n_den // 2
    for k in ("mean", "var"):
        a = t[k].reshape(n_mgau, n_feat, n_den, fl).copy()
        a[:, :, half:] = a[:, :, :half]
        t[k] = a.reshape(-1)
    d = t["det"].copy()
    d[:, :, half:] = d[:, :, :half]
    t["det"] = d
    return t


def run_case(o, g):
    feats, seglen, carry = g["feat"], int(g["seglen"]), int(g["carry"])
    T = feats.shape[0]
    scr = np.empty((T, o.n_sen), np.int16)
    cw = np.empty((T, o.n_mgau, o.n_feat, o.topn), np.uint8)
    raw = np.empty((T, o.n_mgau, o.n_feat, o.topn), np.int32)
    o.reset_hist()
    for s0 in range(0, T, seglen):
        a, b, c = o.score_utt(feats[s0:s0 + seglen], reset_hist=not carry)
        scr[s0:s0 + seglen], cw[s0:s0 + seglen], raw[s0:s0 + seglen] = a, b, c
    return scr, cw, raw


CASES = ["goforward", "goforward_x2_carry", "synth", "synth_utts", "adversarial", "dup_ties"]


@pytest.mark.parametrize("case", CASES)
def test_ptm_oracle_matches_reference(tables, case):
    g = _load("ptm_%s.npz" % case)
    t = dup_tables(tables) if int(g["dup"]) else tables
    o = pso.OraclePTM(t)
    scr, cw, raw = run_case(o, g)
    T = scr.shape[0]
    idx = g["sample_idx"]
    assert np.array_equal(cw[idx], g["topn_cw_sample"])
    assert np.array_equal(raw[idx], g["topn_raw_sample"])
    assert np.array_equal(scr[idx], g["senscr_sample"])
    topn = np.concatenate([cw.reshape(T, -1).astype(np.int32), raw.reshape(T, -1)], axis=1)
    assert np.array_equal(pso.row_hash(topn), g["topn_hash"])
    assert np.array_equal(pso.row_hash(scr), g["senscr_hash"])
    if "topn_cw" in g:
        assert np.array_equal(cw, g["topn_cw"])
        assert np.array_equal(raw, g["topn_raw"])


def test_ptm_oracle_clustered_4bit_sendump(tables):
    """ptm_mgau.c:375-379 on a 4-bit clustered sendump (the en-us weights re-quantised by oracle/make_golden.py ptm4 and
    scored by the unmodified reference), nibble quirk included; and the expansion the device model is built from
    (pocketsphinx_amd.ptm.expand_clustered_mixw) gives the same scores through the plain 8-bit path."""
    from pocketsphinx_amd.ptm import expand_clustered_mixw
    g = _load("ptm_4bit_goforward.npz")
    t4 = pso.clustered_tables(tables, g)
    scr, cw, raw = run_case(pso.OraclePTM(t4), g)
    idx = g["sample_idx"]
    assert np.array_equal(cw[idx], g["topn_cw_sample"])
    assert np.array_equal(scr[idx], g["senscr_sample"])
    assert np.array_equal(pso.row_hash(scr), g["senscr_hash"])
    t8 = dict(tables)
    t8["mixw"] = expand_clustered_mixw(g["mixw4"], g["mixw_cb"], int(tables["n_sen"][0]))
    scr8, _, _ = run_case(pso.OraclePTM(t8), g)
    assert np.array_equal(scr8, scr)
    # the quirk: both senones of a byte get the same weight
    assert np.array_equal(t8["mixw"][..., 0:-1:2], t8["mixw"][..., 1::2])


def test_history_dependence_is_real(tables):
    """a stateless top-N differs from the reference's stateful one
    somewhere (otherwise the carry fixtures pin nothing)."""
    g = _load("ptm_goforward_x2_carry.npz")
    o = pso.OraclePTM(tables)
    n = int(g["seglen"])
    a, _, _ = o.score_utt(g["feat"][:n], reset_hist=True)
    b, _, _ = o.score_utt(g["feat"][n:], reset_hist=False)
    # same features, different seed state: almost all frames equal
    assert a.shape == b.shape
    assert (a != b).any(axis=1).sum() < n // 4


@pytest.mark.parametrize("case", ["default", "fwdtree_only", "ptm_topn2", "ptm_topn6_ds2"])
def test_senlog_replay(tables, case):
    """Replay every frame_eval call the reference made during a real decode
    (active lists, history-slot reuse, pass-2 codebook masking)."""
    g = _load("senlog_%s.npz" % case)
    feats = g["call_feat"]      # the vector the reference handed to each call
    pr = pso.senlog_params(g)
    o = pso.OraclePTM(tables, topn=int(pr["topn"]) if "topn" in pr else None,
                      ds_ratio=int(pr["ds"]) if "ds" in pr else None)
    n = int(g["call_frame"].size)
    off = g["call_act_off"]
    hashes = np.empty(n, np.uint64)
    rows = {}
    sample = set(int(i) for i in g["sample_idx"])
    for c in range(n):
        fr, na = int(g["call_frame"][c]), int(g["call_nact"][c])
        o.set_frame_idx(int(g["call_frame_idx"][c]))
        act = None if na < 0 else g["call_act"][off[c]:off[c] + na]
        scr = o.frame_eval(feats[c], fr, active=act, compallsen=(na < 0))
        hashes[c] = pso.row_hash(scr[None, :])[0]
        if c in sample:
            rows[c] = scr
    bad = np.nonzero(hashes != g["call_scr_hash"])[0]
    assert bad.size == 0, "first mismatching call %d (frame %d)" % (bad[0], g["call_frame"][bad[0]])
    for k, c in enumerate(g["sample_idx"]):
        assert np.array_equal(rows[int(c)], g["call_scr_sample"][k])


def test_flags2list_bridges_gaps():
    flags = np.zeros(5126, np.uint8)
    flags[[0, 3, 300, 301, 1000, 5125]] = 1
    d = pso.flags2list(flags)
    sens = np.cumsum(d.astype(np.int64))
    # every flagged senone is listed; bridging entries are extra senones
    assert set([0, 3, 300, 301, 1000, 5125]) <= set(sens.tolist())
    assert d.max() <= 255 and d[0] == 0
    # acmod.c:1246-1249: a gap of 297 becomes 255 + 42
    assert list(d[:4]) == [0, 3, 255, 42]


@pytest.mark.parametrize("case", ["en_us_3st", "tidigits_5st", "syn_4st", "syn_2st", "syn_1st"])
def test_hmm_oracle_matches_reference(case):
    """pso_hmm_vit_eval vs the reference's hmm_vit_eval (hmm.c:786-805) on the
    state dumps of oracle/ref_dump.c `hmm`: 3-state (en-us) and 5-state
    (tidigits) topologies, multiplex and not, incl. WORST_SCORE clamps,
    BAD_SSID states and saturated senone scores; `hmmsyn`: synthetic contexts
    with 4, 2 and 1 emitting states, which the reference evaluates with
    hmm_vit_eval_anytopo (hmm.c:710-784)."""
    g = _load("hmm_%s.npz" % case)
    for t in range(g["before"].shape[0]):
        after, ret = pso.hmm_step_oracle(g, t)
        bad = np.nonzero((after != g["after"][t]).any(axis=1) | (ret != g["ret"][t]))[0]
        assert bad.size == 0, "step %d: first mismatching HMM %d (mpx %d)\nbefore %s\nref    %s\noracle %s" % (
            t, bad[0], g["mpx"][bad[0]], g["before"][t][bad[0]], g["after"][t][bad[0]], after[bad[0]])


SEMI_CASES = ["tidigits_default", "tidigits_beam", "tidigits_topn6_ds2", "tidigits_topn7_call", "tidigits_topn2",
              "tidigits_topn8_list"]


def semi_oracle_for(g, t):
    p = pso.senlog_params(g)
    beam = None
    if "topn_beam" in p:
        b = [int(x) for x in p["topn_beam"].split(",")]
        beam = (b + [max(b)] * 4)[:int(t["n_feat"][0])]
    return pso.OracleSemi(t, topn=int(p["topn"]) if "topn" in p else None,
                          ds_ratio=int(p["ds"]) if "ds" in p else None, topn_beam=beam)


@pytest.mark.parametrize("case", SEMI_CASES)
def test_semi_senlog_replay(case):
    """pso_semi_frame_eval vs every s2_semi_mgau_frame_eval call of real tidigits
    decodes (4 streams x 256 densities, 4-bit clustered weights): default, per-stream
    top-N beams, topn 6 + ds 2, topn 7 (the `_any` kernels) with compallsen, topn 2."""
    g = _load("senlog_%s.npz" % case)
    t = _load("semi_tidigits_tables.npz")
    o = semi_oracle_for(g, t)
    off = g["call_act_off"]
    n = int(g["call_frame"].size)
    hashes = np.empty(n, np.uint64)
    for c in range(n):
        na = int(g["call_nact"][c])
        o.set_frame_idx(int(g["call_frame_idx"][c]))
        act = None if na < 0 else g["call_act"][off[c]:off[c] + na]
        scr = o.frame_eval(g["call_feat"][c], int(g["call_frame"][c]), active=act, compallsen=(na < 0))
        hashes[c] = pso.row_hash(scr[None, :])[0]
    bad = np.nonzero(hashes != g["call_scr_hash"])[0]
    assert bad.size == 0, "first mismatching call %d (frame %d)" % (bad[0], g["call_frame"][bad[0]])


MS_CASES = [("ms_an4_default", "ms_an4_tables"), ("ms_an4_compall_aw2", "ms_an4_tables"),
            ("ms_en_us_default", "ms_en_us_tables"), ("ms_en_us_topn2_call", "ms_en_us_tables")]


@pytest.mark.parametrize("case,tab", MS_CASES)
def test_ms_senlog_replay(case, tab):
    """pso_ms_frame_eval vs every ms_cont_mgau_frame_eval call of real decodes:
    an4_ci_cont (1 density/codebook: compute_dist_all) and en-us forced through the
    ms scorer (42 codebooks x 3 streams x 128 densities: the top-N scan), incl.
    aw 2, topn 2, compallsen.  The score buffer persists between calls (unlisted
    senones keep stale values in the reference)."""
    g = _load("senlog_%s.npz" % case)
    t = _load("%s.npz" % tab)
    p = pso.senlog_params(g)
    o = pso.OracleMs(t, topn=int(p["topn"]) if "topn" in p else None, aw=int(p["aw"]) if "aw" in p else None)
    off = g["call_act_off"]
    n = int(g["call_frame"].size)
    hashes = np.empty(n, np.uint64)
    for c in range(n):
        na = int(g["call_nact"][c])
        act = None if na < 0 else g["call_act"][off[c]:off[c] + na]
        scr = o.frame_eval(g["call_feat"][c], active=act, compallsen=(na < 0))
        hashes[c] = pso.row_hash(scr[None, :])[0]
    bad = np.nonzero(hashes != g["call_scr_hash"])[0]
    assert bad.size == 0, "first mismatching call %d (frame %d)" % (bad[0], g["call_frame"][bad[0]])


def test_dynfeat_oracle_matches_reference():
    """pso_dynfeat_1s_c_d_dd vs feat_s2mfc2feat_live(begin, end) of the reference on the
    bundled cepstra test/data/goforward.mfc (batch CMN, padding, deltas): memcmp."""
    import ctypes as C
    g = _load("dynfeat_goforward.npz")
    cep = np.ascontiguousarray(g["cep"], np.float32)
    out = np.empty((cep.shape[0], 3 * cep.shape[1]), np.float32)
    L = pso.lib()
    L.pso_dynfeat_1s_c_d_dd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.pso_dynfeat_1s_c_d_dd(cep.ctypes.data, cep.shape[0], cep.shape[1], out.ctypes.data)
    assert out.tobytes() == np.ascontiguousarray(g["feat"], np.float32).tobytes()


MFCC_CASES = ["en_us_goforward", "legacy_dc", "htk_40", "logspec", "smoothspec", "nfft1024", "short", "exact"]


@pytest.mark.parametrize("case", MFCC_CASES)
def test_fe_oracle_matches_reference(case):
    """pso_fe_process_utt vs the reference front end (fe_start_utt / fe_process_frames /
    fe_end_utt on its own fe_t, ref_dump mfcc): cepstra memcmp, first from reset noise
    statistics, then a second pass with the noise tracker carried over.  The only libm
    call on the path is log(); oracle and reference use the same libm here."""
    g = _load("mfcc_%s.npz" % case)
    fe = pso.OracleFe(g)
    for key in ("cep", "cep1"):
        got = fe.process(g["pcm"])
        ref = np.ascontiguousarray(g[key], np.float32)
        assert got.shape == ref.shape
        bad = np.nonzero(got.view(np.uint32) != ref.view(np.uint32))
        assert bad[0].size == 0, "%s: %d values differ, first at frame %d coeff %d" % (
            key, bad[0].size, bad[0][0], bad[1][0])


def test_fe_plus_dynfeat_oracle_reproduces_decoder_features():
    """PCM -> pso_fe_process_utt -> pso_dynfeat_1s_c_d_dd equals, bit for bit, the feature
    vectors the reference decoder itself computed for goforward.raw (ref_dump feats: the
    contents of acmod->feat_buf during a real decode)."""
    import ctypes as C
    g = _load("mfcc_en_us_goforward.npz")
    cep = pso.OracleFe(g).process(g["pcm"])
    out = np.empty((cep.shape[0], 3 * cep.shape[1]), np.float32)
    L = pso.lib()
    L.pso_dynfeat_1s_c_d_dd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.pso_dynfeat_1s_c_d_dd(cep.ctypes.data, cep.shape[0], cep.shape[1], out.ctypes.data)
    assert out.tobytes() == np.ascontiguousarray(_load("ptm_goforward.npz")["feat"], np.float32).tobytes()


def test_dither_sequence_is_one_mersenne_twister_stream(tmp_path):
    """-dither yes (fe_sigproc.c:868-870, :898-901): every sample a frame reads in gets (s3_rand_int31() % 4 == 0) added in int16
    arithmetic -- s3_rand_int31 = MT19937's genrand_int32() >> 1, seeded once by fe_init_dither(-seed) with init_genrand(seed &
    0xffffffff) -- the stream running on from one utterance to the next.  A full frame reads frame_size (the first) or frame_shift
    NEW samples; the tail frame of fe_end_utt (fe_interface.c:529-546) reads the whole overflow buffer -- the last frame_size -
    frame_shift samples over again, from the UNdithered copies kept there, plus what was left -- and draws for all of them anew.
    Pinned here: the oracle front end on PCM dithered that way (numpy's legacy MT19937 seeding is init_genrand) equals the compiled
    reference's cepstra of three runs over one fe_t, for -seed 17 and the default -1.  The device front end draws the same stream
    (csrc/psgpu_fe.hip FeRand; tests/test_fe_gpu.py)."""
    import os
    import subprocess
    import sys
    exe = os.path.join(pso.REF_DIR, "ref_dump")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref (compiled reference + staged data) not built")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    from psgb import read_psgb
    for seed in (17, -1):
        out = os.path.join(str(tmp_path), "d%d.psgb" % (seed & 0xff))
        subprocess.check_call([exe, "mfcc", out, os.path.join(pso.REF_DIR, "model", "an4_ci_cont"), "-", "-", os.path.join(pso.REF_DIR, "data", "goforward.raw"),
                               "3", "--", "dither", "yes", "seed", str(seed), "remove_noise", "no"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL, timeout=300)
        g = read_psgb(out)
        rs = np.random.RandomState(seed & 0xffffffff)
        fe = pso.OracleFe(g)
        pcm = np.ascontiguousarray(g["pcm"], np.int16)
        fs, sh = int(g["par"][0]), int(g["par"][1])
        n_full = 1 + (pcm.size - fs) // sh
        n_reg, tail0 = fs + (n_full - 1) * sh, n_full * sh

        def draws(n):
            raw = rs.randint(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)             # genrand_int32
            return (((raw >> 1) & 3) == 0).astype(np.int32)
        for key in ("cep", "cep1", "cep2"):
            reg = pcm.astype(np.int32)
            reg[:n_reg] += draws(n_reg)
            tail = reg.copy()
            tail[tail0:] = pcm[tail0:].astype(np.int32) + draws(pcm.size - tail0)
            got = fe.process(reg.astype(np.int16))                                             # (wraps like the int16 +=)
            got[-1] = fe.process(tail.astype(np.int16))[-1]
            ref = np.ascontiguousarray(g[key], np.float32)
            assert got.shape == ref.shape and got.shape[0] == n_full + 1
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (seed, key)
