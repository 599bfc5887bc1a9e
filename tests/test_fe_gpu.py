"""GPU parity of the MFCC front end (psgpu_fe_*, the fe_process_utt + fe_end_utt
replacement) against the pinned oracle and the reference's own cepstra.

Tolerance: the path is float64/float32 arithmetic restated operation for operation
(bit-identical by construction) except log(), where the device's libm and the host's
are both faithful, not correctly rounded.  The tests demand bit equality and would
report the number of differing values; on every bundled recording there is none."""
import numpy as np
import pytest

import pso
from test_oracle_golden import _load, MFCC_CASES

pytestmark = pytest.mark.gpu


def _same(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))
    assert bad[0].size == 0, "%d of %d values differ; first: frame %d coeff %d: %r vs %r" % (
        bad[0].size, a.size, bad[0][0], bad[1][0], a[bad[0][0], bad[1][0]], b[bad[0][0], bad[1][0]])


@pytest.mark.parametrize("case", MFCC_CASES)
def test_fe_matches_reference_and_oracle(case):
    """Every configuration of the goldens (dct / legacy / htk transforms, raw and
    smoothed log spectra, DC removal, noise removal on and off, 1024-point FFT
    without pre-emphasis, an utterance shorter than a frame, one with no left-over
    samples): first from reset noise statistics, then with the tracker carried."""
    import pocketsphinx_amd as P
    g = _load("mfcc_%s.npz" % case)
    fe = P.FrontEnd(g)
    o = pso.OracleFe(g)
    noise = np.zeros((1, 4, fe.n_filt), np.float64)
    undefined = np.ones(1, np.int32)
    for key in ("cep", "cep1"):
        cep, fo = fe.process_utts([g["pcm"]], noise, undefined)
        assert fo.tolist() == [0, g[key].shape[0]]
        _same(cep, o.process(g["pcm"]))
        _same(cep, g[key])
        assert undefined[0] == 0 or not int(g["par"][10])
        if int(g["par"][10]):
            assert np.array_equal(noise[0], o.noise)          # the tracker state itself, float64
    fe.close()


def test_fe_ragged_batch():
    """Several utterances in one call (one shorter than a frame, one empty, one
    ending exactly on a frame boundary) equal the same utterances one at a time;
    without noise arrays every utterance starts from reset statistics."""
    import pocketsphinx_amd as P
    g = _load("mfcc_en_us_goforward.npz")
    pcm = g["pcm"]
    cuts = [pcm[:7000], pcm[100:300], pcm[:0], pcm[9000:9000 + 410 + 160 * 20], pcm[20000:], pcm[3:411]]
    fe = P.FrontEnd(g)
    cep, fo = fe.process_utts(cuts)
    assert fo[-1] == cep.shape[0] and fo[3] == fo[2]
    for u, c in enumerate(cuts):
        o = pso.OracleFe(g)
        _same(cep[fo[u]:fo[u + 1]], o.process(c))
    fe.close()


def test_fe_all_bundled_recordings_and_feature_chain(tables):
    """en-us configuration on every bundled 16 kHz recording, then the chain the
    decoder runs: PCM -> cepstra -> 1s_c_d_dd features on the device equals the
    feature vectors of the reference decoder's own acmod buffer."""
    import os
    import pocketsphinx_amd as P
    g = _load("mfcc_en_us_goforward.npz")
    fe = P.FrontEnd(g)
    data = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "data")
    names = [n for n in ("goforward.raw", "numbers.raw", "something.raw", "librivox-0870.raw")
             if os.path.exists(os.path.join(data, n))]
    assert names, "staged recordings missing (make -C oracle)"
    pcms = [np.fromfile(os.path.join(data, n), dtype=np.int16) for n in names]
    cep, fo = fe.process_utts(pcms)
    for u, p in enumerate(pcms):
        _same(cep[fo[u]:fo[u + 1]], pso.OracleFe(g).process(p))
    c0 = cep[fo[0]:fo[1]]
    feat = P.dynfeat_1s_c_d_dd(c0, [c0.shape[0]])
    assert feat.tobytes() == np.ascontiguousarray(_load("ptm_goforward.npz")["feat"], np.float32).tobytes()
    fe.close()


def test_fe_dither_draws_the_references_sequence(tmp_path):
    """-dither yes: every sample the front end consumes gets (s3_rand_int31() % 4 == 0) added (fe_sigproc.c:868-870, :898-901), from
    ONE Mersenne-Twister stream seeded at fe_init with -seed and never again.  The compiled reference runs goforward.raw three
    times through one fe_t (ref_dump mfcc, -seed 17 and the default -1): three DIFFERENT sets of cepstra.  The device front end
    reproduces all three bit for bit, as three calls and as one call of three utterances; and differs from the undithered
    cepstra (so the bits were applied)."""
    import os
    import subprocess
    import sys
    import pocketsphinx_amd as P
    exe = os.path.join(pso.REF_DIR, "ref_dump")
    if not os.path.exists(exe):
        pytest.fail("oracle/_ref (compiled reference + staged data) not built: run __graft_entry__.build() where /root/reference is present")
    sys.path.insert(0, os.path.join(os.path.dirname(pso.__file__), "..", "oracle"))
    from psgb import read_psgb
    plain = _load("mfcc_en_us_goforward.npz")
    for seed in ("17", "-1"):
        out = os.path.join(str(tmp_path), "dither_%s.psgb" % seed.replace("-", "m"))
        subprocess.check_call([exe, "mfcc", out, os.path.join(pso.REF_DIR, "model", "an4_ci_cont"), "-", "-", os.path.join(pso.REF_DIR, "data", "goforward.raw"),
                               "3", "--", "dither", "yes", "seed", seed, "remove_noise", "no"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                              timeout=300)
        g = read_psgb(out)
        assert int(g["par"][13]) == 1 and int(g["dither_seed"][0]) == int(seed)
        want = [g["cep"], g["cep1"], g["cep2"]]
        assert not np.array_equal(want[0], want[1]) and not np.array_equal(want[1], want[2])
        fe = P.FrontEnd(g)
        for r in range(3):
            cep, fo = fe.process_utts([g["pcm"]])
            _same(cep, want[r])
        fe.close()
        fe = P.FrontEnd(g)
        cep, fo = fe.process_utts([g["pcm"]] * 3)
        for r in range(3):
            _same(cep[fo[r]:fo[r + 1]], want[r])
        fe.close()
        # the dither bits were applied: same shape as the undithered cepstra of the same recording, other values
        assert want[0].shape[1] == plain["cep"].shape[1]
    und = os.path.join(str(tmp_path), "undithered.psgb")
    subprocess.check_call([exe, "mfcc", und, os.path.join(pso.REF_DIR, "model", "an4_ci_cont"), "-", "-", os.path.join(pso.REF_DIR, "data", "goforward.raw"),
                           "1", "--", "dither", "no", "remove_noise", "no"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    gu = read_psgb(und)
    assert gu["cep"].shape == want[0].shape and not np.array_equal(gu["cep"], want[0])


def test_device_log_equals_the_hosts_libm_over_the_mel_range():
    """The front end's one libm call, log(mel spectrum + 1e-4) in double precision (fe_sigproc.c:1215-1228), is the only
    operation of the path that is not bit-identical by construction: the device's log and glibc's are both faithful, neither
    is correctly rounded.  Pinned by measurement: 2^24 arguments spread log-uniformly over the value range of a mel
    spectrum of 16-bit audio (1e-4, the floor, to 1e13), plus arguments straddling every power of two in it (where an
    argument reduction changes branch), plus the neighbourhood of 1 -- every result equals libm's."""
    import ctypes as C
    import torch
    from pocketsphinx_amd import capi
    rng = np.random.default_rng(2026)
    x = np.exp(rng.uniform(np.log(1e-4), np.log(1e13), 1 << 24))
    edges = np.concatenate([np.nextafter(2.0 ** k, b) * np.ones(1) for k in range(-14, 45) for b in (0.0, np.inf)] +
                           [2.0 ** np.arange(-14, 45)])
    near1 = 1.0 + np.concatenate([np.linspace(-1e-3, 1e-3, 20001), rng.uniform(-0.3, 0.3, 100000)])
    x = np.ascontiguousarray(np.concatenate([x, edges, near1, np.array([1e-4, 1.0001e-4])]), np.float64)
    want = np.empty_like(x)
    L = pso.lib()
    L.pso_libm_log.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.pso_libm_log(x.ctypes.data_as(C.c_void_p), x.size, want.ctypes.data_as(C.c_void_p))
    d_x = torch.from_numpy(x).cuda()
    d_o = torch.empty_like(d_x)
    capi.check(capi.lib().psgpu_fe_log_dev(C.c_void_p(d_x.data_ptr()), C.c_int64(x.size), C.c_void_p(d_o.data_ptr()), None),
               "psgpu_fe_log_dev")
    torch.cuda.synchronize()
    got = d_o.cpu().numpy()
    bad = np.nonzero(got.view(np.int64) != want.view(np.int64))[0]
    assert bad.size == 0, "%d of %d arguments differ, first x = %r: device %r libm %r" % (
        bad.size, x.size, x[bad[0]], got[bad[0]], want[bad[0]])
