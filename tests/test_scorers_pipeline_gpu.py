"""GPU parity of the device pipeline (psgpu_decode_*) with the two scorers that are not PTM -- BASELINE configs[3]:
the multi-stream / continuous scorer (ms_cont_mgau_frame_eval, reference src/ms_mgau.c:192-282: any model decoded with
-senmgau, and every model without a sendump) and the semi-continuous one (s2_semi_mgau_frame_eval,
src/s2_semi_mgau.c:837-883: tidigits), which acmod_init_am (src/acmod.c:62-130) picks instead of ptm_mgau.

Checker: the compiled reference decoding the same input on the host -- ref_decode_bench (PCM) for the ms models,
`ref_dump fwdtree` (cepstra files, one new decoder per utterance) for tidigits -- words, frames, scores, table sizes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import pso
from test_oracle_golden import _load

pytestmark = pytest.mark.gpu
REF = pso.REF_DIR


def _need_ref():
    if not os.path.exists(os.path.join(REF, "ref_dump")) or not os.path.exists(os.path.join(REF, "ref_decode_bench")):
        pytest.fail("oracle/_ref (compiled reference + staged data) not built: run __graft_entry__.build() where /root/reference is present")


def _ref_dump(tmp_path, cmd, model, lm, dic, args, extra=()):
    sys.path.insert(0, os.path.join(os.path.dirname(pso.__file__), "..", "oracle"))
    from psgb import read_psgb
    out = os.path.join(str(tmp_path), "%s_%d.psgb" % (cmd, len(os.listdir(str(tmp_path)))))
    argv = [os.path.join(REF, "ref_dump"), cmd, out, os.path.join(REF, "model", model), lm or "-", dic or "-"] + [str(a) for a in args]
    if extra:
        argv += ["--"] + list(extra)
    subprocess.check_call(argv, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    return read_psgb(out)


def _ref_decode(tmp_path, model, pcms, extra):
    raw = tmp_path / "utts.raw"
    np.concatenate(pcms).tofile(raw)
    data = os.path.join(REF, "data")
    argv = [os.path.join(REF, "ref_decode_bench"), os.path.join(REF, "model", model), os.path.join(data, "turtle.lm.bin"),
            os.path.join(data, "turtle.dic"), str(raw), str(pcms[0].size)]
    if extra:
        argv += ["--"] + list(extra)
    out = subprocess.run(argv, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return [json.loads(ln) for ln in out.stdout.strip().splitlines() if ln.startswith("{")][:-1]


def _same(u, r, hn, hyp, res, what):
    assert int(res[u, 3]) == 0 and int(res[u, 2]) == r["frames"], (what, res[u], r["frames"])
    assert int(res[u, 0]) == r["n_bp"] and int(res[u, 1]) == r["n_bss"], (what, res[u], r["n_bp"], r["n_bss"])
    got = [tuple(int(v) for v in hyp[u, i, :3]) for i in range(int(hn[u, 0]))]
    want = [(s[1], s[2], s[3]) for s in r["seg"]]
    assert got == want, "%s: %r vs %r" % (what, got[:6], want[:6])
    assert int(hn[u, 1]) == r["score"], what


def test_pipeline_with_the_ms_scorer_en_us(tmp_path):
    """en-us forced through the multi-stream scorer (-senmgau .ptm.: 42 codebooks x 3 streams x 128 densities, top-4 lists by
    full scan, 16-bit log-add): PCM -> hypotheses on the device, against the reference decoding the same PCM with that scorer"""
    _need_ref()
    import pocketsphinx_amd as P
    from pocketsphinx_amd import synth
    pcms = [synth.utterance(i, 6.0) for i in (2, 4, 9)]
    refs = _ref_decode(tmp_path, "en-us-ms", pcms, ("senmgau", ".ptm."))
    gt = _load("fwdtree_trace_goforward.npz")
    ms = P.MsMgau(_load("ms_en_us_tables.npz"))
    p = P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), None, _load("fwdtree_static_en_us_turtle.npz"), gt["par"], gt, scorer=ms)
    p.run(pcms)
    hn, hyp, res = p.fetch()
    for u, r in enumerate(refs):
        _same(u, r, hn, hyp, res, "en-us-ms utterance %d" % u)
    p.session(True); p.session(False)                     # (the ms scorer has no history: a session carries the search's state only)
    p.close()


def test_pipeline_with_a_continuous_model_of_en_us_size(tmp_path):
    """BASELINE configs[3] as written: a continuous-density model at scale -- 5126 senones x 16 densities x 39 dimensions, one
    codebook a senone (`.cont.` mixtures; staged by oracle/stage_cont_model.py from en-us's own parameters, the scorer's tables
    exported by integration/psgpu_export_tables from a decoder the reference initialised with it) -- PCM -> hypotheses through the
    device pipeline (the fused continuous kernel of csrc/psgpu_ms.hip), against the reference decoding the same PCM with that
    model directory"""
    _need_ref()
    import pocketsphinx_amd as P
    from pocketsphinx_amd import synth
    from pocketsphinx_amd.tablefile import read_psgb
    path = os.path.join(os.path.dirname(pso.__file__), "..", "integration", "_tables", "en_us_cont_turtle.psgb")
    if not os.path.exists(path) or not os.path.exists(os.path.join(REF, "model", "en-us-cont", "means")):
        pytest.fail("integration/_tables/en_us_cont_turtle.psgb / oracle/_ref/model/en-us-cont not built (__graft_entry__.build())")
    cg = read_psgb(path)
    ct = {k[3:]: v for k, v in cg.items() if k.startswith("ms_")}
    ct["sen2mgau"] = ct["sen2mgau"].astype(np.uint32)
    assert int(ct["n_mgau"][0]) == int(ct["n_sen"][0]) == 5126 and int(ct["n_density"][0]) == 16 and int(ct["featlen"][0]) == 39
    pcms = [synth.utterance(i, 6.0) for i in (2, 4, 9, 11)]
    refs = _ref_decode(tmp_path, "en-us-cont", pcms, ())
    gt = _load("fwdtree_trace_goforward.npz")
    ms = P.MsMgau(ct)
    p = P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), None, _load("fwdtree_static_en_us_turtle.npz"), gt["par"], gt, scorer=ms)
    p.run(pcms)
    hn, hyp, res = p.fetch()
    for u, r in enumerate(refs):
        _same(u, r, hn, hyp, res, "en-us-cont utterance %d" % u)
    p.close()


def test_pipeline_with_the_ms_scorer_an4_continuous(tmp_path):
    """an4_ci_cont (the reference's test_mllr model: 102 codebooks x 1 density x 39 dims, senone i owns codebook i; 40 mel
    filters, other band edges; CI phones only, so part of the dictionary is dropped): every table of this configuration read
    out of a decoder the reference initialised, the decode on the device"""
    _need_ref()
    import pocketsphinx_amd as P
    from pocketsphinx_amd import synth
    data = os.path.join(REF, "data")
    lm, dic = os.path.join(data, "turtle.lm.bin"), os.path.join(data, "turtle.dic")
    x = ("dither", "no")
    g = _ref_dump(tmp_path, "fwdtree", "an4_ci_cont", lm, dic, [os.path.join(data, "goforward.raw")], ("fwdflat", "no", "bestpath", "no") + x)
    fe = _ref_dump(tmp_path, "mfcc", "an4_ci_cont", lm, dic, [os.path.join(data, "goforward.raw"), 1], x)
    clips = _load("speech_clips.npz")
    pcms = [synth.utterance(i, 6.0) for i in (2, 4)]
    refs = _ref_decode(tmp_path, "an4_ci_cont", pcms, x)
    ms = P.MsMgau(_load("ms_an4_tables.npz"))
    p = P.DecodePipeline(fe, None, g, g["par"], g, scorer=ms)
    p.run(pcms)
    hn, hyp, res = p.fetch()
    for u, r in enumerate(refs):
        _same(u, r, hn, hyp, res, "an4 utterance %d" % u)
    # the dump's own recording (first pass alone: the reference's "go forward ten meters" for this model, SURVEY 8c (5), comes
    # out of its later passes): identical tables
    p.run([clips["goforward"]])
    hn, hyp, res = p.fetch()
    tab = p.tables(0, res)
    assert np.array_equal(tab["bp"], g["bp"]) and np.array_equal(tab["bscore_stack"], g["bscore_stack"])
    assert int(hn[0, 1]) == int(g["hyp_score"][0])
    p.close()


def test_pipeline_with_the_semi_continuous_scorer_tidigits(tmp_path):
    """tidigits: s2_semi (4 streams x 256 densities, 4-bit weights), 5-state HMMs, s2_4x feature vectors (51 dims) computed by the
    reference's feature module from the bundled cepstra files and handed over as psgpu_decode_first_pass_feat takes them: the
    first pass of all the utterances of the reference's regression set in ONE batch, every back-pointer table against the
    reference's own decode of that file (a new decoder each, as the pipeline scores each utterance)"""
    _need_ref()
    import pocketsphinx_amd as P
    tdir = os.path.join(REF, "data", "tidigits")
    lm, dic = os.path.join(tdir, "tidigits.lm.bin"), os.path.join(tdir, "tidigits.dic")
    names = [ln.strip() for ln in open(os.path.join(tdir, "tidigits.ctl")) if ln.strip()]
    assert len(names) >= 30
    feats, golds = [], []
    for n in names:
        mfc = os.path.join(tdir, n + ".mfc")
        feats.append(_ref_dump(tmp_path, "dynfeat", "tidigits", lm, dic, [mfc])["feat"])
        golds.append(_ref_dump(tmp_path, "fwdtree", "tidigits", lm, dic, [mfc], ("fwdflat", "no", "bestpath", "no")))
    g0 = golds[0]
    semi = P.SemiMgau(_load("semi_tidigits_tables.npz"))
    p = P.DecodePipeline(None, None, g0, g0["par"], g0, scorer=semi)
    p.run_feat(np.concatenate(feats), [f.shape[0] for f in feats])
    hn, hyp, res = p.fetch()
    for u, g in enumerate(golds):
        assert int(res[u, 3]) == 0, (names[u], res[u])
        tab = p.tables(u, res)
        assert np.array_equal(tab["bp"], g["bp"]), names[u]
        assert np.array_equal(tab["bscore_stack"], g["bscore_stack"]), names[u]
        assert int(hn[u, 1]) == int(g["hyp_score"][0]), names[u]
    with pytest.raises(P.PsgpuError):
        p.run([np.zeros(16000, np.int16)])                # (from PCM this pipeline has no front end / another feature type: refused)
    # a session with this scorer: the first utterance of a new session is a new decoder's (its ring starts from codeword = rank)
    p.session(True)
    p.run_feat(feats[3], [feats[3].shape[0]])
    hn, hyp, res = p.fetch()
    tab = p.tables(0, res)
    assert np.array_equal(tab["bp"], golds[3]["bp"]) and int(hn[0, 1]) == int(golds[3]["hyp_score"][0])
    p.close()


def test_pipeline_from_audio_with_the_semi_continuous_scorer_tidigits(tmp_path):
    """the same model from AUDIO: the pipeline's front end with tidigits's parameters (20 filters to 4 kHz, DC removal, DITHER -- the
    model's feat.params insist on it; -seed 17 on both sides, the reference would seed it from the clock) and the s2_4x feature type
    installed (psgpu_decode_set_feat) -- PCM -> MFCC -> s2_4x -> s2_semi scores -> phone loop -> tree search on the device, against the
    reference decoding the same file.  (The dither's generator goes on from utterance to utterance: the batch's FIRST utterance is the
    one a new decoder's first decode equals.)"""
    _need_ref()
    import pocketsphinx_amd as P
    tdir = os.path.join(REF, "data", "tidigits")
    lm, dic = os.path.join(tdir, "tidigits.lm.bin"), os.path.join(tdir, "tidigits.dic")
    raw = os.path.join(tdir, "dhd.2934z.raw")
    knobs = ("seed", "17", "fwdflat", "no", "bestpath", "no")
    fe_t = _ref_dump(tmp_path, "mfcc", "tidigits", lm, dic, [raw, 1], knobs)
    assert int(fe_t["par"][13]) == 1 and int(fe_t["dither_seed"][0]) == 17
    g = _ref_dump(tmp_path, "fwdtree", "tidigits", lm, dic, [raw], knobs)
    pcm = np.fromfile(raw, np.int16)
    semi = P.SemiMgau(_load("semi_tidigits_tables.npz"))
    p = P.DecodePipeline(fe_t, None, g, g["par"], g, scorer=semi)
    with pytest.raises(P.PsgpuError):
        p.run([pcm])                                       # (1s_c_d_dd vectors are not what this scorer takes)
    ft = P.FeatType("s2_4x", cmn="current")
    p.set_feat(ft)
    p.run([pcm, pcm[:12000], pcm])
    hn, hyp, res = p.fetch()
    assert not res[:, 3].any()
    tab = p.tables(0, res)
    assert np.array_equal(tab["bp"], g["bp"]) and np.array_equal(tab["bscore_stack"], g["bscore_stack"])
    assert int(hn[0, 1]) == int(g["hyp_score"][0])
    assert int(hn[1, 0]) > 0 and int(res[2, 2]) == int(res[0, 2])
    p.close(); ft.close()


def test_streams_with_the_semi_continuous_scorer_tidigits(tmp_path):
    """a batch of live decoders with the semi-continuous scorer: eight tidigits utterances as eight streams, fed 11-23 frames a step
    (psgpu_decode_streams_step: the scorer's lists carried per stream by psgpu_semi_score_batch_carry_dev, frames numbered on from
    where each stream stands); every stream's final tables equal the reference's decode of that file by a new decoder"""
    _need_ref()
    import pocketsphinx_amd as P
    tdir = os.path.join(REF, "data", "tidigits")
    lm, dic = os.path.join(tdir, "tidigits.lm.bin"), os.path.join(tdir, "tidigits.dic")
    names = [ln.strip() for ln in open(os.path.join(tdir, "tidigits.ctl")) if ln.strip()][:8]
    feats, golds = [], []
    for n in names:
        mfc = os.path.join(tdir, n + ".mfc")
        feats.append(_ref_dump(tmp_path, "dynfeat", "tidigits", lm, dic, [mfc])["feat"])
        golds.append(_ref_dump(tmp_path, "fwdtree", "tidigits", lm, dic, [mfc], ("fwdflat", "no", "bestpath", "no")))
    g0 = golds[0]
    semi = P.SemiMgau(_load("semi_tidigits_tables.npz"))
    p = P.DecodePipeline(None, None, g0, g0["par"], g0, scorer=semi)
    n = len(names)
    p.streams_begin(n, max(f.shape[0] for f in feats) + 8, 24)
    pos = [0] * n; done = [False] * n
    for step in range(200):
        fs, fin = [], []
        for u in range(n):
            k = 0 if done[u] else min(11 + (step + 3 * u) % 13, feats[u].shape[0] - pos[u])
            fs.append(feats[u][pos[u]:pos[u] + k]); pos[u] += k
            fin.append(k > 0 and pos[u] == feats[u].shape[0])
        p.streams_step(fs, fin)
        hn, hyp, res = p.fetch()
        for u in range(n):
            if fin[u]:
                done[u] = True
                tab = p.tables(u, res)
                assert int(res[u, 3]) == 0 and np.array_equal(tab["bp"], golds[u]["bp"]), (names[u], step)
                assert np.array_equal(tab["bscore_stack"], golds[u]["bscore_stack"]) and int(hn[u, 1]) == int(golds[u]["hyp_score"][0]), names[u]
        if all(done):
            break
    assert all(done) and p.live_frames_searched() == sum(f.shape[0] for f in feats)
    p.close()
