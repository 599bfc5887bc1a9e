"""Drop-in proof on the GPU box: the UNMODIFIED reference decoder
(oracle/_ref/libpocketsphinx.so) with its GMM scorer swapped for the psgpu
shim (integration/psgpu_mgau_shim.c -> libpsgpu.so) must produce, for whole
ps_decode_raw-style decodes, bit-identical senone scores on every
frame_eval call of every pass, the same hypothesis + path score and the same
segmentation as the same decoder with its CPU scorer.  The comparison is done
by oracle/dropin_decode.c (test infrastructure)."""
import json
import os
import subprocess

import pytest

import pso

REF = pso.REF_DIR
BIN = os.path.join(REF, "dropin_decode")
MODEL = os.path.join(REF, "model", "en-us")
DATA = os.path.join(REF, "data")


BIN_FULL = os.path.join(REF, "dropin_decode_full")


def run(raw, nrep, *extra, lm="turtle.lm.bin", dic="turtle.dic", binary=None, model=None):
    BIN = binary or globals()["BIN"]
    if not os.path.exists(BIN):
        pytest.fail("oracle/_ref/dropin_decode is missing: run __graft_entry__.build() where "
                    "/root/reference is present (the built oracle/_ref travels with gpurun)")
    inp = raw if raw.startswith("@") else os.path.join(DATA, raw)
    argv = [BIN, model or MODEL, "-" if lm == "-" else os.path.join(DATA, lm),
            "-" if dic == "-" else os.path.join(DATA, dic), inp, str(nrep)] + [str(e) for e in extra]
    p = subprocess.run(argv, capture_output=True, text=True, timeout=600)
    assert p.stdout.strip(), "no output (rc %d): %s" % (p.returncode, p.stderr[-2000:])
    r = json.loads(p.stdout.strip().splitlines()[-1])
    r["rc"] = p.returncode
    return r


def _write_mllr(tmp_path, n_feat=3, veclen=13, seed=5):
    """ps_mllr_read format (ps_mllr.c:57-125): n_class, n_feat, then per stream
    veclen, A (veclen x veclen), b, h."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lines = ["1", str(n_feat)]
    for _ in range(n_feat):
        lines.append(str(veclen))
        A = np.eye(veclen) + 0.02 * rng.standard_normal((veclen, veclen))
        for row in A:
            lines.append(" ".join("%.6f" % v for v in row))
        lines.append(" ".join("%.6f" % v for v in 0.05 * rng.standard_normal(veclen)))
        lines.append(" ".join("%.6f" % v for v in 1.0 + 0.05 * rng.random(veclen)))
    p = tmp_path / "mllr_3x13"
    p.write_text("\n".join(lines) + "\n")
    return str(p)


CASES = {
    # name: (raw, nrep, extra config)
    "default_3pass_x2": ("goforward.raw", 2, ()),                 # fwdtree + fwdflat + bestpath, carry-over (F7)
    "fwdtree_only": ("goforward.raw", 1, ("fwdflat", "no", "bestpath", "no")),
    "compallsen_plw0": ("goforward.raw", 1, ("compallsen", "yes", "pl_window", "0")),
    "ds2": ("goforward.raw", 1, ("ds", "2")),                     # codebook scan every 2nd frame
    "fwdflat_only": ("goforward.raw", 1, ("fwdtree", "no")),      # pass-2 codebook masking without pass 1
    "numbers": ("numbers.raw", 1, ()),
    "topn2": ("goforward.raw", 1, ("topn", "2")),                 # any-shape per-call kernels
    "topn6": ("goforward.raw", 1, ("topn", "6")),
    "something": ("something.raw", 1, ()),
    "librivox_0870": ("librivox-0870.raw", 1, ()),
    # vt->transform through the shim: 1-class MLLR for 3 streams x 13 written below (the bundled
    # test/data/mllr_matrices is 1 x 39 for an4_ci_cont and crashes the reference itself on en-us)
    "mllr_after_attach": ("goforward.raw", 1, ("mllr_after", "@MLLR@")),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_dropin_decode_identical(case, tmp_path):
    raw, nrep, extra = CASES[case]
    if "@MLLR@" in extra:
        extra = tuple(_write_mllr(tmp_path) if e == "@MLLR@" else e for e in extra)
    r = run(raw, nrep, *extra)
    assert r["mgau"] == "ptm-psgpu"
    if case in ("default_3pass_x2", "fwdtree_only", "numbers", "librivox_0870"):
        # pass-1 calls are answered from the look-ahead cache (one batched pass per utterance)
        assert r["cache_served"] > r["calls_gpu"] // 2, r
    assert r["device_calls"] == r["calls_gpu"] > 0
    assert r["calls_cpu"] == r["calls_gpu"]
    assert r["mismatching_calls"] == 0, r
    assert r["hyp_equal"] and r["seg_equal"], r
    assert r["ok"] and r["rc"] == 0, r
    if raw == "goforward.raw" and "mllr_after" not in extra:
        assert r["hyp_gpu"] == "go forward ten meters"


@pytest.mark.gpu
def test_dropin_clustered_4bit_sendump(tmp_path):
    """an acoustic model whose sendump is 4-bit clustered (read_sendump, ptm_mgau.c:457-654): the shim expands the
    weights as ptm_mgau_senone_eval looks them up (:375-379, nibble chosen by the low bit of the byte) -- every
    frame_eval call's scores, hypothesis and segmentation equal the CPU scorer's on the same model"""
    import numpy as np
    g = np.load(os.path.join(pso.GOLDEN_DIR, "ptm_4bit_goforward.npz"))
    model = pso.write_clustered_model_dir(str(tmp_path / "en-us-4bit"), MODEL, g)
    r = run("goforward.raw", 1, model=model)
    assert r["mgau"] == "ptm-psgpu" and r["device_calls"] == r["calls_gpu"] > 0, r
    assert r["calls_cpu"] == r["calls_gpu"] and r["mismatching_calls"] == 0, r
    assert r["hyp_equal"] and r["seg_equal"] and r["ok"] and r["rc"] == 0, r


FULL_CASES = {
    # name: (raw, nrep, extra) -- decoder B runs GMM scoring AND every hmm_vit_eval loop on the device
    "default_3pass_x2": ("goforward.raw", 2, ()),
    "fwdtree_only": ("goforward.raw", 1, ("fwdflat", "no", "bestpath", "no")),
    "fwdflat_only": ("goforward.raw", 1, ("fwdtree", "no")),
    "plw0": ("goforward.raw", 1, ("pl_window", "0")),
    "search_only_cpu_gmm": ("goforward.raw", 1, ("psgpu_mgau", "no")),
    "numbers": ("numbers.raw", 1, ()),
    "librivox_0870": ("librivox-0870.raw", 1, ()),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(FULL_CASES))
def test_dropin_full_decode_identical(case):
    """GMM shim + the three hmm_vit_eval loops (fwdtree evaluate_channels,
    fwdflat_eval_chan, phone-loop evaluate_hmms) on the device: identical
    senone-score calls (=> identical active sets, i.e. identical search
    trajectories), hypothesis, path score and segmentation."""
    raw, nrep, extra = FULL_CASES[case]
    r = run(raw, nrep, *extra, binary=BIN_FULL)
    assert r["search_hooks"] and r["hmm_batches"] > 0 and r["hmm_evals"] > r["hmm_batches"]
    assert r["calls_cpu"] == r["calls_gpu"] and r["mismatching_calls"] == 0, r
    assert r["hyp_equal"] and r["seg_equal"], r
    assert r["ok"] and r["rc"] == 0, r
    if raw == "goforward.raw":
        assert r["hyp_gpu"] == "go forward ten meters"
    if case == "default_3pass_x2":
        # decoder A of the hooked library (hooks idle) still equals the unmodified reference
        import numpy as np
        g = np.load(os.path.join(pso.GOLDEN_DIR, "decode_default.npz"))
        assert r["hyp_cpu"] == bytes(g["hyp"]).decode()


TD = os.path.join(DATA, "tidigits")
TD_KW = dict(model=os.path.join(REF, "model", "tidigits"), lm="tidigits/tidigits.lm.bin",
             dic="tidigits/tidigits.dic")


def _match_file():
    out = []
    for line in open(os.path.join(TD, "test-tidigits-simple.match")):
        line = line.strip()
        if line:
            hyp, tail = line.rsplit("(", 1)
            uid, score = tail.rstrip(")").split()
            out.append((uid, hyp.strip(), int(score)))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("binary", ["mgau", "full"])
def test_tidigits_regression_vs_reference_match_file(binary):
    """The reference's own regression (test/regression/test-tidigits-simple.sh):
    the tidigits utterances of tidigits.ctl, s2_semi scorer (4-bit weights) + n-gram fwdtree /
    fwdflat / bestpath over 5-state HMMs, hypotheses AND path scores pinned by
    test/data/tidigits/test-tidigits-simple.match.  Decoder B runs the scorer
    (and, for "full", every 5-state Viterbi step) on the device; it must equal
    decoder A call for call and reproduce the match file exactly."""
    r = run("@%s:%s" % (os.path.join(TD, "tidigits.ctl"), TD), 1, **TD_KW,
            binary=BIN_FULL if binary == "full" else BIN)
    assert r["mgau"] == "s2_semi-psgpu" and r["device_calls"] == r["calls_gpu"] > 0
    assert r["mismatching_calls"] == 0 and r["hyp_equal"] and r["seg_equal"] and r["ok"], \
        {k: v for k, v in r.items() if k != "utts"}
    if binary == "full":
        assert r["hmm_evals"] > 0
    want = _match_file()
    assert len(r["utts"]) == len(want) >= 30
    for u, (uid, hyp, score) in zip(r["utts"], want):
        assert u["id"] == uid
        assert u["hyp"] == hyp, u
        # path scores: the reference's own check is compare_table with tolerance 100000
        # (test/regression/test-tidigits-simple.sh); the CPU build here already differs from
        # the published file by a few hundred, and decoder B equals decoder A exactly (above)
        assert abs(u["score"] - score) <= 100000, u


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [("topn_beam", "20,35,10,20"), ("topn", "6", "ds", "2"),
                                   ("topn", "7", "compallsen", "yes"), ("topn", "2", "pl_window", "0")])
def test_tidigits_scorer_options(extra):
    r = run("tidigits/woman.ak.276317oa.mfc", 2, *extra, **TD_KW, binary=BIN_FULL)
    assert r["mgau"] == "s2_semi-psgpu"
    assert r["mismatching_calls"] == 0 and r["hyp_equal"] and r["seg_equal"] and r["ok"], \
        {k: v for k, v in r.items() if k != "utts"}


MS_KW_AN4 = dict(model=os.path.join(REF, "model", "an4_ci_cont"))
MS_KW_ENUS = dict(model=os.path.join(REF, "model", "en-us-ms"))


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,extra", [
    ("an4", MS_KW_AN4, ()),
    ("an4_compall_aw2", MS_KW_AN4, ("compallsen", "yes", "aw", "2")),
    ("en_us_ms", MS_KW_ENUS, ("senmgau", ".ptm.")),
    ("en_us_ms_topn2", MS_KW_ENUS, ("senmgau", ".ptm.", "topn", "2", "fwdflat", "no")),
])
def test_ms_dropin_decode_identical(name, kw, extra):
    """ms scorer behind the shim (+ every Viterbi step on the device): an4_ci_cont,
    the reference's only bundled continuous model (test/unit/test_mllr.c decodes
    goforward with it), and en-us forced through the ms scorer."""
    r = run("goforward.raw", 2, *extra, **kw, binary=BIN_FULL)
    assert r["mgau"] == "ms-psgpu" and r["device_calls"] == r["calls_gpu"] > 0
    assert r["mismatching_calls"] == 0 and r["hyp_equal"] and r["seg_equal"] and r["ok"], \
        {k: v for k, v in r.items() if k != "utts"}
    assert r["hyp_gpu"] == "go forward ten meters"


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,extra", [
    ("an4", MS_KW_AN4, ()),                                               # + the reference's second and third pass
    ("an4_first_pass", MS_KW_AN4, ("fwdflat", "no", "bestpath", "no")),
    ("en_us_ms", MS_KW_ENUS, ("senmgau", ".ptm.", "fwdflat", "no", "bestpath", "no")),
    ("en_us_ms_live", MS_KW_ENUS, ("senmgau", ".ptm.", "fwdflat", "no", "bestpath", "no", "chunked", "6000")),
])
def test_ms_dropin_device_search_vtable(name, kw, extra):
    """BASELINE configs[3] behind ps_decode_raw: a decoder whose scorer is the multi-stream one (ms_cont_mgau_frame_eval,
    reference src/ms_mgau.c:192-282 -- acmod_init_am's route for -senmgau and for models without a sendump, src/acmod.c:62-130)
    with the device ps_searchfuncs_t bound: psgpu_device_search_attach takes the ms model out of the psgpu scorer and the
    first pass -- ms scores, phone loop, lexicon-tree search -- runs on the MI355X, twice in a row (the second utterance
    inherits the search's session state), the reference's own later passes on the injected table where configured.
    Hypothesis, score and every segment equal the CPU decoder's; with `chunked` every partial hypothesis too."""
    r = run("goforward.raw", 2, "psgpu_device_vtable", "yes", *extra, **kw)
    assert r["ok"] and r["rc"] == 0, {k: v for k, v in r.items() if k != "utts"}
    assert r["mgau"] == "ms-psgpu" and r["hyp_equal"] and r["seg_equal"] and r["score_cpu"] == r["score_gpu"]
    assert r["device_search_frames"] > 0 and r["n_seg"] > 0
    if "chunked" in extra:
        assert r["partial_equal"] and r["partial_results"] >= 5


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [(), ("fwdflat", "no", "bestpath", "no")])
def test_large_vocabulary_dropin(extra):
    """126,052-word vocabulary (every base word of cmudict-en-us.dict, synthetic
    Zipf LM: the reference ships no large LM): ~250k lextree
    channels and thousands of active HMMs per frame go through the Viterbi
    kernel each frame; scores, hypothesis and segmentation stay identical."""
    r = run("goforward.raw", 1, *extra, lm="big.arpa", dic="cmudict-en-us.dict", binary=BIN_FULL)
    assert r["mismatching_calls"] == 0 and r["hyp_equal"] and r["seg_equal"] and r["ok"], \
        {k: v for k, v in r.items() if k != "utts"}
    assert r["hyp_gpu"] == "go forward ten meters"
    assert r["hmm_evals"] > 1000 * r["n_frames"], r      # > 1000 HMMs per frame on average
    assert r["cache_served"] > 0


OTHER_SEARCHES = {
    # the other consumers of the same boundary: they call acmod_score()
    # exactly like the n-gram search, so the GMM shim serves them unchanged
    "fsg": dict(lm="-", extra=("fsg", os.path.join(DATA, "goforward.fsg")), hyp="go forward ten meters"),
    "jsgf": dict(lm="-", extra=("jsgf", os.path.join(DATA, "goforward.gram")), hyp="go forward ten meters"),
    "keyphrase": dict(lm="-", extra=("keyphrase", "forward", "kws_threshold", "1e-20"), hyp=None),
    "allphone": dict(lm="-", dic="-", extra=("allphone", os.path.join(DATA, "en-us-phone.lm.bin"), "beam", "1e-20",
                                            "pbeam", "1e-10", "allphone_ci", "false", "lw", "2.0"),
                     hyp="SIL G OW F AO R W ER D T AE N M IY IH ZH ER Z S V SIL"),    # test/unit/test_allphone.c
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(OTHER_SEARCHES))
def test_other_searches_through_the_gmm_shim(name):
    c = OTHER_SEARCHES[name]
    r = run("goforward.raw", 2, *c["extra"], lm=c["lm"], dic=c.get("dic", "turtle.dic"))
    assert r["mgau"] == "ptm-psgpu" and r["device_calls"] == r["calls_gpu"] > 0
    assert r["mismatching_calls"] == 0 and r["hyp_equal"] and r["seg_equal"] and r["ok"], \
        {k: v for k, v in r.items() if k != "utts"}
    if c["hyp"] is not None:
        assert r["utts"][0]["hyp"] == c["hyp"]      # first utterance (the second starts from updated CMN state)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(OTHER_SEARCHES) + ["align"])
def test_other_searches_viterbi_hooks(name):
    """The same searches with their own hmm_vit_eval loops (fsg_search_hmm_eval,
    phmm_eval_all, kws_search_hmm_eval, state_align evaluate_hmms) routed to the
    Viterbi kernel through integration/hook_*.c."""
    if name == "align":
        c = dict(lm="-", extra=("align_text", "go forward ten meters"), hyp="go forward ten meters")
    else:
        c = OTHER_SEARCHES[name]
    r = run("goforward.raw", 2, *c["extra"], lm=c["lm"], dic=c.get("dic", "turtle.dic"), binary=BIN_FULL)
    assert r["search_hooks"] and r["hmm_batches"] > 0 and r["hmm_evals"] > r["hmm_batches"], r
    assert r["mismatching_calls"] == 0 and r["hyp_equal"] and r["seg_equal"] and r["ok"], \
        {k: v for k, v in r.items() if k != "utts"}
    if c["hyp"] is not None:
        assert r["utts"][0]["hyp"] == c["hyp"]


@pytest.mark.gpu
def test_cards_regression_jsgf():
    """test/regression/test-cards.sh: en-us PTM + JSGF grammar (fsg search), 5 utterances,
    hypotheses pinned by test/data/cards/cards.hyp."""
    cards = os.path.join(DATA, "cards")
    r = run("@%s:%s:raw" % (os.path.join(cards, "cards.fileids"), cards), 1,
            "jsgf", os.path.join(cards, "cards.gram"), "bestpath", "no",
            lm="-", dic="cmudict-en-us.dict")
    assert r["mismatching_calls"] == 0 and r["hyp_equal"] and r["seg_equal"] and r["ok"], \
        {k: v for k, v in r.items() if k != "utts"}
    want = [l.rsplit("(", 1)[0].strip() for l in open(os.path.join(cards, "cards.hyp")) if l.strip()]
    assert [u["hyp"] for u in r["utts"]] == want


@pytest.mark.gpu
def test_tidigits_fsg_regression():
    """test/regression/test-tidigits-fsg.sh: s2_semi + FSG search, hypotheses pinned by
    test/data/tidigits/test-tidigits-fsg.match."""
    r = run("@%s:%s" % (os.path.join(TD, "tidigits.ctl"), TD), 1,
            "fsg", os.path.join(TD, "tidigits.fsg"), "wbeam", "1e-48", "bestpath", "no",
            model=TD_KW["model"], lm="-", dic=TD_KW["dic"])
    assert r["mgau"] == "s2_semi-psgpu"
    assert r["mismatching_calls"] == 0 and r["hyp_equal"] and r["seg_equal"] and r["ok"], \
        {k: v for k, v in r.items() if k != "utts"}
    want = [l.rsplit("(", 1)[0].strip() for l in open(os.path.join(TD, "test-tidigits-fsg.match")) if l.strip()]
    assert [u["hyp"] for u in r["utts"]] == want


@pytest.mark.gpu
def test_batch_decode_threads_share_the_gpu():
    """integration/psgpu_batch_decode.c: one decoder per host thread, each with its own
    psgpu model/state/stream on the same device: hypotheses stay identical."""
    exe = os.path.join(REF, "psgpu_batch_decode")
    p = subprocess.run([exe, MODEL, os.path.join(DATA, "turtle.lm.bin"), os.path.join(DATA, "turtle.dic"),
                        os.path.join(DATA, "goforward.raw"), "12", "4", "gpu"],
                       capture_output=True, text=True, timeout=600)
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 0 and r["hyp_mismatches"] == 0 and r["mgau"] == "ptm-psgpu", r
    assert r["hyp"] == "go forward ten meters" and r["frames"] == 12 * 279


def test_attach_fails_loudly_without_gpu():
    """No CPU fallback inside the product: on a box without a gfx950 device the
    attach fails and the checker exits 3."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref not built")
    p = subprocess.run([BIN, MODEL, os.path.join(DATA, "turtle.lm.bin"), os.path.join(DATA, "turtle.dic"),
                        os.path.join(DATA, "goforward.raw"), "1"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 3
    assert "psgpu_mgau_attach failed" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("raw,nrep,extra,binary", [
    ("goforward.raw", 3, (), None),                         # noise tracker carried over three utterances
    ("numbers.raw", 1, (), None),
    ("librivox-0870.raw", 1, ("psgpu_search", "yes"), "full"),   # front end + GMM + Viterbi all on the device
])
def test_dropin_device_front_end(raw, nrep, extra, binary):
    """psgpu_fe yes: decoder B gets its cepstra from the device front end
    (integration/psgpu_fe_shim.c: tables read out of the decoder's own fe_t,
    psgpu_process_raw_full = ps_process_raw(full_utt)), decoder A runs the reference's
    fe on the host.  Same features => every frame_eval call hashes equal, same
    hypothesis, score and segmentation."""
    r = run(raw, nrep, "psgpu_fe", "yes", *extra, binary=BIN_FULL if binary == "full" else None)
    assert r["device_fe"] is True
    assert r["ok"] and r["rc"] == 0, r
    assert r["mismatching_calls"] == 0 and r["hyp_equal"] and r["seg_equal"], r
    assert r["hyp_gpu"], r


def _batch_api(workers, flags, files, *extra, full=False, lm="turtle.lm.bin", dic="turtle.dic", devices=None):
    binary = os.path.join(REF, "batch_api_check_full" if full else "batch_api_check")
    if not os.path.exists(binary):
        pytest.fail("oracle/_ref/batch_api_check is missing (make -C oracle where /root/reference is present)")
    argv = [binary, MODEL, os.path.join(DATA, lm), os.path.join(DATA, dic), str(workers), str(flags)] + \
           [os.path.join(DATA, f) for f in files]
    if extra:
        argv += ["--"] + [str(e) for e in extra]
    env = dict(os.environ)
    if devices:
        env["BATCH_CHECK_DEVICES"] = devices
    p = subprocess.run(argv, capture_output=True, text=True, timeout=900, env=env)
    assert p.stdout.strip(), "no output (rc %d): %s" % (p.returncode, p.stderr[-2000:])
    r = json.loads(p.stdout.strip().splitlines()[-1])
    r["rc"] = p.returncode
    return r


FILES = ["goforward.raw", "numbers.raw", "something.raw", "librivox-0870.raw", "goforward.raw", "numbers.raw"]


@pytest.mark.gpu
@pytest.mark.parametrize("workers,flags,full,extra", [
    (1, 0, False, ()),        # GMM on the device, one worker
    (3, 0, False, ()),        # more utterances than workers: work queue, per-utterance reset
    (3, 1, False, ()),        # + cepstra of the whole batch from one device front-end call
    (4, 3, True, ()),         # + hmm_vit_eval loops on the device (hooked library)
    (3, 1, False, ("fwdflat", "no", "bestpath", "no")),   # pass 1 only: its scores are not masked by later passes
    (2, 1, False, ("fwdtree", "no")),                     # pass 2 only
    (3, 9, False, ("fwdflat", "no", "bestpath", "no")),   # + each utterance's phone loop in one device launch
    (3, 9, False, ()),
])
def test_decode_batch_api(workers, flags, full, extra):
    """psgpu_decode_batch (the additive batch call): every utterance's
    hypothesis, path score, frame count and full segmentation equal what a fresh
    unmodified CPU decoder gives for it -- for the batch in order, reversed, and one
    utterance at a time (B = 1 is the drop-in path)."""
    r = _batch_api(workers, flags, FILES, *extra, full=full)
    assert r["ok"] and r["rc"] == 0, r
    assert r["B"] == len(FILES) and r["mismatch_batch"] == r["mismatch_reversed"] == r["mismatch_single"] == 0
    assert r["hyps"][0] == "go forward ten meters"


@pytest.mark.parametrize("extra", [(), ("fwdflat", "no", "bestpath", "no")])
def test_decode_batch_api_cpu_only_is_order_independent(extra):
    """PSGPU_BATCH_CPU_ONLY (flag 4, no device involved): the per-utterance reset of the
    batch call makes the REFERENCE's own decoders order-independent -- without it a
    decoder's path scores depend on what it decoded before (multiplex HMMs of the lexicon
    tree keep their senone-sequence ids across utterances; see psgpu_decode_batch.c)."""
    if not os.path.exists(os.path.join(REF, "batch_api_check")):
        pytest.skip("oracle/_ref not built")
    r = _batch_api(2, 4, ["goforward.raw", "numbers.raw", "goforward.raw"], *extra)
    assert r["ok"] and r["rc"] == 0, r


def test_decode_batch_api_refuses_without_device():
    """No silent CPU fallback: without a usable device psgpu_batch_init fails (exit code 3)
    unless PSGPU_BATCH_CPU_ONLY is asked for."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    binary = os.path.join(REF, "batch_api_check")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref not built")
    p = subprocess.run([binary, MODEL, os.path.join(DATA, "turtle.lm.bin"), os.path.join(DATA, "turtle.dic"), "1", "0",
                        os.path.join(DATA, "goforward.raw")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stdout, p.stderr[-500:])


@pytest.mark.gpu
@pytest.mark.parametrize("raw,nrep,extra,full", [
    ("goforward.raw", 2, (), False),                                   # 3-pass, two utterances
    ("numbers.raw", 1, ("fwdflat", "no", "bestpath", "no"), False),    # pass 1 alone: nothing masks its scores
    ("librivox-0870.raw", 1, ("compallsen", "yes"), False),            # normaliser = the all-senone minimum
    ("something.raw", 1, ("pl_window", "2", "pl_weight", "1.5"), False),
    ("goforward.raw", 1, ("psgpu_search", "yes", "psgpu_fe", "yes"), True),   # with every other device component
])
def test_dropin_device_phone_loop(raw, nrep, extra, full):
    """psgpu_phone_loop yes: the whole phone-loop search of an utterance is one
    device launch on the scorer's device-resident rows; the decoder's phone_loop_search_t only
    receives pls->penalties per step and makes no frame_eval calls of its own.  Every step's
    penalties vector must equal the CPU decoder's, and hypothesis, path score and segmentation
    (whose acoustic scores depend on every frame's normalisation) must be identical."""
    r = run(raw, nrep, "psgpu_phone_loop", "yes", *extra, binary=BIN_FULL if full else None)
    assert r["ok"] and r["rc"] == 0, r
    assert r["pl_mismatch"] == 0 and r["pl_steps"] > 0 and r["hyp_equal"] and r["seg_equal"], r
    assert r["pl_device_steps"] == r["pl_steps"] and r["pl_host_steps"] == 0, r


@pytest.mark.gpu
@pytest.mark.parametrize("break_at", [1, 7, 123])
def test_dropin_device_phone_loop_resumes_on_host(break_at, monkeypatch):
    """Leaving the device path in the middle of an utterance (test hook PSGPU_PL_BREAK_AT): the
    reference's HMMs and penalty ring are loaded with the device's state of the previous frame and
    the reference's own step carries on -- penalties and results still identical."""
    monkeypatch.setenv("PSGPU_PL_BREAK_AT", str(break_at))
    r = run("goforward.raw", 1, "psgpu_phone_loop", "yes", "fwdflat", "no", "bestpath", "no")
    assert r["ok"] and r["rc"] == 0, r
    assert r["pl_mismatch"] == 0 and r["hyp_equal"] and r["seg_equal"], r
    assert r["pl_device_steps"] == break_at and r["pl_host_steps"] == r["pl_steps"] - break_at, r


@pytest.mark.gpu
@pytest.mark.parametrize("raw,nrep,extra,lm,dic,model", [
    ("goforward.raw", 1, (), "turtle.lm.bin", "turtle.dic", None),
    ("numbers.raw", 1, (), "turtle.lm.bin", "turtle.dic", None),
    ("something.raw", 1, ("pl_window", "2", "pl_weight", "1.5"), "turtle.lm.bin", "turtle.dic", None),
    ("goforward.raw", 1, ("maxhmmpf", "100", "maxwpf", "5"), "turtle.lm.bin", "turtle.dic", None),   # histogram + word pruning
    # -compallsen yes: every senone scored, rows normalised over all of them (psgpu_decode_compallsen)
    ("goforward.raw", 1, ("compallsen", "yes"), "turtle.lm.bin", "turtle.dic", None),
    ("numbers.raw", 1, ("compallsen", "yes", "bestpath", "yes"), "turtle.lm.bin", "turtle.dic", None),
    # pass 3 on the host over the injected table: ngram_search_lattice + ps_lattice_bestpath (SURVEY f-2)
    ("goforward.raw", 1, ("bestpath", "yes"), "turtle.lm.bin", "turtle.dic", None),
    ("numbers.raw", 1, ("bestpath", "yes"), "turtle.lm.bin", "turtle.dic", None),
    # 715 words (oracle/make_medium_task.py): the language scores come from the model's trie on the device
    ("goforward.raw", 1, (), "medium.arpa", "medium.dic", None),
    ("numbers.raw", 1, ("bestpath", "yes", "maxwpf", "8"), "medium.arpa", "medium.dic", None),
    # the full cmudict vocabulary (134,865 words, 248 k tree channels, SURVEY F9b's synthetic large LM)
    ("goforward.raw", 1, (), "big.arpa", "cmudict-en-us.dict", None),
])
def test_dropin_device_first_pass(raw, nrep, extra, lm, dic, model):
    """psgpu_device_search yes (integration/psgpu_device_decode.c): decoder B's whole first
    pass -- front end, features, senone scores, phone loop, lexicon-tree Viterbi search -- runs on the
    MI355X; the back-pointer table it produces is copied into the live ngram_search_t in the
    reference's layout and the REFERENCE's own ps_get_hyp / ps_seg_iter read it.  Hypothesis, path
    score and every segment (word, frames, acoustic / language score, back-off) must be identical to
    the CPU decoder's on the same audio."""
    flags = ("fwdflat", "no") + (() if "bestpath" in extra else ("bestpath", "no"))
    r = run(raw, nrep, "psgpu_device_search", "yes", *flags, *extra, lm=lm, dic=dic, model=model)
    assert r["ok"] and r["rc"] == 0, r
    assert r["hyp_equal"] and r["seg_equal"] and r["score_cpu"] == r["score_gpu"], r
    # (ps_get_n_frames() reports acmod->output_frame + 1: one more than the frames searched)
    assert r["device_search_frames"] == r["total_frames"] - r["n_utts"] and r["n_seg"] > 0, r


@pytest.mark.gpu
@pytest.mark.parametrize("raw,nrep,extra,lm,dic", [
    ("goforward.raw", 2, ("fwdflat", "no", "bestpath", "no"), "turtle.lm.bin", "turtle.dic"),
    ("numbers.raw", 1, ("fwdflat", "no", "bestpath", "yes"), "turtle.lm.bin", "turtle.dic"),     # lattice pass over the injected table
    ("goforward.raw", 1, (), "turtle.lm.bin", "turtle.dic"),                                      # the default three passes
    ("numbers.raw", 1, (), "turtle.lm.bin", "turtle.dic"),
    ("something.raw", 1, ("fwdflat", "no", "bestpath", "no", "pl_window", "2", "pl_weight", "1.5"), "turtle.lm.bin", "turtle.dic"),
    ("goforward.raw", 1, ("fwdflat", "no", "bestpath", "no"), "medium.arpa", "medium.dic"),      # trie language model on the device
])
def test_dropin_device_search_vtable(raw, nrep, extra, lm, dic):
    """psgpu_device_vtable yes (integration/psgpu_device_decode.c, psgpu_device_search_attach): decoder B's n-gram search
    object carries the device ps_searchfuncs_t and is driven by the UNMODIFIED public calls -- ps_start_utt,
    ps_process_raw(full_utt), ps_end_utt, ps_get_hyp, ps_seg_iter: step() buffers feature vectors, finish() runs scorer ->
    phone loop -> lexicon-tree search on the MI355X and injects the tables; with -fwdflat yes / -bestpath yes the
    reference's own later passes run on them.  Hypothesis, path score and every segment equal the CPU decoder's."""
    r = run(raw, nrep, "psgpu_device_vtable", "yes", *extra, lm=lm, dic=dic)
    assert r["ok"] and r["rc"] == 0, r
    assert r["hyp_equal"] and r["seg_equal"] and r["score_cpu"] == r["score_gpu"], r
    assert r["n_seg"] > 0 and r["device_search_frames"] > 0, r


@pytest.mark.gpu
@pytest.mark.parametrize("lmname", ["turtle", "medium"])
def test_dropin_device_search_with_a_model_set_and_a_word_class(lmname, tmp_path):
    """-lmctl (ngram_model_set_read, lm/ngram_model_set.c:185-330): two models over the medium dictionary -- the turtle LM with a word
    class defined on one of its words (forward -> the class "forward" = {forward:forward 0.7, ahead:forward 0.3}) and the medium LM --
    decoded with each member selected by -lmname (ps_init insists on one, pocketsphinx.c:400; the class member: class words resolved
    on the device's word map, psgpu_lm_tables_read_member.  A set WITHOUT a current model -- ngram_model_set_interp through the API --
    is tests/test_lm_set.py's: look-ups against the reference's, and a search with a set handle).
    Decoder B's first pass on the device (the ps_searchfuncs_t binding), the reference's lattice pass on its tables: hypothesis, path
    score and every segment equal the CPU decoder's."""
    ctl = tmp_path / "set.lmctl"
    (tmp_path / "cls.probdef").write_text("LMCLASS forward\nforward:forward 0.7\nahead:forward 0.3\nEND forward\n")
    ctl.write_text("{ %s }\n%s turtle { forward }\n%s medium\n" % (tmp_path / "cls.probdef", os.path.join(DATA, "turtle.lm.bin"), os.path.join(DATA, "medium.arpa")))
    extra = ("lmctl", str(ctl), "fwdflat", "no", "bestpath", "yes", "lmname", lmname)
    r = run("goforward.raw", 1, "psgpu_device_vtable", "yes", *extra, lm="-", dic="medium.dic")
    assert r["ok"] and r["rc"] == 0, r
    assert r["hyp_equal"] and r["seg_equal"] and r["score_cpu"] == r["score_gpu"], r
    assert r["n_seg"] > 0 and r["device_search_frames"] > 0, r


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [("fwdflat", "no", "bestpath", "no"), ()])
@pytest.mark.parametrize("what", ["librivox-0870.raw", 30.0, 60.0])
def test_dropin_device_search_vtable_long_utterances_in_one_call(what, extra, tmp_path):
    """one-call ps_decode_raw-style decodes of utterances LONGER than the 256 frame marks the reference allocates and doubles once
    per call of ngram_search_mark_bptable (ngram_search.c:184, 324-340): librivox-0870 (710 frames), 30 s and 60 s synthetic
    (3,000 / 6,000 frames), two utterances through one decoder (nrep 2), first pass only and with the reference's own later
    passes over the injected tables.  (Round 4 shipped a heap overflow here: dev_search_finish called ngram_fwdtree_finish before
    the binding had grown bp_table_idx.  tests/test_zz_asan_gpu.py runs the same path under AddressSanitizer.)"""
    if isinstance(what, float):
        from pocketsphinx_amd import synth
        raw = tmp_path / "long.raw"
        synth.utterance(11, what).tofile(str(raw))
        what = str(raw)
    r = run(what, 2, "psgpu_device_vtable", "yes", *extra)
    assert r["ok"] and r["rc"] == 0, r
    assert r["hyp_equal"] and r["seg_equal"] and r["score_cpu"] == r["score_gpu"], r
    assert r["n_frames"] > 512 and r["device_search_frames"] >= 2 * (r["n_frames"] - 1), r


@pytest.mark.gpu
@pytest.mark.parametrize("raw,chunk,extra,lm,dic", [
    ("goforward.raw", 4096, ("fwdflat", "no", "bestpath", "no"), "turtle.lm.bin", "turtle.dic"),
    ("numbers.raw", 2048, ("fwdflat", "no", "bestpath", "no"), "turtle.lm.bin", "turtle.dic"),
    ("goforward.raw", 8000, (), "turtle.lm.bin", "turtle.dic"),      # the reference's later passes at the end, partial results before
    ("numbers.raw", 4000, ("fwdflat", "no", "bestpath", "no"), "medium.arpa", "medium.dic"),
    # the full cmudict vocabulary: the search's slab layout, 1024 work-items, resumed at every read-out
    ("goforward.raw", 4096, ("fwdflat", "no", "bestpath", "no"), "big.arpa", "cmudict-en-us.dict"),
])
def test_dropin_device_search_partial_results(raw, chunk, extra, lm, dic):
    """live decoding through the device ps_searchfuncs_t: the utterance arrives `chunk` samples at a time
    (ps_process_raw without full_utt, reference src/pocketsphinx.c:1220-1257) and ps_get_hyp is asked after every piece, as
    a live application does (:1372, ngram_search_hyp, src/ngram_search.c:845).  The reference's search has then stepped
    through output_frame - pl_window frames; the binding hands the device pipeline the frames it has not seen yet, the device
    search goes on from where it stopped to as far short of the phone loop (psgpu_decode_live_step), and that table is injected.
    EVERY partial hypothesis and score equals the CPU decoder's, and so does the final result."""
    r = run(raw, 1, "psgpu_device_vtable", "yes", "chunked", str(chunk), *extra, lm=lm, dic=dic)
    assert r["ok"] and r["rc"] == 0, r
    assert r["partial_equal"] and r["partial_results"] >= 5, r
    assert r["hyp_equal"] and r["seg_equal"] and r["score_cpu"] == r["score_gpu"], r
    # the utterance ran as a live utterance of the device pipeline: a step per read-out that found new frames plus the last one,
    # and the device search stepped through every frame ONCE (not through the prefix again at every read-out)
    assert r["live_steps"] >= 5 and r["live_restarts"] == 0, r
    assert r["live_frames_searched"] == r["live_utt_frames"] > 0, r
    assert any(part.split("|")[0].strip() for part in r["last_partial_cpu"].split(";") if "|" in part), r    # (words before the end)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [("fwdflat", "no", "bestpath", "no"), ()])
def test_dropin_device_search_vtable_session(extra, tmp_path):
    """one decoder, four different utterances one after another through the device ps_searchfuncs_t: what an utterance
    inherits from the one before -- the per-state ssids of the multiplexed permanent channels (hmm_clear keeps them), the
    scorer's history slot that seeds frame 0 -- goes through the device pass and back into the reference's structures
    (integration/psgpu_device_decode.c session_push / session_pull), with and without the reference's own second and third
    pass in between.  Every utterance's hypothesis, score and segmentation equal the CPU decoder's, which carries the same."""
    ctl = tmp_path / "session.ctl"
    ctl.write_text("numbers\ngoforward\nsomething\nnumbers\n")
    r = run("@%s:%s:raw" % (ctl, DATA), 1, "psgpu_device_vtable", "yes", *extra)
    assert r["ok"] and r["rc"] == 0, r
    assert r["n_utts"] == 4 and r["hyp_equal"] and r["seg_equal"] and r["score_cpu"] == r["score_gpu"], r
    assert r["n_seg"] > 0 and r["device_search_frames"] > 0, r


@pytest.mark.gpu
@pytest.mark.parametrize("workers,extra,reps", [
    (1, ("fwdflat", "no", "bestpath", "no"), 1),
    (2, ("fwdflat", "no", "bestpath", "yes"), 1),          # each utterance's lattice pass on the host over its injected table
    (1, ("fwdflat", "no", "bestpath", "no"), 11),          # B = 66
    (2, ("fwdflat", "no", "bestpath", "no"), 86),          # B = 516
])
def test_decode_batch_api_device_first_pass(workers, extra, reps):
    """psgpu_decode_batch(..., PSGPU_BATCH_DEVICE_FIRST_PASS = 16): B utterances through ONE launch set of the device
    pipeline (front end, features, scores, phone loop, lexicon-tree search), the reference's own ps_get_hyp / ps_seg_iter
    reading each utterance's injected tables.  Every result equals a fresh unmodified CPU decoder's -- batch in order,
    reversed, and one utterance at a time."""
    r = _batch_api(workers, 16, FILES * reps, *extra)
    assert r["ok"] and r["rc"] == 0, r
    assert r["B"] == len(FILES) * reps and r["mismatch_batch"] == r["mismatch_reversed"] == r["mismatch_single"] == 0
    assert r["hyps"][0] == "go forward ten meters"


@pytest.mark.gpu
@pytest.mark.parametrize("workers,extra,reps", [
    (1, ("fwdflat", "yes", "bestpath", "no"), 1),
    (2, (), 4),                                            # the reference's defaults: its lattice pass on the host over the injected table
])
def test_decode_batch_api_device_both_passes(workers, extra, reps, monkeypatch):
    """PSGPU_BATCH_DEVICE_FIRST_PASS with -fwdflat yes and PSGPU_DEVICE_SECOND_PASS=1: BOTH search passes of the whole batch on the
    device (psgpu_decode_first_pass + psgpu_decode_second_pass), the second pass's tables injected into the worker's decoder,
    ps_get_hyp / ps_seg_iter (and -bestpath yes) the reference's own.  Every result equals a fresh unmodified CPU decoder's."""
    monkeypatch.setenv("PSGPU_DEVICE_SECOND_PASS", "1")
    r = _batch_api(workers, 16, FILES * reps, *extra)
    assert r["ok"] and r["rc"] == 0, r
    assert r["B"] == len(FILES) * reps and r["mismatch_batch"] == r["mismatch_reversed"] == r["mismatch_single"] == 0
    assert r["hyps"][0] == "go forward ten meters"


@pytest.mark.gpu
def test_decode_batch_api_device_first_pass_refuses_the_second_pass_on_the_host():
    """without PSGPU_DEVICE_SECOND_PASS the combination is refused at psgpu_batch_init (exit code 3 of the harness: the reference's
    own second pass would find no feature vectors in acmod), not at the first decode"""
    binary = os.path.join(REF, "batch_api_check")
    env = {k: v for k, v in os.environ.items() if k != "PSGPU_DEVICE_SECOND_PASS"}
    p = subprocess.run([binary, MODEL, os.path.join(DATA, "turtle.lm.bin"), os.path.join(DATA, "turtle.dic"), "1", "16",
                        os.path.join(DATA, "goforward.raw"), "--", "fwdflat", "yes", "bestpath", "no"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 3, (p.returncode, p.stdout, p.stderr[-500:])


@pytest.mark.gpu
def test_dropin_device_first_pass_refuses_other_setups():
    """attach fails loudly (exit code 3 of the harness) when the decoder is not a pass-1-only n-gram setup."""
    argv = [BIN, MODEL, os.path.join(DATA, "turtle.lm.bin"), os.path.join(DATA, "turtle.dic"),
            os.path.join(DATA, "goforward.raw"), "1", "psgpu_device_search", "yes", "fwdtree", "no"]
    p = subprocess.run(argv, capture_output=True, text=True, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stderr[-500:])


@pytest.mark.gpu
@pytest.mark.parametrize("flags,extra,reps", [
    (16, ("fwdflat", "no", "bestpath", "no"), 4),          # the whole first pass on the device, per "device"
    (0, (), 1),                                            # GMM scoring on the device, the reference's three passes
])
def test_decode_batch_api_multi_device_dispatcher(flags, extra, reps):
    """psgpu_decode_batch_multi (integration/psgpu_decode_batch.c): the utterances split into consecutive blocks, one batch
    object and host thread per entry of the device list.  This box has one GPU: the list is "0,0" -- two batch objects, two
    threads, two sets of device buffers and streams working side by side on it, the same code path as two GPUs.  Every
    utterance equals a fresh CPU decoder's, batch in order, reversed, and one at a time."""
    r = _batch_api(2, flags, FILES * reps, *extra, devices="0,0")
    assert r["ok"] and r["rc"] == 0, r
    assert r["B"] == len(FILES) * reps and r["mismatch_batch"] == r["mismatch_reversed"] == r["mismatch_single"] == 0
    assert r["hyps"][0] == "go forward ten meters"


@pytest.mark.gpu
@pytest.mark.parametrize("seconds,chunk,restarts", [(30.0, 4000, 0), (42.0, 16000, 1)])
def test_dropin_live_decode_of_a_long_utterance(tmp_path, seconds, chunk, restarts):
    """a synthetic 30 s utterance through the device ps_searchfuncs_t in 250 ms pieces with ps_get_hyp after every piece (120 read-outs)
    -- the device search steps through each of its ~3,000 frames ONCE, every partial result and the final one equal the CPU decoder's;
    and the same decode costs about what the one-call decode of the utterance does (both include the reference's own front end on
    the host): the ratio is printed, DESIGN.md quotes it.  42 s: the live utterance outgrows the capacity it was begun with
    (3,000 frames) once, is begun again with twice that and catches up -- frames searched = the utterance's + the prefix searched
    before the restart."""
    from pocketsphinx_amd import synth
    raw = tmp_path / "long.raw"
    synth.utterance(5, seconds).tofile(str(raw))
    x = ("fwdflat", "no", "bestpath", "no")
    live = run(str(raw), 2, "psgpu_device_vtable", "yes", "chunked", str(chunk), *x)
    assert live["ok"] and live["rc"] == 0 and live["partial_equal"] and live["hyp_equal"] and live["seg_equal"], live
    assert live["partial_results"] >= 2 * int(seconds * 16000 / chunk) and live["live_restarts"] == 2 * restarts, live
    if restarts == 0:
        assert live["live_frames_searched"] == live["live_utt_frames"] >= 2 * (live["n_frames"] - 1), live
    else:
        assert live["live_utt_frames"] < live["live_frames_searched"] <= 2 * live["live_utt_frames"], live
    once = run(str(raw), 2, "psgpu_device_vtable", "yes", *x)
    assert once["ok"] and once["hyp_equal"], once        # (its words differ from the live decode's: batch instead of live cepstral mean normalisation)
    print("live decode of %.0f s in %d-sample pieces: %.3f s; in one call: %.3f s; ratio %.2f (CPU decoder live: %.3f s)"
          % (seconds, chunk, live["decode_s_gpu"], once["decode_s_gpu"], live["decode_s_gpu"] / once["decode_s_gpu"], live["decode_s_cpu"]))
    assert live["decode_s_gpu"] < 3.0 * once["decode_s_gpu"], (live["decode_s_gpu"], once["decode_s_gpu"])


@pytest.mark.gpu
def test_dropin_device_search_semi_continuous_session():
    """the semi-continuous scorer (tidigits: s2_semi, 4 streams x 256 densities, 4-bit weights, s2_4x feature vectors, 5-state HMMs) behind
    the device ps_searchfuncs_t: the 31 utterances of the reference's regression list through ONE decoder, one after another -- the
    scorer's ring slot that seeds an utterance's first frame (s2_semi_mgau.c:853-860) and the multiplexed channels' ssids go through the
    device pass and back (psgpu_semi_score_batch_carry_dev, session_push / session_pull).  The feature vectors are the decoder's own
    acmod's (the binding's PCM entries serve 1s_c_d_dd only).  Every hypothesis, score and segmentation equals the CPU decoder's."""
    r = run("@%s:%s" % (os.path.join(TD, "tidigits.ctl"), TD), 1, "psgpu_device_vtable", "yes", "fwdflat", "no", "bestpath", "no", **TD_KW)
    assert r["ok"] and r["rc"] == 0, r
    assert r["n_utts"] >= 30 and r["hyp_equal"] and r["seg_equal"] and r["score_cpu"] == r["score_gpu"], r
    assert r["device_search_frames"] > 0, r


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [("fwdflat", "no", "bestpath", "no"), ()])
def test_dropin_device_search_semi_continuous_partial_results(extra, tmp_path):
    """... and in mid-utterance: three tidigits utterances through one decoder, 7 cepstral frames a piece (ps_process_cep without
    full_utt) with ps_get_hyp after every piece -- live utterances of the pipeline with the semi-continuous scorer carried from step to
    step; every partial and final result equals the CPU decoder's, each frame is searched once.  (Pieces longer than acmod's feature
    ring -- 14 frames here -- make the REFERENCE's own live decode from cepstra fail in acmod_score, "outside queue": not used.)"""
    names = [ln.strip() for ln in open(os.path.join(TD, "tidigits.ctl")) if ln.strip()][:3]
    ctl = tmp_path / "three.ctl"
    ctl.write_text("\n".join(names) + "\n")
    r = run("@%s:%s" % (ctl, TD), 1, "psgpu_device_vtable", "yes", "chunked", "7", *extra, **TD_KW)
    assert r["ok"] and r["rc"] == 0, r
    assert r["partial_equal"] and r["partial_results"] >= 9 and r["hyp_equal"] and r["seg_equal"], r
    assert r["live_frames_searched"] == r["live_utt_frames"] > 0 and r["live_restarts"] == 0, r


@pytest.mark.gpu
def test_dropin_group_of_live_decoders():
    """psgpu_live_group_create: THREE reference decoders, their n-gram searches bound to the device, ONE device pipeline in streams mode
    (integration/psgpu_device_decode.c group_step; checker oracle/streams_decode.c).  Every decoder is driven by the unmodified calls --
    ps_start_utt, ps_process_raw in pieces of its own size, ps_get_hyp after every round, ps_end_utt, the next utterance on the same
    decoder -- beside a CPU decoder fed the same pieces: every partial hypothesis and score, every final hypothesis, score and
    segmentation are equal (a stream's second utterance inherits what a decoder's does), and the device searched each frame once."""
    exe = os.path.join(REF, "streams_decode")
    if not os.path.exists(exe):
        pytest.fail("oracle/_ref/streams_decode is missing: run __graft_entry__.build() where /root/reference is present")
    argv = [exe, MODEL, os.path.join(DATA, "turtle.lm.bin"), os.path.join(DATA, "turtle.dic"), DATA, "3",
            "goforward,numbers;numbers,something;something,goforward,numbers", "2048,4096,3000", "fwdflat", "no", "bestpath", "no"]
    p = subprocess.run(argv, capture_output=True, text=True, timeout=600)
    assert p.stdout.strip(), "no output (rc %d): %s" % (p.returncode, p.stderr[-2000:])
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["ok"] and p.returncode == 0, r
    assert r["partial_results"] >= 30 and r["partial_mismatches"] == 0 and r["final_results"] == 7 and r["final_mismatches"] == 0, r
    assert r["frames_searched"] == r["frames"] > 0, r
