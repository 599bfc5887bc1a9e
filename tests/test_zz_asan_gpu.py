"""The host C of the binding under AddressSanitizer + UndefinedBehaviorSanitizer on the GPU box (`make -C oracle asan`:
oracle/_ref/asan/ holds the reference, integration/*.c and the three checker programs instrumented; libpsgpu.so is the
product library as built).  integration/psgpu_device_decode.c writes into the reference's own heap blocks -- back-pointer
table, score stack, frame marks (ngram_search.c:184-190, 324-340), per-frame word lists -- out of the cadence the
reference's own search keeps (round 4's overflow: a one-call utterance longer than 511 frames through the device
ps_searchfuncs_t wrote bp_table_idx[cf] past a 2 KB block): every such write is bounds-checked here, in the
device-vtable, live, session, batch and group cases, with utterances longer than 512 frames."""
import json
import os
import re
import subprocess

import pytest

import pso

REF = pso.REF_DIR
ASAN = os.path.join(REF, "asan")
MODEL = os.path.join(REF, "model", "en-us")
DATA = os.path.join(REF, "data")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0",
           UBSAN_OPTIONS="print_stacktrace=0")
# the reference's own sources shift negative values left (ps_lattice.c, ngram_search.c ...): reported by UBSan, not ours to change
OURS = re.compile(r"(integration|oracle)/[A-Za-z0-9_]+\.[ch]:\d+:\d+: runtime error")


def _run(exe, argv, timeout=900):
    path = os.path.join(ASAN, exe)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/asan/%s is missing: `make -C oracle asan` where /root/reference is present builds it "
                    "(__graft_entry__.build() tries; best effort: gcc's sanitizer runtimes may be absent)" % exe)
    p = subprocess.run([path] + [str(a) for a in argv], capture_output=True, text=True, timeout=timeout, env=ENV)
    assert "AddressSanitizer" not in p.stderr, p.stderr[max(0, p.stderr.index("AddressSanitizer") - 200):][:6000]
    bad = [ln for ln in p.stderr.splitlines() if OURS.search(ln)]
    assert not bad, "\n".join(bad[:20])
    assert p.stdout.strip(), "no output (rc %d): %s" % (p.returncode, p.stderr[-3000:])
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 0 and r["ok"], (p.returncode, r)
    return r


def _dropin(raw, nrep, *extra, lm="turtle.lm.bin", dic="turtle.dic"):
    inp = raw if os.path.isabs(raw) else os.path.join(DATA, raw)
    return _run("dropin_decode", [MODEL, os.path.join(DATA, lm), os.path.join(DATA, dic), inp, nrep] + list(extra))


@pytest.fixture(scope="module")
def long_raw(tmp_path_factory):
    from pocketsphinx_amd import synth
    p = tmp_path_factory.mktemp("asan") / "long12.raw"
    synth.utterance(5, 12.0).tofile(str(p))          # 1,200 frames: more than two doublings of the 256 frame marks
    return str(p)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [("fwdflat", "no", "bestpath", "no"), ()])
def test_asan_device_vtable_one_call_long_utterance(long_raw, extra):
    r = _dropin(long_raw, 2, "psgpu_device_vtable", "yes", *extra)
    assert r["hyp_equal"] and r["seg_equal"] and r["score_cpu"] == r["score_gpu"] and r["device_search_frames"] > 2000, r


@pytest.mark.gpu
def test_asan_device_vtable_librivox():
    r = _dropin("librivox-0870.raw", 2, "psgpu_device_vtable", "yes")
    assert r["hyp_equal"] and r["seg_equal"] and r["n_frames"] > 512, r


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [4000, 50000])
def test_asan_device_vtable_live(long_raw, chunk):
    """pieces (chunk 50,000: sparse read-outs -- about 300 frames apart, then the finish far past the last one)"""
    r = _dropin(long_raw, 2, "psgpu_device_vtable", "yes", "chunked", chunk, "fwdflat", "no", "bestpath", "no")
    assert r["partial_equal"] and r["hyp_equal"] and r["seg_equal"], r


@pytest.mark.gpu
def test_asan_gmm_shim_and_device_first_pass():
    r = _dropin("goforward.raw", 2)
    assert r["mismatching_calls"] == 0 and r["hyp_equal"] and r["seg_equal"], r
    r = _dropin("librivox-0870.raw", 1, "psgpu_device_search", "yes", "fwdflat", "no", "bestpath", "no")
    assert r["hyp_equal"] and r["seg_equal"], r


@pytest.mark.gpu
@pytest.mark.parametrize("flags,extra,second", [(16, ("fwdflat", "no", "bestpath", "no"), False), (16, (), True), (9, (), False)])
def test_asan_batch_api(long_raw, flags, extra, second):
    """psgpu_decode_batch: the whole first pass (16) / both passes (PSGPU_DEVICE_SECOND_PASS) of the batch on the device with the
    tables injected per utterance, and the GMM + front end + phone loop on the device under the reference's own search (9)"""
    files = [os.path.join(DATA, "librivox-0870.raw"), long_raw, os.path.join(DATA, "goforward.raw")]
    argv = [MODEL, os.path.join(DATA, "turtle.lm.bin"), os.path.join(DATA, "turtle.dic"), 2, flags] + files
    if extra:
        argv += ["--"] + list(extra)
    if second:
        ENV["PSGPU_DEVICE_SECOND_PASS"] = "1"
    try:
        r = _run("batch_api_check", argv)
    finally:
        ENV.pop("PSGPU_DEVICE_SECOND_PASS", None)
    assert r["B"] == 3 and r["mismatch_batch"] == r["mismatch_reversed"] == r["mismatch_single"] == 0, r


@pytest.mark.gpu
def test_asan_group_of_live_decoders():
    r = _run("streams_decode", [MODEL, os.path.join(DATA, "turtle.lm.bin"), os.path.join(DATA, "turtle.dic"), DATA, "3",
                                "goforward,numbers;numbers,something;something,goforward,numbers", "2048,4096,3000",
                                "fwdflat", "no", "bestpath", "no"])
    assert r["partial_mismatches"] == 0 and r["final_mismatches"] == 0, r
