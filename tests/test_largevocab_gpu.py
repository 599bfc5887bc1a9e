"""GPU parity of the device pipeline at LARGE vocabulary (en-us PTM + the synthetic 126k-unigram
LM over cmudict-en-us.dict, 134,865 dictionary words, 248 k lexicon-tree channels, `-maxhmmpf 30000`) and of the
pipeline's table-capacity / status paths.

Checker: the compiled reference (oracle/_ref/ref_decode_bench) decoding the SAME PCM on the host with the same LM and
dictionary -- word ids, start / end frames, path score, frame count, and the sizes of the back-pointer table and the
right-context score stack (reference src/ngram_search_fwdtree.c:885-1429, :1454-1495; src/ngram_search.c:377-498)."""
import json
import os
import subprocess

import numpy as np
import pytest

import pso
from test_oracle_golden import _load

pytestmark = pytest.mark.gpu


def _reference(tmp_path, pcms, lm, dic, extra=()):
    ref = os.path.join(pso.REF_DIR, "ref_decode_bench")
    if not os.path.exists(ref):
        pytest.fail("oracle/_ref (compiled reference + staged data) not built: run __graft_entry__.build() where /root/reference is present")
    raw = tmp_path / "utts.raw"
    np.concatenate(pcms).tofile(raw)
    data = os.path.join(pso.REF_DIR, "data")
    out = subprocess.run([ref, os.path.join(pso.REF_DIR, "model", "en-us"), os.path.join(data, lm), os.path.join(data, dic), str(raw),
                          str(pcms[0].size)] + list(extra), capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    return [json.loads(ln) for ln in out.stdout.strip().splitlines()][:-1]


def _same(u, r, hn, hyp, res, what):
    assert int(res[u, 3]) == 0 and int(res[u, 2]) == r["frames"], (what, res[u], r["frames"])
    assert int(res[u, 0]) == r["n_bp"] and int(res[u, 1]) == r["n_bss"], (what, res[u], r["n_bp"], r["n_bss"])
    got = [tuple(int(v) for v in hyp[u, i, :3]) for i in range(int(hn[u, 0]))]
    want = [(s[1], s[2], s[3]) for s in r["seg"]]
    assert got == want, "%s: %r vs %r" % (what, got[:6], want[:6])
    assert int(hn[u, 1]) == r["score"], what


@pytest.fixture(scope="module")
def big_task():
    """the task's tables as the product gets them: the table file integration/psgpu_export_tables wrote"""
    from pocketsphinx_amd import largevocab as lv
    if not lv.available():
        pytest.fail("table file %s not found: `make -C integration tables` (part of __graft_entry__.build())" % lv.table_path())
    return lv.tables()


@pytest.fixture(scope="module")
def big_golden(tmp_path_factory):
    """the checker's side: `ref_dump fwdtree` of the compiled reference on the same task -- its trace of goforward.raw and, as a
    cross-check of the export tool, the same static tables written by the test harness"""
    from pocketsphinx_amd.tablefile import read_psgb
    ref = pso.REF_DIR
    need = [os.path.join(ref, "ref_dump"), os.path.join(ref, "data", "big.arpa"), os.path.join(ref, "data", "cmudict-en-us.dict")]
    if not all(os.path.exists(p) for p in need):
        pytest.fail("oracle/_ref (ref_dump + big.arpa + cmudict-en-us.dict) not built: run __graft_entry__.build() where /root/reference is present")
    out = str(tmp_path_factory.mktemp("big") / "big.psgb")
    subprocess.check_call([need[0], "fwdtree", out, os.path.join(ref, "model", "en-us"), need[1], need[2],
                           os.path.join(ref, "data", "goforward.raw"), "--", "fwdflat", "no", "bestpath", "no"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    return read_psgb(out)


def test_large_vocabulary_pipeline_equals_the_reference_on_30s_utterances(tables, big_task, big_golden, tmp_path):
    """three DIFFERENT 30 s synthetic utterances (the benchmark's generator) at 134,865 words, PCM -> hypotheses on the device:
    ~9 k HMM evaluations and ~30 new back-pointers per frame, 75-90 k back-pointers and 2-2.4 M score-stack entries per
    utterance -- everything the reference's decode of the same PCM gives, incl. the table sizes"""
    from pocketsphinx_amd import largevocab as lv, synth
    ids = (1, 5, 200)
    pcms = [synth.utterance(i, 30.0) for i in ids]
    refs = _reference(tmp_path, pcms, "big.arpa", "cmudict-en-us.dict")
    p = lv.pipeline(big_task, _load("mfcc_en_us_goforward.npz"), tables)
    assert not p.search.lds_layout()
    p.run(pcms)
    hn, hyp, res = p.fetch()
    for u, r in enumerate(refs):
        _same(u, r, hn, hyp, res, "utterance %d" % ids[u])
    words = lv.words_of(big_task)
    assert len(words) >= int(big_task["par"][3]) and words[int(hyp[0, 1, 0])] == refs[0]["seg"][1][0]
    # the dump's own golden: goforward.raw through the same object (a short utterance after long ones: buffers are re-used)
    clips = _load("speech_clips.npz")
    p.run([clips["goforward"]])
    hn1, hyp1, res1 = p.fetch()
    tab = p.tables(0, res1)
    assert np.array_equal(tab["bp"], big_golden["bp"]) and np.array_equal(tab["bscore_stack"], big_golden["bscore_stack"])
    assert int(hn1[0, 1]) == int(big_golden["hyp_score"][0])
    p.close()
    # the table file against the harness's dump of the same decoder configuration: every static array identical
    for k in ("par", "node_ci", "node_child", "node_sib", "node_penult_wid", "homophone_set", "w1_wid", "dict_last", "dict_basewid",
              "rssid_ssid", "rssid_cimap", "ldiph_lc", "tp", "sseq", "unigrams", "ngram_mem", "levels", "quant", "widmap"):
        assert np.array_equal(big_task[k], big_golden[k]), k


def test_large_vocabulary_tables_grow_on_demand(tables, big_task, tmp_path):
    """the pipeline's default table allowance (16 back-pointers, 320 score-stack entries per frame: small-vocabulary figures) is
    too small at this vocabulary; the reference grows its tables on demand (ngram_search.c:449-463) -- so does fetch(), by
    repeating the search with doubled tables, and the result is the reference's"""
    from pocketsphinx_amd import largevocab as lv, synth
    pcms = [synth.utterance(9, 12.0)]
    refs = _reference(tmp_path, pcms, "big.arpa", "cmudict-en-us.dict")
    assert refs[0]["n_bp"] > 16 * refs[0]["frames"] + 2048 or refs[0]["n_bss"] > 320 * refs[0]["frames"] + 8192
    p = lv.pipeline(big_task, _load("mfcc_en_us_goforward.npz"), tables)
    p.table_capacity(16, 320, False)
    p.run(pcms)
    hn, hyp, res = p.fetch()
    v = p.view()
    assert int(res[0, 3]) == 1 and int(res[0, 2]) < refs[0]["frames"]            # ended early, said so
    assert int(res[0, 0]) <= v.bp_cap and int(res[0, 1]) <= v.bss_cap
    assert refs[0]["n_bp"] > v.bp_cap or refs[0]["n_bss"] > v.bss_cap
    p.table_capacity(16, 320, True)
    p.run(pcms)
    hn, hyp, res = p.fetch()
    assert p.tables_grown() >= 1
    _same(0, refs[0], hn, hyp, res, "after growing")
    p.close()


def test_full_tables_are_reported_not_overrun(tables, tmp_path):
    """tiny capacities on the small task, growth off: every utterance ends with status 1, the counts it reports stay within the
    capacities (nothing was written past them: the neighbouring utterance's tables hold ITS entries), and with growth on the
    same object then reproduces the reference"""
    import pocketsphinx_amd as P
    from pocketsphinx_amd import synth
    gt = _load("fwdtree_trace_goforward.npz")
    pcms = [synth.utterance(i, 6.0) for i in (2, 4, 6)]
    refs = _reference(tmp_path, pcms, "turtle.lm.bin", "turtle.dic")
    p = P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), tables, _load("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)
    p.table_capacity(1, 1, False)                         # 1 x frames + 2048 back-pointers, 1 x frames + 8192 stack entries
    p.run(pcms)
    hn, hyp, res = p.fetch()
    v = p.view()
    assert all(int(res[u, 3]) == 1 for u in range(3)), res[:, :4]
    assert all(int(res[u, 0]) <= v.bp_cap and int(res[u, 1]) <= v.bss_cap for u in range(3))
    for u in range(3):
        assert refs[u]["n_bss"] > v.bss_cap                 # (the reference needed more than the capacity: the test is one)
        assert int(res[u, 2]) < refs[u]["frames"]
    p.table_capacity(0, 0, True)
    p.run(pcms)
    hn, hyp, res = p.fetch()
    assert p.tables_grown() >= 1
    for u, r in enumerate(refs):
        _same(u, r, hn, hyp, res, "utterance %d after growing" % u)
    p.close()


def test_pipeline_equals_the_reference_on_32_of_the_benchmarks_utterances(tables, tmp_path):
    """a wider sample of the benchmark's utterance ids than test_decode_pipeline_gpu's three (round 2: an evaluation list that
    14 of the 512 overflowed was seen by the bench only): 32 ids spread over 0..511, 10 s each, in one batch"""
    import pocketsphinx_amd as P
    from pocketsphinx_amd import synth
    ids = [int(i) for i in np.linspace(0, 511, 32)]
    pcms = [synth.utterance(i, 10.0) for i in ids]
    refs = _reference(tmp_path, pcms, "turtle.lm.bin", "turtle.dic")
    gt = _load("fwdtree_trace_goforward.npz")
    p = P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), tables, _load("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)
    for lists in (False, True):
        p.score_mode(lists)
        p.run(pcms)
        hn, hyp, res = p.fetch()
        for u, r in enumerate(refs):
            _same(u, r, hn, hyp, res, "utterance %d (lists %s)" % (ids[u], lists))
    p.close()


@pytest.mark.parametrize("knob,value,status", [("PSGPU_FWDTREE_LISTED_CAP", "64", 4), ("PSGPU_FWDTREE_RC_BLOCKS", "4", 5), ("PSGPU_FWDTREE_WL_CAP", "16", 6)])
def test_slab_layout_capacities_are_reported_and_grown(tables, tmp_path, monkeypatch, knob, value, status):
    """status 4 / 5 (slab layouts): the compact channels hold the tree nodes ONE frame lists, the right-context channels come from a pool of
    blocks; both start small and grow on demand; the word level's searched array lives in LDS until a frame outgrows it (status 6: the
    slab from then on).  With the capacity cut down (the knob is read when the search is created) and growth off
    the utterances end early with that status; with growth on fetch() doubles the capacity (psgpu_fwdtree_grow) until the search gets
    through, and the result is the reference's -- and the capacity stays: the next call does not repeat."""
    import pocketsphinx_amd as P
    from pocketsphinx_amd import synth
    gt = _load("fwdtree_trace_goforward.npz")
    pcms = [synth.utterance(i, 6.0) for i in (2, 4, 6)]
    refs = _reference(tmp_path, pcms, "turtle.lm.bin", "turtle.dic")
    monkeypatch.setenv("PSGPU_FWDTREE_LAYOUT", "slab")
    monkeypatch.setenv(knob, value)
    p = P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), tables, _load("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)
    monkeypatch.delenv(knob); monkeypatch.delenv("PSGPU_FWDTREE_LAYOUT")
    assert not p.search.lds_layout()
    p.score_mode(False)
    p.table_capacity(0, 0, False)
    p.run(pcms)
    hn, hyp, res = p.fetch()
    assert all(int(res[u, 3]) == status and int(res[u, 2]) < refs[u]["frames"] for u in range(3)), res[:, :4]
    p.table_capacity(0, 0, True)
    p.run(pcms)
    hn, hyp, res = p.fetch()
    assert p.tables_grown() >= 1
    for u, r in enumerate(refs):
        _same(u, r, hn, hyp, res, "utterance %d after growing" % u)
    p.run(pcms)
    n = p.tables_grown()
    hn, hyp, res = p.fetch()
    assert p.tables_grown() == n
    for u, r in enumerate(refs):
        _same(u, r, hn, hyp, res, "utterance %d, second call" % u)
    p.close()


def test_a_full_evaluation_list_is_reported_and_recovered_from(tables, tmp_path, monkeypatch):
    """status 2: the LDS layout's evaluation list (sized by what the workgroup's LDS pool has left) is too short for a frame.
    With a list of 64 entries (PSGPU_FWDTREE_EVL_CAP, read when the search is created; the task evaluates ~240 channels a
    frame) and growth off every utterance ends early with status 2; with growth on fetch() switches the search to the slab layout
    (psgpu_fwdtree_use_slab_layout: a list that holds every channel), repeats the search, and the result is the reference's"""
    import pocketsphinx_amd as P
    from pocketsphinx_amd import synth
    gt = _load("fwdtree_trace_goforward.npz")
    pcms = [synth.utterance(i, 6.0) for i in (2, 4, 6)]
    refs = _reference(tmp_path, pcms, "turtle.lm.bin", "turtle.dic")
    monkeypatch.setenv("PSGPU_FWDTREE_EVL_CAP", "64")
    p = P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), tables, _load("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)
    monkeypatch.delenv("PSGPU_FWDTREE_EVL_CAP")
    assert p.search.lds_layout()
    p.score_mode(False)
    p.table_capacity(0, 0, False)
    p.run(pcms)
    hn, hyp, res = p.fetch()
    assert all(int(res[u, 3]) == 2 and int(res[u, 2]) < refs[u]["frames"] for u in range(3)), res[:, :4]
    p.table_capacity(0, 0, True)
    p.run(pcms)
    hn, hyp, res = p.fetch()
    assert not p.search.lds_layout() and p.tables_grown() >= 1
    for u, r in enumerate(refs):
        _same(u, r, hn, hyp, res, "utterance %d in the slab layout" % u)
    p.run(pcms)                                           # (and it stays there: no repeat this time)
    n = p.tables_grown()
    hn, hyp, res = p.fetch()
    assert p.tables_grown() == n
    for u, r in enumerate(refs):
        _same(u, r, hn, hyp, res, "utterance %d, second call" % u)
    p.close()


def test_both_passes_at_large_vocabulary_equal_the_reference(big_task):
    """BASELINE configs[2]'s shape (fwdtree + fwdflat, large LM and dictionary): two different 12 s synthetic utterances, both
    search passes on the device at 134,865 words -- the flat-lexicon pass on a vocabulary of several hundred words per
    utterance, trie language scores, scoring its own senones -- against the reference's two-pass decode of the same PCM
    (-fwdflat yes -bestpath no): words, frames, path score.  (tools/two_pass_bench.py with TP_TASK=big.)"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TP_TASK="big", TP_B="2", TP_SYNTH="12", TP_CHECK_EVERY="1")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "two_pass_bench.py")], capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["status_nonzero"] == 0
    assert j["parity"]["checked"] == 2 and j["parity"]["identical"] == 2, j["parity"]
