"""GPU parity of the flat-lexicon second-pass kernel (psgpu_fwdflat_*, the ngram_fwdflat_search replacement,
SURVEY 8a row 18) against dumps of the unmodified reference (`ref_dump fwdflat`, the goldens of
tests/test_oracle_flat.py): handed the first pass's back-pointer table and the senone scores the reference's
second pass was handed frame by frame, the kernel must produce its back-pointer table (ten columns),
right-context score stack, frame marks and per-frame best score / back-pointer count -- bit for bit, including
the float-weighted language scores.  (Sorts last on purpose: written in a session that had no GPU minutes left;
the kernel source is checked on the CPU by tests/test_flat_hostsim.py.)"""
import numpy as np
import pytest

from conftest import run_isolated
from test_flat_hostsim import check_flat, flat_rows
from test_oracle_flat import FLAT_CASES, load_flat

pytestmark = pytest.mark.gpu
ME = "test_zz_flat_gpu"                    # every test body runs in a child process (conftest.run_isolated)


@pytest.mark.parametrize("case", FLAT_CASES)
def test_fwdflat_kernel_matches_reference(case):
    run_isolated(ME, "impl_matches_reference", case)


def impl_matches_reference(case):
    import pocketsphinx_amd as P
    g, st, fst = load_flat(case)
    lm = P.NGramTrieLM(fst) if "lm" not in st else None
    s = P.FwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"], lm=lm)
    r = s.search(flat_rows(g, s.n_sen), [int(g["flat_n_steps"][0])], [g["bp1"]], [g["flat_w1_ssid"]])[0]
    check_flat(r, g, case)
    s.close()


def test_fwdflat_kernel_batch_of_utterances():
    run_isolated(ME, "impl_batch_of_utterances")


def impl_batch_of_utterances():
    import pocketsphinx_amd as P
    loaded = [load_flat(n) for n in ("goforward", "numbers")]
    g0, st, fst = loaded[0]
    s = P.FwdflatSearch(st, fst, g0["par"], g0["flat_par"], g0["flat_lwf"])
    gs = [loaded[0][0], loaded[1][0], loaded[0][0]]
    rows = [flat_rows(g, s.n_sen) for g in gs]
    out = s.search(np.concatenate(rows), [r.shape[0] for r in rows], [g["bp1"] for g in gs], [g["flat_w1_ssid"] for g in gs])
    for r, g in zip(out, gs):
        check_flat(r, g, "batch")
    s.close()


@pytest.mark.parametrize("case", ["goforward", "numbers", "man_ah_2934za"])
def test_two_passes_chained_on_the_device(case):
    """tree search -> flat search, the hand-over (back-pointer table, result record, single-phone ssids) staying in
    device buffers; the second pass must end with the reference's pass-2 tables"""
    run_isolated(ME, "impl_two_passes_chained", case)


def impl_two_passes_chained(case):
    import pocketsphinx_amd as P
    from test_oracle_golden import _load
    from test_search_gpu import _inputs
    g, st, fst = load_flat(case)
    g1 = _load("fwdtree_trace_%s.npz" % case)
    s1 = P.FwdtreeSearch(st, g1["par"])
    rows1, pen1 = _inputs(g1, s1.n_sen)
    h = {}
    r1 = s1.search(rows1, pen1, [rows1.shape[0]], handover=h)[0]
    assert np.array_equal(r1["bp"], g["bp1"]) and np.array_equal(h["w1_ssid"][0].cpu().numpy(), g["flat_w1_ssid"])
    s2 = P.FwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"])
    r2 = s2.search(flat_rows(g, s2.n_sen), [int(g["flat_n_steps"][0])], h)[0]
    check_flat(r2, g, case)
    s1.close(); s2.close()


@pytest.mark.parametrize("case", ["goforward", "numbers", "something_efwid2_sfwin8"])
def test_fwdflat_kernel_scoring_its_own_senones(case):
    """psgpu_fwdflat_search_feats_dev: features and PTM tables in, the reference's pass-2 tables out"""
    run_isolated(ME, "impl_scoring_its_own_senones", case)


def impl_scoring_its_own_senones(case):
    import pocketsphinx_amd as P
    import pso
    tables = pso.load_tables()
    g, st, fst = load_flat(case)
    model = P.PtmModel(tables)
    s = P.FwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"])
    r = s.search(g["flat_feat"], [g["flat_feat"].shape[0]], [g["bp1"]], [g["flat_w1_ssid"]], ptm=model, topn_seed=g["flat_ptm_seed"][None])[0]
    check_flat(r, g, case)
    s.close(); model.close()


@pytest.mark.parametrize("name", ["goforward", "numbers"])
def test_two_pass_decode_chain_audio_to_second_pass_backpointers(name):
    """Both search passes on the device with nothing through the host in between but the launch parameters: PCM ->
    MFCC -> features -> PTM scores (un-normalised rows + top-N lists) -> phone loop -> tree search (own active
    lists) -> flat search scoring its own senones from the features, seeded from the batch scorer's lists.  The
    second pass's back-pointer table must be the reference decoder's for the same recording (-fwdflat yes)."""
    run_isolated(ME, "impl_two_pass_chain_from_audio", name)


def impl_two_pass_chain_from_audio(name):
    import ctypes as C
    import os
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import capi
    import pso
    from test_oracle_golden import _load
    tables = pso.load_tables()
    g, st, fst = load_flat(name)
    g1 = _load("fwdtree_trace_%s.npz" % name)
    raw = os.path.join(pso.REF_DIR, "data", name + ".raw")
    assert os.path.exists(raw), "staged recordings missing (make -C oracle)"
    pcm = np.fromfile(raw, dtype=np.int16)
    dev = torch.device("cuda", 0)
    L = capi.lib()
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    fe = P.FrontEnd(_load("mfcc_en_us_goforward.npz"))
    cep, _ = fe.process_utts([pcm])
    feats = P.dynfeat_1s_c_d_dd(cep, [cep.shape[0]])
    T = feats.shape[0]
    assert T == int(g["n_frame"][0]) and np.array_equal(feats, g["flat_feat"])
    model = P.PtmModel(tables)
    d_f = torch.from_numpy(feats).to(dev)
    d_off = torch.tensor([0, T], dtype=torch.int32, device=dev)
    tsc = torch.empty((model.n_chain, T, model.topn), dtype=torch.int32, device=dev)
    tcw = torch.empty((model.n_chain, T, model.topn), dtype=torch.uint8, device=dev)
    rows = torch.empty((T, model.n_sen), dtype=torch.int16, device=dev)
    best = torch.empty(T, dtype=torch.int32, device=dev)
    capi.check(L.psgpu_ptm_score_batch_dev(model.h, p(d_f), p(d_off), 1, T, None, None, p(tsc), p(tcw), p(rows), p(best),
                                           1, sp), "score (PSGPU_PTM_RAW_SCORES)")
    # phone loop of the utterance (as tests/test_search_gpu.py)
    n_ci, window = int(g1["pl_par"][0]), int(g1["pl_par"][1])
    ctx = P.HmmContext(st["tp"], st["sseq"], model.n_sen)

    class PlPar(C.Structure):
        _fields_ = [("n_phones", C.c_int32), ("window", C.c_int32), ("beam", C.c_int32), ("pbeam", C.c_int32),
                    ("pip", C.c_int32), ("penalty_weight", C.c_double)]
    par = PlPar(n_ci, window, int(g1["pl_par"][2]), int(g1["pl_par"][3]), int(g1["pl_par"][4]), float(g1["pl_weight"][0]))
    flags = np.zeros(model.n_sen, bool)
    flags[st["sseq"][g1["pl_ssid"]].reshape(-1)] = True
    ci_list, last = [], 0
    for s_ in np.nonzero(flags)[0]:
        while s_ - last > 255:
            last += 255; ci_list.append(last)
        ci_list.append(int(s_)); last = int(s_)
    d_ssid = torch.from_numpy(g1["pl_ssid"].astype(np.uint16).view(np.int16)).to(dev)
    d_tm = torch.from_numpy(g1["pl_tmat"].astype(np.int16)).to(dev)
    d_ci = torch.from_numpy(np.array(ci_list, np.uint16).view(np.int16)).to(dev)
    pen = torch.empty((T, n_ci), dtype=torch.int32, device=dev)
    now = torch.empty((T, n_ci), dtype=torch.int32, device=dev)
    state = torch.empty((T, n_ci, 8), dtype=torch.int32, device=dev)
    L.psgpu_phone_loop_run_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                           C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
    capi.check(L.psgpu_phone_loop_run_dev(ctx.h, C.byref(par), p(d_ssid), p(d_tm), p(d_ci), len(ci_list), p(rows),
                                          model.n_sen, None, p(d_off), 1, T, p(pen), p(now), p(state), sp), "phone loop")
    # first pass on the raw rows, second pass on the features
    s1 = P.FwdtreeSearch(st, g1["par"])
    h = {}
    r1 = s1.search(rows, pen, [T], raw_scores=True, pl_window=int(g1["pl_par"][5]), handover=h)[0]
    assert np.array_equal(r1["bp"], g["bp1"])
    H = int(tables["n_fast_hist"][0])
    ts = max(x for x in range(T) if x % H == H - 1)
    seed = tcw[:, ts, :].to(torch.int32).contiguous()
    assert np.array_equal(seed.cpu().numpy().reshape(g["flat_ptm_seed"].shape), g["flat_ptm_seed"])
    s2 = P.FwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"])
    r2 = s2.search(d_f, [T], h, ptm=model, topn_seed=seed)[0]
    check_flat(r2, g, name)
    # the same taking the first pass's lists where their entries are not open (the closed form) instead of scanning codebooks
    r3 = s2.search(d_f, [T], h, ptm=model, topn_seed=seed, lists=(tsc, tcw))[0]
    check_flat(r3, g, name + ", lists taken")
    s1.close(); s2.close(); model.close()


@pytest.mark.parametrize("raw,extra,lm,dic", [
    ("goforward.raw", ("bestpath", "no"), "turtle.lm.bin", "turtle.dic"),
    ("numbers.raw", ("bestpath", "no"), "turtle.lm.bin", "turtle.dic"),
    ("goforward.raw", (), "turtle.lm.bin", "turtle.dic"),                       # + the lattice pass on the injected table
    ("goforward.raw", ("bestpath", "no"), "medium.arpa", "medium.dic"),        # trie language scores
])
def test_dropin_device_two_passes(raw, extra, lm, dic, monkeypatch):
    """Decoder B's first AND second pass on the MI355X (integration/psgpu_device_decode.c with
    PSGPU_DEVICE_SECOND_PASS=1 and -fwdflat yes): front end, features, scores, phone loop, lexicon-tree search, then
    the flat-lexicon search scoring its own senones; the second pass's back-pointer table is copied into the live
    ngram_search_t and the REFERENCE's ps_get_hyp / ps_seg_iter (and, with -bestpath yes, its lattice pass) read it.
    Hypothesis, path score and every segment must equal the CPU decoder's three-/two-pass decode of the same audio.
    (The harness is a child process.)"""
    from test_dropin_gpu import run
    monkeypatch.setenv("PSGPU_DEVICE_SECOND_PASS", "1")
    r = run(raw, 1, "psgpu_device_search", "yes", "fwdflat", "yes", *extra, lm=lm, dic=dic)
    assert r["ok"] and r["rc"] == 0, r
    assert r["hyp_equal"] and r["seg_equal"] and r["score_cpu"] == r["score_gpu"], r
    assert r["n_seg"] > 0, r


def test_fwdflat_kernel_full_cmudict_vocabulary(tmp_path):
    """134,865-word dictionary, trie language scores: the second pass on the first pass's 2,613-entry table"""
    import os
    import pso
    if not os.path.exists(os.path.join(pso.REF_DIR, "ref_dump")):
        pytest.fail("oracle/_ref (compiled reference + staged data) not built: run __graft_entry__.build() where /root/reference is present")
    run_isolated(ME, "impl_full_cmudict", str(tmp_path), timeout=1500)


def impl_full_cmudict(tmp):
    import pocketsphinx_amd as P
    import pso
    from conftest import make_big_flat_trace
    g = make_big_flat_trace(tmp)
    lm = P.NGramTrieLM(g)
    s = P.FwdflatSearch(g, g, g["par"], g["flat_par"], g["flat_lwf"], lm=lm)
    nfr = int(g["n_frame"][0])
    check_flat(s.search(flat_rows(g, s.n_sen), [nfr], [g["bp1"]], [g["flat_w1_ssid"]])[0], g, "cmudict")
    model = P.PtmModel(pso.load_tables())
    check_flat(s.search(g["flat_feat"], [nfr], [g["bp1"]], [g["flat_w1_ssid"]], ptm=model, topn_seed=g["flat_ptm_seed"][None])[0], g, "cmudict, scoring")
    s.close(); model.close()


def test_two_passes_on_different_synthetic_utterances_equal_the_reference():
    """both passes on the device for a batch of DIFFERENT utterances (the benchmark's generator, 6 s each; the second pass scoring
    its own senones with the batch scorer's lists at hand: entries taken, chains left alone, open entries scanned by a
    wavefront), every utterance against the reference's two-pass decode (-fwdflat yes -bestpath no) of the same PCM: words,
    frames, path score of the second pass's hypothesis.  (tools/two_pass_bench.py, the bench's two-pass extra, with its sample = all.)"""
    import json
    import os
    import subprocess
    import sys
    import pso
    if not os.path.exists(os.path.join(pso.REF_DIR, "ref_decode_bench")):
        pytest.fail("oracle/_ref (compiled reference + staged data) not built: run __graft_entry__.build() where /root/reference is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TP_B="24", TP_SYNTH="6.0", TP_CHECK_EVERY="1")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "two_pass_bench.py")], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["status_nonzero"] == 0
    assert j["parity"]["checked"] == 24 and j["parity"]["identical"] == 24, j["parity"]


def test_pipeline_object_runs_both_passes(tmp_path):
    """psgpu_decode_second_pass: PCM -> first pass -> second pass inside ONE pipeline object (the C ABI's batch entry), 20
    different synthetic utterances of 6 s against the reference's two-pass decode of the same PCM; the first pass's tables stay
    readable through the view; a second call on the same object (other utterances, tables re-used) as well; tiny tables grow."""
    import json
    import os
    import subprocess
    import pso
    import pocketsphinx_amd as P
    from pocketsphinx_amd import synth
    from test_oracle_golden import _load
    exe = os.path.join(pso.REF_DIR, "ref_decode_bench")
    if not os.path.exists(exe):
        pytest.fail("oracle/_ref (compiled reference + staged data) not built: run __graft_entry__.build() where /root/reference is present")
    gt, st = _load("fwdtree_trace_goforward.npz"), _load("fwdtree_static_en_us_turtle.npz")
    gf, fst = _load("fwdflat_trace_goforward.npz"), _load("fwdflat_static_en_us_turtle.npz")
    p = P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), pso.load_tables(), st, gt["par"], gt)
    flat = P.FwdflatSearch(st, fst, gf["par"], gf["flat_par"], gf["flat_lwf"])
    data = os.path.join(pso.REF_DIR, "data")

    def reference(pcms):
        raw = tmp_path / ("utts%d.raw" % len(os.listdir(str(tmp_path))))
        np.concatenate(pcms).tofile(raw)
        o = subprocess.run([exe, os.path.join(pso.REF_DIR, "model", "en-us"), os.path.join(data, "turtle.lm.bin"), os.path.join(data, "turtle.dic"),
                            str(raw), str(pcms[0].size), "--", "fwdflat", "yes", "bestpath", "no"], capture_output=True, text=True, timeout=900)
        assert o.returncode == 0, o.stderr[-2000:]
        return [json.loads(ln) for ln in o.stdout.strip().splitlines() if ln.startswith("{")][:-1]

    def check(ids, seconds):
        pcms = [synth.utterance(i, seconds) for i in ids]
        refs = reference(pcms)
        p.run(pcms)
        hn1, hyp1, res1 = p.fetch()                       # (the first pass's, before the second runs)
        p.second_pass(flat)
        hn, hyp, res = p.fetch()
        for u, r in enumerate(refs):
            assert int(res[u, 3]) == 0 and int(res[u, 2]) == r["frames"], (ids[u], res[u])
            got = [tuple(int(v) for v in hyp[u, i, :3]) for i in range(int(hn[u, 0]))]
            assert got == [(s_[1], s_[2], s_[3]) for s_ in r["seg"]], ids[u]
            assert int(hn[u, 1]) == r["score"], ids[u]
            assert int(res[u, 0]) == r["n_bp"], (ids[u], res[u], r["n_bp"])       # (the reference's table is the last pass's)
        t2 = p.tables(0, res)
        assert t2["bp"].shape[0] == int(res[0, 0]) and t2["status"] == 0
        return hn1, hn

    hn1, hn = check(list(range(20)), 6.0)
    assert any(int(hn1[u, 1]) != int(hn[u, 1]) for u in range(20))       # (the two passes' path scores differ: it did run)
    check([31, 7], 9.0)
    p.score_mode(True)                                    # (the first pass scoring from lists: no score rows in the object at all)
    check([40, 41, 42], 6.0)
    p.score_mode(False)
    p.close()
    p = P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), pso.load_tables(), st, gt["par"], gt)
    p.table_capacity(1, 1, True)                          # a new object whose tables are too small for either pass: both grow
    check([3, 4, 5], 6.0)
    assert p.tables_grown() >= 2
    # a batch of nothing but empty utterances: the first pass launches no search, the second pass of nothing is nothing --
    # empty result records, not "no first pass" (the first-pass-only path returns the same).  (A 100-sample utterance is NOT
    # empty: fe_end_utt flushes its overflow buffer as one frame, which both passes then search.)
    p.run([np.zeros(0, np.int16), np.zeros(0, np.int16)])
    p.second_pass(flat)
    hn, hyp, res = p.fetch()
    assert not res.any() and not hn[:, 0].any()
    # ... and the object still decodes afterwards
    check([8], 6.0)
    flat.close(); p.close()
