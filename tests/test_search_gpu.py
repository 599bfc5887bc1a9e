"""GPU parity of the lexicon-tree search kernel (psgpu_fwdtree_*, the ngram_fwdtree_search
replacement) against dumps of the unmodified reference: fed the senone
scores and phone-loop penalties the reference's search was handed frame by frame, the kernel must
produce the reference's back-pointer table (all ten columns), right-context score stack, per-frame
marks, and per-frame best scores -- bit for bit.  Same goldens as the CPU oracle
(tests/test_oracle_search.py): en-us + turtle LM on two recordings, forced histogram and
word-exit pruning, no phone-loop look-ahead, tidigits (5-state HMMs)."""
import numpy as np
import pytest

from test_oracle_golden import _load
from test_oracle_search import CASES

pytestmark = pytest.mark.gpu


def _inputs(g, n_sen):
    n = int(g["n_steps"][0])
    off, act, scr = g["step_act_off"], g["step_act"], g["step_scr"]
    rows = np.empty((n, n_sen), np.int16)
    for i in range(n):
        rows[i] = g["step_rest"][i]
        rows[i, act[off[i]:off[i + 1]]] = scr[off[i]:off[i + 1]]
    return rows, np.ascontiguousarray(g["step_pen"], np.int32)


def _check(r, g, what):
    n = int(g["n_steps"][0])
    assert r["status"] == 0, what
    st = r["step"]
    ref = np.stack([g["step_best"], g["step_lpbest"], g["step_bpidx"]], axis=1)
    m = min(st.shape[0], n)
    bad = np.nonzero((st[:m, :3] != ref[:m]).any(axis=1))[0]
    assert bad.size == 0, "%s: first diverging frame %d: kernel %r reference %r" % (what, bad[0], st[bad[0]], ref[bad[0]])
    assert r["n_frame"] == int(g["n_frame"][0]), what
    assert r["bp"].shape == g["bp"].shape, what
    badbp = np.nonzero((r["bp"] != g["bp"]).any(axis=1))[0]
    assert badbp.size == 0, "%s: back-pointer %d: %r vs %r" % (what, badbp[0], r["bp"][badbp[0]], g["bp"][badbp[0]])
    assert np.array_equal(r["bscore_stack"], g["bscore_stack"]), what
    assert np.array_equal(r["bp_table_idx"], g["bp_table_idx"]), what


@pytest.mark.parametrize("case", CASES)
def test_fwdtree_kernel_matches_reference(case):
    import pocketsphinx_amd as P
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    s = P.FwdtreeSearch(st, g["par"])
    rows, pen = _inputs(g, s.n_sen)
    _check(s.search(rows, pen, [rows.shape[0]])[0], g, case)
    s.close()


@pytest.mark.parametrize("cap", ["0", "2"])
@pytest.mark.parametrize("case", ["goforward", "numbers"])
def test_fwdtree_kernel_word_transitions_from_the_table(case, cap, monkeypatch):
    """the LDS layout's fall-back on frames with more exits than its LDS copy holds (PSGPU_FWDTREE_XFR_CAP, read when the search
    is created): the word transitions read the back-pointer table itself -- same tables"""
    import pocketsphinx_amd as P
    monkeypatch.setenv("PSGPU_FWDTREE_XFR_CAP", cap)
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    s = P.FwdtreeSearch(st, g["par"])
    rows, pen = _inputs(g, s.n_sen)
    _check(s.search(rows, pen, [rows.shape[0]])[0], g, "%s, xfr cap %s" % (case, cap))
    s.close()


def test_fwdtree_kernel_second_utterance_of_a_session():
    """psgpu_fwdtree_search_session_dev on the device: as tests/test_search_hostsim.py's session test (the reference's
    decoder had decoded numbers.raw before goforward.raw; raw-score mode, the kernel lists the senones itself)"""
    import pocketsphinx_amd as P
    from test_search_hostsim import _raw_rows
    g = _load("fwdtree_trace_goforward_after_numbers.npz")
    g1 = _load("fwdtree_trace_numbers.npz")
    st = _load("fwdtree_static_en_us_turtle.npz")
    s = P.FwdtreeSearch(st, g["par"])
    rows, pen = _inputs(g, s.n_sen)
    raw = _raw_rows(g, rows)
    _check(s.search(raw, pen, [rows.shape[0]], raw_scores=True, pl_window=0, mpx_in=g["mpx_init"][None])[0], g, "session")
    rows1, pen1 = _inputs(g1, s.n_sen)
    out = {}
    _check(s.search(_raw_rows(g1, rows1), pen1, [rows1.shape[0]], raw_scores=True, pl_window=0, mpx_out=out)[0], g1, "first")
    R = int(g["par"][4]); mpx = np.asarray(st["w1_mpx"]) != 0
    assert np.array_equal(out["mpx"][0][:R], g["mpx_init"][:R]) and np.array_equal(out["mpx"][0][R:][mpx], g["mpx_init"][R:][mpx])
    both = s.search(np.concatenate([raw, raw]), np.concatenate([pen, pen]), [rows.shape[0]] * 2, raw_scores=True, pl_window=0,
                    mpx_in=np.stack([g["mpx_init"], out["mpx"][0]]))
    _check(both[0], g, "batch 0"); _check(both[1], g, "batch 1")
    s.close()


def test_fwdtree_kernel_batch_of_utterances():
    """Several utterances in one launch (one workgroup each): every one equals its own golden."""
    import pocketsphinx_amd as P
    names = ["goforward", "numbers", "goforward"]
    gs = [_load("fwdtree_trace_%s.npz" % n) for n in names]
    st = _load("fwdtree_static_en_us_turtle.npz")
    s = P.FwdtreeSearch(st, gs[0]["par"])
    ins = [_inputs(g, s.n_sen) for g in gs]
    out = s.search(np.concatenate([i[0] for i in ins]), np.concatenate([i[1] for i in ins]), [i[0].shape[0] for i in ins])
    for r, g, n in zip(out, gs, names):
        _check(r, g, n)
    s.close()


@pytest.mark.parametrize("name", ["goforward", "numbers"])
def test_device_decode_chain_audio_to_backpointers(name, tables):
    """The whole first pass on the device, nothing through the host in between: 16-bit PCM -> MFCC front
    end -> 1s_c_d_dd features -> PTM senone scores (un-normalised rows) -> phone-loop search of the
    utterance -> lexicon-tree search reading those rows and penalties directly (it builds each frame's
    active senone list and normaliser itself).  The back-pointer table, score stack and frame marks must be
    the reference decoder's for the same recording (ref_dump fwdtree), and the phone-loop penalties the
    ones its search read."""
    import ctypes as C
    import os
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import capi
    import pso
    g = _load("fwdtree_trace_%s.npz" % name)
    st = _load("fwdtree_static_en_us_turtle.npz")
    raw = os.path.join(pso.REF_DIR, "data", name + ".raw")
    assert os.path.exists(raw), "staged recordings missing (make -C oracle)"
    pcm = np.fromfile(raw, dtype=np.int16)
    dev = torch.device("cuda", 0)
    L = capi.lib()
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    # front end + dynamic features
    fe = P.FrontEnd(_load("mfcc_en_us_goforward.npz"))
    cep, fo = fe.process_utts([pcm])
    feats = P.dynfeat_1s_c_d_dd(cep, [cep.shape[0]])
    T = feats.shape[0]
    assert T == int(g["n_frame"][0])
    # senone scores, un-normalised, resident
    model = P.PtmModel(tables)
    d_f = torch.from_numpy(feats).to(dev)
    d_off = torch.tensor([0, T], dtype=torch.int32, device=dev)
    tsc = torch.empty((T, model.n_chain, model.topn), dtype=torch.int32, device=dev)
    tcw = torch.empty((T, model.n_chain, model.topn), dtype=torch.uint8, device=dev)
    rows = torch.empty((T, model.n_sen), dtype=torch.int16, device=dev)
    best = torch.empty(T, dtype=torch.int32, device=dev)
    capi.check(L.psgpu_ptm_score_batch_dev(model.h, p(d_f), p(d_off), 1, T, None, None, p(tsc), p(tcw), p(rows), p(best),
                                           1, sp), "score (PSGPU_PTM_RAW_SCORES)")
    # phone loop of the utterance
    n_ci, window = int(g["pl_par"][0]), int(g["pl_par"][1])
    ctx = P.HmmContext(st["tp"], st["sseq"], model.n_sen)

    class PlPar(C.Structure):
        _fields_ = [("n_phones", C.c_int32), ("window", C.c_int32), ("beam", C.c_int32), ("pbeam", C.c_int32),
                    ("pip", C.c_int32), ("penalty_weight", C.c_double)]
    par = PlPar(n_ci, window, int(g["pl_par"][2]), int(g["pl_par"][3]), int(g["pl_par"][4]), float(g["pl_weight"][0]))
    flags = np.zeros(model.n_sen, bool)
    flags[st["sseq"][g["pl_ssid"]].reshape(-1)] = True
    ci_list, last = [], 0
    for s_ in np.nonzero(flags)[0]:
        while s_ - last > 255:
            last += 255; ci_list.append(last)
        ci_list.append(int(s_)); last = int(s_)
    d_ssid = torch.from_numpy(g["pl_ssid"].astype(np.uint16).view(np.int16)).to(dev)
    d_tm = torch.from_numpy(g["pl_tmat"].astype(np.int16)).to(dev)
    d_ci = torch.from_numpy(np.array(ci_list, np.uint16).view(np.int16)).to(dev)
    pen = torch.empty((T, n_ci), dtype=torch.int32, device=dev)
    now = torch.empty((T, n_ci), dtype=torch.int32, device=dev)
    state = torch.empty((T, n_ci, 8), dtype=torch.int32, device=dev)
    L.psgpu_phone_loop_run_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                           C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
    capi.check(L.psgpu_phone_loop_run_dev(ctx.h, C.byref(par), p(d_ssid), p(d_tm), p(d_ci), len(ci_list), p(rows),
                                          model.n_sen, None, p(d_off), 1, T, p(pen), p(now), p(state), sp), "phone loop")
    torch.cuda.synchronize()
    pen_h = pen.cpu().numpy()
    want = g["step_pen"]
    got = pen_h[np.minimum(np.arange(T) + int(g["pl_par"][5]), T - 1)]
    assert np.array_equal(got, want), "phone-loop penalties differ from what the reference search read"
    # tree search on the raw rows and the phone loop's output
    s = P.FwdtreeSearch(st, g["par"])
    r = s.search(rows, pen, [T], raw_scores=True, pl_window=int(g["pl_par"][5]))[0]
    _check(r, g, name + " (device chain)")
    # the hypothesis the host reads off the table: same path score and word boundaries as ps_get_hyp / ps_seg_iter
    score, words = P.backtrace(r, int(g["par"][20]))
    assert score == int(g["hyp_score"][0])
    assert [(sf, ef) for _, sf, ef in words] == [(int(a), int(b)) for a, b in g["seg"][:, :2]]
    s.close(); ctx.close(); model.close(); fe.close()


@pytest.mark.parametrize("case,cuts,lag", [("goforward", [1, 2, 40, 41, 150], 0), ("goforward", [30, 100, 200], 7), ("numbers", [97], 3),
                                           ("man_ah_2934za", [10, 11, 60], 2)])
def test_fwdtree_kernel_resumed_between_calls(case, cuts, lag):
    """psgpu_fwdtree_search_resume on the device: as tests/test_search_hostsim.py's test of the same name -- one utterance in several
    calls, every frame searched once, the golden's tables at the end"""
    import pocketsphinx_amd as P
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    s = P.FwdtreeSearch(st, g["par"])
    rows, pen = _inputs(g, s.n_sen)
    T = rows.shape[0]
    _check(s.search(rows, pen, [T], cuts=cuts, lag=lag)[0], g, "%s resumed at %r" % (case, cuts))
    assert s.searched == [max(c - lag, 0) for c in cuts] + [T]
    _check(s.search(rows, pen, [T])[0], g, "the call after")
    s.close()


def test_fwdtree_kernel_resumed_session_raw_scores():
    """... in raw-score mode with the look-ahead's window (the pipeline's mode) and a session's inherited channel state"""
    import pocketsphinx_amd as P
    from test_search_hostsim import _raw_rows
    g = _load("fwdtree_trace_goforward_after_numbers.npz")
    st = _load("fwdtree_static_en_us_turtle.npz")
    s = P.FwdtreeSearch(st, g["par"])
    rows, pen = _inputs(g, s.n_sen)
    raw = _raw_rows(g, rows)
    T = rows.shape[0]
    _check(s.search(raw, pen, [T], raw_scores=True, pl_window=0, mpx_in=g["mpx_init"][None], cuts=[3, 50, 51, 199], lag=4)[0], g, "session, resumed")
    assert s.searched == [0, 46, 47, 195, T]
    s.close()
