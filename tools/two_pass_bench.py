#!/usr/bin/env python3
"""Both search passes on the device for a batch of utterances: PCM -> MFCC -> features -> PTM scores (un-normalised rows
+ top-N lists) -> phone loop -> lexicon-tree search -> flat-lexicon search scoring its own senones (seeded from the batch
scorer's lists), back-pointer tables of both passes left on the device.  Prints ONE JSON line.  Run by bench.py in a
child process (the second-pass kernel had not run on a GPU when this was written: a fault must not take the headline
measurement with it).   TP_B = utterances (default 256)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import capi
    G = os.path.join(ROOT, "tests", "golden")
    ld = lambda n: (lambda z: {k: z[k] for k in z.files})(np.load(os.path.join(G, n)))  # noqa: E731
    gm, t = ld("mfcc_en_us_goforward.npz"), ld("en_us_ptm_tables.npz")
    big = os.environ.get("TP_TASK") == "big"              # the 134,865-word task (big.arpa + cmudict, trie LM on the device): configs[2]'s shape
    lm = None
    if big:
        from pocketsphinx_amd import largevocab as lv
        gt = st = gf = fst = lv.tables()        # the table file (integration/psgpu_export_tables): both passes' tables, no trace
        lm = P.NGramTrieLM(gt)
    else:
        gt, st = ld("fwdtree_trace_goforward.npz"), ld("fwdtree_static_en_us_turtle.npz")
        gf, fst = ld("fwdflat_trace_goforward.npz"), ld("fwdflat_static_en_us_turtle.npz")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = capi.lib()
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    q = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    B = int(os.environ.get("TP_B", "256"))
    model = P.PtmModel(t)
    fe = P.FrontEnd(gm)
    s1 = P.FwdtreeSearch(st, gt["par"], lm=lm)
    s2 = P.FwdflatSearch(st, fst, gf["par"], gf["flat_par"], gf["flat_lwf"], lm=lm)
    ctx = P.HmmContext(st["tp"], st["sseq"], model.n_sen)
    pcm1 = gm["pcm"].astype(np.float32)
    rng = np.random.default_rng(3)
    if os.environ.get("TP_SYNTH"):                                 # DIFFERENT utterances: the benchmark's generator, TP_SYNTH seconds each
        from pocketsphinx_amd import synth
        pcm_h = np.concatenate([synth.utterance(i, float(os.environ["TP_SYNTH"])) for i in range(B)])
        ns = pcm_h.size // B
    else:
        gains = rng.uniform(0.6, 1.0, B); gains[0] = 1.0          # utterance 0 is the bundled recording itself
        pcm_h = np.concatenate([(pcm1 * g_).astype(np.int16) for g_ in gains])
        ns = pcm1.size
    Tu = fe.n_frames(ns); Tn = B * Tu
    soff = (np.arange(B + 1, dtype=np.int64) * ns)
    pcm = torch.from_numpy(pcm_h).to(dev)
    cep = torch.empty((Tn, fe.out_dim), dtype=torch.float32, device=dev)
    ft = torch.empty((Tn, 3 * fe.out_dim), dtype=torch.float32, device=dev)
    foff = torch.empty(B + 1, dtype=torch.int32, device=dev)
    tsc = torch.empty((model.n_chain, Tn, model.topn), dtype=torch.int32, device=dev)
    tcw = torch.empty((model.n_chain, Tn, model.topn), dtype=torch.uint8, device=dev)      # chain-major (psgpu.h)
    rows = torch.empty((Tn, model.n_sen), dtype=torch.int16, device=dev)
    bst = torch.empty(Tn, dtype=torch.int32, device=dev)
    n_ci, window = int(gt["pl_par"][0]), int(gt["pl_par"][1])

    class PlPar(C.Structure):
        _fields_ = [("n_phones", C.c_int32), ("window", C.c_int32), ("beam", C.c_int32), ("pbeam", C.c_int32),
                    ("pip", C.c_int32), ("penalty_weight", C.c_double)]
    ppar = PlPar(n_ci, window, int(gt["pl_par"][2]), int(gt["pl_par"][3]), int(gt["pl_par"][4]), float(gt["pl_weight"][0]))
    fl = np.zeros(model.n_sen, bool); fl[st["sseq"][gt["pl_ssid"]].reshape(-1)] = True
    cil, last = [], 0
    for s_ in np.nonzero(fl)[0]:
        while s_ - last > 255:
            last += 255; cil.append(last)
        cil.append(int(s_)); last = int(s_)
    d_ssid = torch.from_numpy(gt["pl_ssid"].astype(np.uint16).view(np.int16)).to(dev)
    d_tm = torch.from_numpy(gt["pl_tmat"].astype(np.int16)).to(dev)
    d_ci = torch.from_numpy(np.array(cil, np.uint16).view(np.int16)).to(dev)
    pen = torch.empty((Tn, n_ci), dtype=torch.int32, device=dev)
    now = torch.empty((Tn, n_ci), dtype=torch.int32, device=dev)
    pstate = torch.empty((Tn, n_ci, 8), dtype=torch.int32, device=dev)
    L.psgpu_fe_process_utts_dev.argtypes = [C.c_void_p] * 10
    L.psgpu_phone_loop_run_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                           C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
    H = int(t["n_fast_hist"][0])
    ts = max(x for x in range(Tu) if x % H == H - 1)
    lens = [Tu] * B
    bp_cap, bss_cap = max(4096, (64 if big else 24) * Tu + 2048), max(65536, (1600 if big else 640) * Tu + 8192)     # (per-utterance table capacities of both passes)
    out = {}

    def two_pass():
        capi.check(L.psgpu_fe_process_utts_dev(fe.h, q(pcm), soff.ctypes.data_as(C.c_void_p), B, None, None, q(cep), q(foff), None, sp), "fe")
        capi.check(L.psgpu_feat_1s_c_d_dd_dev(q(cep), q(foff), B, fe.out_dim, q(ft), sp), "feat")
        capi.check(L.psgpu_ptm_score_batch_dev(model.h, q(ft), q(foff), B, Tn, None, None, q(tsc), q(tcw), q(rows), q(bst), 1, sp), "score")
        capi.check(L.psgpu_phone_loop_run_dev(ctx.h, C.byref(ppar), q(d_ssid), q(d_tm), q(d_ci), len(cil), q(rows),
                                              model.n_sen, None, q(foff), B, Tn, q(pen), q(now), q(pstate), sp), "phone loop")
        h = {}
        torch.cuda.synchronize(); ta = time.perf_counter()
        r1 = s1.search(rows, pen, lens, bp_cap=bp_cap, bss_cap=bss_cap, raw_scores=True, pl_window=int(gt["pl_par"][5]), handover=h)
        torch.cuda.synchronize(); tb = time.perf_counter()
        seed = tcw[:, ts::Tu, :].permute(1, 0, 2).to(torch.int32).contiguous()          # [B][n_chain][topn]
        r2 = s2.search(ft, lens, h, bp_cap=bp_cap, bss_cap=bss_cap, ptm=model, topn_seed=seed,
                       lists=None if os.environ.get("TP_NO_LISTS") else (tsc, tcw))
        torch.cuda.synchronize(); tc = time.perf_counter()
        return r1, r2, tb - ta, tc - tb

    two_pass()
    t0 = time.perf_counter()
    r1, r2, d1, d2 = two_pass()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fin = int(gt["par"][20])
    golden_applies = (not os.environ.get("TP_SYNTH")) and "bp1" in gf
    w0 = [w for w, _, _ in P.backtrace(r2[0], fin)[1]]
    out["task"] = "134,865 words (big.arpa + cmudict-en-us.dict, trie LM)" if big else "115 words (turtle)"
    # parity of the sampled utterances: the compiled reference decodes the SAME PCM with both passes (-fwdflat yes -bestpath no,
    # a new decoder's state per utterance) -- words, frame boundaries and path score of the second pass's hypothesis
    ids = list(range(0, B, int(os.environ.get("TP_CHECK_EVERY", "17"))))
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "release", "ref_decode_bench")        # (upstream's Release flags: `make -C oracle release`)
    if not os.path.exists(ref_exe):
        ref_exe = os.path.join(ROOT, "oracle", "_ref", "ref_decode_bench")
    parity = {"checked": 0, "note": "oracle/_ref/ref_decode_bench not built"}
    if os.path.exists(ref_exe):
        import subprocess
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".raw", delete=False) as fh:
            for u in ids:
                pcm_h[u * ns:(u + 1) * ns].tofile(fh)
            path = fh.name
        try:
            ref = os.path.join(ROOT, "oracle", "_ref")
            o = subprocess.run([ref_exe, os.path.join(ref, "model", "en-us"), os.path.join(ref, "data", "big.arpa" if big else "turtle.lm.bin"),
                                os.path.join(ref, "data", "cmudict-en-us.dict" if big else "turtle.dic"), path, str(ns), "--", "fwdflat", "yes",
                                "bestpath", "no"], capture_output=True, text=True, timeout=1800)
            tot = json.loads(o.stdout.strip().splitlines()[-1])
            out["reference"] = {"frames_per_s": tot.get("frames_per_s"), "cpu_s": tot.get("cpu_s"), "what": "the compiled reference, one thread, "
                                "-fwdflat yes -bestpath no, on the sampled utterances (its own timing line)"}
            lines = [json.loads(ln) for ln in o.stdout.strip().splitlines() if ln.startswith("{")][:-1]
        finally:
            os.unlink(path)
        bad = []
        for u, r in zip(ids, lines):
            score, words = P.backtrace(r2[u], fin)
            if words != [(s_[1], s_[2], s_[3]) for s_ in r["seg"]] or score != r["score"]:
                bad.append(u)
        parity = {"checked": len(lines), "identical": len(lines) - len(bad), "mismatching_utterances": bad,
                  "what": "second-pass hypothesis (word ids, start / end frames, path score) of every sampled utterance vs the reference's "
                          "two-pass decode of the same PCM"}
    out.update(utterances=B, frames=Tn, audio_s=round(B * ns / 16000.0, 1), seconds=round(dt, 5), frames_per_s=round(Tn / dt, 1),
               xrt=round(dt / (B * ns / 16000.0), 8), first_pass_call_s=round(d1, 5), second_pass_call_s=round(d2, 5),
               second_pass_frames_per_s=round(Tn / d2, 1),
               status_nonzero=int(sum(r["status"] != 0 for r in r1) + sum(r["status"] != 0 for r in r2)),
               # (the recorded tables of goforward.raw apply when utterance 0 IS that recording decoded with the golden's task; else null)
               utt0_first_pass_table_is_reference=(bool(r1[0]["bp"].shape == gf["bp1"].shape and np.array_equal(r1[0]["bp"], gf["bp1"]))
                                                   if golden_applies else None),
               utt0_second_pass_table_is_reference=(bool(r2[0]["bp"].shape == gf["bp"].shape and np.array_equal(r2[0]["bp"], gf["bp"]))
                                                    if golden_applies else None),
               parity=parity, words_in_hyp=len(w0),
               what="PCM -> MFCC -> features -> PTM scores -> phone loop -> lexicon-tree search -> flat-lexicon search scoring its own "
                    "senones; call times include the Python wrappers' result read-back and, for the second pass, the host-side "
                    "vocabulary build from the first pass's table")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
