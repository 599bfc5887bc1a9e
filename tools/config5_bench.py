#!/usr/bin/env python3
"""BASELINE configs[4] at full size on one GPU, device side only: 512 utterances x 30 s of
synthetic 16 kHz PCM -> MFCC front end -> 1s_c_d_dd features -> en-us PTM senone scores
(1.536 M frames, 15.7 GB of int16 scores resident in HBM).  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import capi
    L = capi.lib()
    dev = torch.device("cuda", 0)
    n_utt = int(os.environ.get("C5_UTTS", 512)); secs = float(os.environ.get("C5_SECS", 30.0))
    t = bench.load_tables()
    model = P.PtmModel(t)
    g = np.load(os.path.join(ROOT, "tests", "golden", "mfcc_en_us_goforward.npz"))
    fe = P.FrontEnd({k: g[k] for k in g.files})
    fsz, fsh = int(g["par"][0]), int(g["par"][1])
    n_fr = int(secs * 100)
    n_samp = fsz + fsh * (n_fr - 2)
    assert fe.n_frames(n_samp) == n_fr
    rng = np.random.default_rng(1)
    base = (2000 * rng.standard_normal(n_samp)).astype(np.int16)
    tt = np.arange(n_samp)
    pcm = torch.empty(n_utt * n_samp, dtype=torch.int16, device=dev)
    for u in range(n_utt):         # noise + an utterance-specific gated tone, generated on the host in slices
        tone = (6000 * np.sin(2 * np.pi * (150 + u % 400) * tt / 16000.0) * (np.sin(2 * np.pi * (2 + u % 5) * tt / 16000.0) > 0))
        pcm[u * n_samp:(u + 1) * n_samp] = torch.from_numpy((base + tone).astype(np.int16)).to(dev)
    soff = np.arange(n_utt + 1, dtype=np.int64) * n_samp
    T = n_utt * n_fr
    cep = torch.empty((T, fe.out_dim), dtype=torch.float32, device=dev)
    ft = torch.empty((T, 3 * fe.out_dim), dtype=torch.float32, device=dev)
    foff = torch.empty(n_utt + 1, dtype=torch.int32, device=dev)
    tsc = torch.empty((T, model.n_chain, model.topn), dtype=torch.int32, device=dev)
    tcw = torch.empty((T, model.n_chain, model.topn), dtype=torch.uint8, device=dev)
    scr = torch.empty((T, model.n_sen), dtype=torch.int16, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.psgpu_fe_process_utts_dev.argtypes = [C.c_void_p] * 10
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731

    def run():
        capi.check(L.psgpu_fe_process_utts_dev(fe.h, p(pcm), soff.ctypes.data_as(C.c_void_p), n_utt, None, None, p(cep),
                                               p(foff), None, sp), "fe")
        capi.check(L.psgpu_feat_1s_c_d_dd_dev(p(cep), p(foff), n_utt, fe.out_dim, p(ft), sp), "feat")
        capi.check(L.psgpu_ptm_score_batch_dev(model.h, p(ft), p(foff), n_utt, T, None, None, p(tsc), p(tcw), p(scr),
                                               None, 0, sp), "score")
    run()
    torch.cuda.synchronize()
    K = 3
    t0 = time.perf_counter()
    for _ in range(K):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    chk = int(scr[::100003 // 7].to(torch.int32).min(dim=1).values.abs().sum().item())
    print(json.dumps({"workload": "configs[4]: %d utterances x %.0f s synthetic PCM -> en-us PTM senone scores" % (n_utt, secs),
                      "frames": T, "audio_s": n_utt * n_samp / 16000.0, "seconds": round(dt, 5),
                      "frames_per_s": round(T / dt, 1), "xrt": round(dt / (n_utt * n_samp / 16000.0), 9),
                      "scores_bytes": T * model.n_sen * 2, "normalisation_check": chk}))


if __name__ == "__main__":
    main()
