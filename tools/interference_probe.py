#!/usr/bin/env python3
"""Which of the other batch's stage kernels slows the tree search down?  One pipeline object runs the benchmark's 512 x 30 s step; when
its search starts (psgpu_decode_wait_scored) an AGGRESSOR -- the scorer's top-N kernels, its senone kernel, the front end, or nothing
-- is launched over and over on a second stream (a model and buffers of its own) until the search ends.  Prints the search's time per
aggressor.   IP_REPS=3"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import faulthandler; faulthandler.enable()
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import synth, decode as pdec, capi
    from test_oracle_golden import _load
    dev = torch.device("cuda", 0)
    B, sec = 512, 30.0
    L = capi.lib()
    gt = _load("fwdtree_trace_goforward.npz")
    tables = _load("en_us_ptm_tables.npz")
    mk = lambda: P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), tables, _load("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)  # noqa: E731
    pipe, other = mk(), mk()
    pipe.search_after(other)                          # (only to have the "scored" event)
    pipe.stage_timing(True)
    pcm_h = np.concatenate([synth.utterance(i % 64, sec) for i in range(B)])
    pcm = torch.from_numpy(pcm_h).to(dev)
    soff = np.arange(B + 1, dtype=np.int64) * (pcm_h.size // B)
    sa = pdec.dedicated_stream(); sb = pdec.dedicated_stream()
    sync = lambda s: capi.check(L.psgpu_stream_sync(s), "sync")  # noqa: E731
    pipe.run_dev(pcm, soff, sa); sync(sa)
    v = pipe.view()
    T = int(v.total_frames)
    # the aggressors' own inputs: the features the pipeline computed, copied; a model of their own
    model = P.PtmModel(tables)
    feat = torch.empty((T, model.veclen), dtype=torch.float32, device=dev)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(feat.data_ptr()), C.c_void_p(v.feat_dev), C.c_size_t(feat.numel() * 4), 3) == 0      # (device to device)
    off = torch.from_numpy((np.arange(B + 1) * (T // B)).astype(np.int32)).to(dev)
    tsc = torch.empty((model.n_chain, T, model.topn), dtype=torch.int32, device=dev)
    tcw = torch.empty((model.n_chain, T, model.topn), dtype=torch.uint8, device=dev)
    rows = torch.empty((T, model.n_sen), dtype=torch.int16, device=dev)
    best = torch.empty(T, dtype=torch.int32, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    fe = P.FrontEnd(_load("mfcc_en_us_goforward.npz"))
    cep = torch.empty((T + 64, 13), dtype=torch.float32, device=dev)
    foff = torch.empty(B + 1, dtype=torch.int32, device=dev)

    def topn():
        capi.check(L.psgpu_ptm_topn_dev(model.h, p(feat), p(off), B, T, None, None, p(tsc), p(tcw), sb), "topn")

    def senone():
        capi.check(L.psgpu_ptm_senone_dev(model.h, T, p(tsc), p(tcw), p(rows), p(best), 1, sb), "senone")

    def front_end():
        capi.check(L.psgpu_fe_process_utts_dev(fe.h, p(pcm), soff.ctypes.data_as(C.c_void_p), B, None, None, p(cep), p(foff), None, C.c_void_p(sb if isinstance(sb, int) else sb.value)), "fe")
    topn(); senone(); front_end(); sync(sb)
    reps = int(os.environ.get("IP_REPS", "3"))
    for name, fn, n in (("nothing", None, 0), ("top-N kernels", topn, 3), ("senone kernel", senone, 5), ("front end", front_end, 8),
                        ("nothing", None, 0)):
        ts = []
        for _ in range(reps):
            pipe.run_dev(pcm, soff, sa)
            pipe.wait_scored()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            sync(sa)
            t1 = time.perf_counter()
            sync(sb)
            t2 = time.perf_counter()
            ts.append((pipe.last_stage_ms()["search"], 1e3 * (t2 - t0)))
        print("%-14s x %d beside the search: search %s ms; aggressors done after %s ms" % (
            name, n, " ".join("%.1f" % a for a, _ in ts), " ".join("%.1f" % b for _, b in ts)), flush=True)


if __name__ == "__main__":
    main()
