#!/usr/bin/env python3
"""The pipeline object's two passes (bench.py's decode_two_pass leg) with host timers around each call: where a step's time goes outside
the two search kernels.  TPP_B utterances (512) of TPP_SEC seconds (30), TPP_STEPS timed steps (3).  Run it under
`rocprofv3 --kernel-trace --hip-trace --stats` for the API side."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    B = int(os.environ.get("TPP_B", "512")); sec = float(os.environ.get("TPP_SEC", "30")); steps = int(os.environ.get("TPP_STEPS", "3"))
    pcm_all = bench.synth_pcm(0, B, sec)
    n_samp = pcm_all.size // B
    import torch
    import pocketsphinx_amd as P
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    gt = bench._npz("fwdtree_trace_goforward.npz"); static = bench._npz("fwdtree_static_en_us_turtle.npz")
    gf, fst = bench._npz("fwdflat_trace_goforward.npz"), bench._npz("fwdflat_static_en_us_turtle.npz")
    pipe = P.DecodePipeline(bench._npz("mfcc_en_us_goforward.npz"), bench._npz("en_us_ptm_tables.npz"), static, gt["par"], gt)
    flat = P.FwdflatSearch(static, fst, gf["par"], gf["flat_par"], gf["flat_lwf"])
    pcm = torch.from_numpy(pcm_all).to(dev)
    soff = np.arange(B + 1, dtype=np.int64) * n_samp
    stream = torch.cuda.current_stream().cuda_stream
    pipe.run_dev(pcm, soff, stream); pipe.second_pass(flat); pipe.fetch(want_hyp=False)
    torch.cuda.synchronize()
    rows = []
    for _ in range(steps):
        t0 = time.perf_counter()
        pipe.run_dev(pcm, soff, stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        pipe.second_pass(flat); t3 = time.perf_counter()
        torch.cuda.synchronize(); t4 = time.perf_counter()
        hn, hyp, res = pipe.fetch(); t5 = time.perf_counter()
        rows.append([1e3 * (b - a) for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))])
    m = np.array(rows).mean(axis=0)
    print(json.dumps({"utterances": B, "seconds": sec, "steps": steps, "first_pass_launch_ms": round(m[0], 2), "first_pass_wait_ms": round(m[1], 2),
                      "second_pass_call_ms": round(m[2], 2), "after_second_pass_sync_ms": round(m[3], 2), "fetch_ms": round(m[4], 2),
                      "frames": int(res[:, 2].sum()), "status_nonzero": int((res[:, 3] != 0).sum())}))
    flat.close(); pipe.close()


if __name__ == "__main__":
    main()
