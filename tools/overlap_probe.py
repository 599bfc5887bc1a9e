#!/usr/bin/env python3
"""Do a batch's scorer kernels run beside another batch's search kernel?  Two pipeline objects on two streams, the benchmark's
512 x 30 s workload: per step wall time with 1 and 2 steps in flight, with and without stage-timing events, and the
completion order of stream B's whole step relative to stream A's search."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import synth
    import pso
    from test_oracle_golden import _load
    dev = torch.device("cuda", 0)
    B = int(os.environ.get("OP_B", "512")); sec = float(os.environ.get("OP_SEC", "30"))
    gt = _load("fwdtree_trace_goforward.npz")
    tables = _load("en_us_ptm_tables.npz")
    mk = lambda: P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), tables, _load("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)  # noqa: E731
    pcm_h = np.concatenate([synth.utterance(i % 64, sec) for i in range(B)])
    n_samp = pcm_h.size // B
    pcm = torch.from_numpy(pcm_h).to(dev)
    soff = np.arange(B + 1, dtype=np.int64) * n_samp
    for n_pipe, timing, prio in ((1, False, False), (2, False, False), (2, True, False), (2, False, True)):
        pipes = [mk() for _ in range(n_pipe)]
        streams = [torch.cuda.Stream(device=dev, priority=(-1 if (prio and k == 1) else 0)) for k in range(n_pipe)]
        for q in pipes:
            q.stage_timing(timing)
        n = 8
        for k in range(n_pipe):                       # warm
            pipes[k].run_dev(pcm, soff, streams[k].cuda_stream); streams[k].synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            pipes[k % n_pipe].run_dev(pcm, soff, streams[k % n_pipe].cuda_stream)
            if k >= n_pipe - 1:
                streams[(k - (n_pipe - 1)) % n_pipe].synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("pipes %d stage_timing %s high-priority-second %s: %.1f ms per step" % (n_pipe, timing, prio, dt * 1e3), flush=True)
        for q in pipes:
            q.close()


if __name__ == "__main__":
    main()
