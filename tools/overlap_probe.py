#!/usr/bin/env python3
"""Steps in flight: does one batch's front end + scorer run beside another batch's search kernel?  The benchmark's 512 x 30 s
workload through 1, 2 or 3 pipeline objects taking turns (psgpu_decode_search_after), each on a stream with a hardware queue of
its own; per step wall time and the completion time of every step.   OP_PIPES=1,2,3  OP_STEPS=12  OP_FETCH=1 (read the
hypotheses back after every step, as the benchmark does)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import synth, decode as pdec
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
    n = int(os.environ.get("OP_STEPS", "12"))
    fetch = bool(int(os.environ.get("OP_FETCH", "1")))
    ref = None
    for n_pipe in [int(x) for x in os.environ.get("OP_PIPES", "1,2,3").split(",")]:
        pipes = [mk() for _ in range(n_pipe)]
        for q in pipes:
            q.stage_timing(True)
        streams = [pdec.dedicated_stream() for _ in range(n_pipe)]
        if n_pipe > 1:
            for k in range(n_pipe):
                pipes[k].search_after(pipes[(k - 1) % n_pipe])
        sync = lambda k: P.capi.check(P.capi.lib().psgpu_stream_sync(streams[k]), "sync")  # noqa: E731
        pause = float(os.environ.get("OP_PAUSE_US", "0")) * 1e-6

        def collect(k):
            # the step's search has finished -> the NEXT object's search is dispatched at this very instant (it waited for
            # this one's event); anything of ours dispatched in the same instant -- the read-back's copy kernels -- was seen
            # stalling behind the half-placed search for its whole duration: give the dispatcher a moment first
            sync(k)
            if n_pipe > 1 and pause > 0:
                time.sleep(pause)
            return pipes[k].fetch() if fetch else None
        for k in range(n_pipe):                       # warm (buffers, code objects)
            pipes[k].run_dev(pcm, soff, streams[k]); sync(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = []
        for k in range(n):
            j = k % n_pipe
            if k >= n_pipe:                           # the object's previous step: results to the host before it is reused
                out = collect(j)
                done.append(time.perf_counter() - t0)
            pipes[j].run_dev(pcm, soff, streams[j])
        for k in range(n, n + n_pipe):
            j = k % n_pipe
            if k - n_pipe >= 0:
                out = collect(j)
                done.append(time.perf_counter() - t0)
        dt = (time.perf_counter() - t0) / n
        if fetch:
            hn = out[0]
            if ref is None:
                ref = hn.copy()
            print("  hypotheses of the last step equal to the single-pipeline run's: %s; utterances with a non-zero status: %d"
                  % (bool(np.array_equal(ref, hn)), int((out[2][:, 3] != 0).sum())))
        print("  step completions (ms): " + " ".join("%.0f" % (1e3 * t) for t in done), flush=True)
        print("pipes %d: %.1f ms per step; stages of the last step: %s" % (n_pipe, dt * 1e3, {k: round(v, 2) for k, v in pipes[(n - 1) % n_pipe].last_stage_ms().items()}), flush=True)
        for q in pipes:
            q.close()
        for s in streams:
            pdec.free_stream(s)


if __name__ == "__main__":
    main()
