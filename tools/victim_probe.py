#!/usr/bin/env python3
"""How long does a small kernel on a second stream take beside the pipeline's search kernel, as a function of the time since the
search started?  One pipeline object runs the benchmark's 512 x 30 s step; psgpu_decode_wait_scored returns when the stages
before the search are done (the search starts then); the host sleeps X ms and launches a victim (a fill of N elements) on a
dedicated stream, timed with events and on the host."""
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
    B, sec = 512, 30.0
    gt = _load("fwdtree_trace_goforward.npz")
    tables = _load("en_us_ptm_tables.npz")
    pipe = P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), tables, _load("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)
    other = P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), tables, _load("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)
    pipe.search_after(other)                          # (only to have the "scored" event)
    pcm_h = np.concatenate([synth.utterance(i % 64, sec) for i in range(B)])
    pcm = torch.from_numpy(pcm_h).to(dev)
    soff = np.arange(B + 1, dtype=np.int64) * (pcm_h.size // B)
    sa = pdec.dedicated_stream()
    sb = torch.cuda.ExternalStream(pdec.dedicated_stream(), device=dev)
    buf = torch.empty(1 << 24, dtype=torch.int32, device=dev)
    host = torch.empty(1 << 16, dtype=torch.int32).pin_memory()
    pipe.run_dev(pcm, soff, sa); P.capi.check(P.capi.lib().psgpu_stream_sync(sa), "sync")
    with torch.cuda.stream(sb):
        buf[:256].fill_(1); host.copy_(buf[:1 << 16], non_blocking=True)
    torch.cuda.synchronize()
    for what, n in (("fill 256", 256), ("fill 16M", 1 << 24), ("d2h 256 KB", -1)):
        for x_ms in (0.0, 0.3, 1.0, 3.0, 10.0, 30.0, 60.0):
            pipe.run_dev(pcm, soff, sa)
            pipe.wait_scored()
            t_s = time.perf_counter()
            if x_ms:
                time.sleep(x_ms * 1e-3)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            h0 = time.perf_counter()
            with torch.cuda.stream(sb):
                e0.record()
                if n > 0:
                    buf[:n].fill_(2)
                else:
                    host.copy_(buf[:1 << 16], non_blocking=True)
                e1.record()
            sb.synchronize()
            h1 = time.perf_counter()
            P.capi.check(P.capi.lib().psgpu_stream_sync(sa), "sync")
            h2 = time.perf_counter()
            print("%-10s launched %5.1f ms after the search started: events %8.3f ms, host %8.3f ms; search ended %6.1f ms after its start"
                  % (what, 1e3 * (h0 - t_s), e0.elapsed_time(e1), 1e3 * (h1 - h0), 1e3 * (h2 - t_s)), flush=True)
    pipe.close(); other.close()


if __name__ == "__main__":
    main()
