#!/usr/bin/env python3
"""Probe: does running two half-batches on two streams (two model handles, each with its own
scratch) overlap the VALU-bound top-N kernel of one with the latency-bound senone kernel of the
other?  Prints frames/s for 1 stream x 10k frames and 2 streams x 5k frames."""
import ctypes as C
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
    t = bench.load_tables()
    n_streams = int(os.environ.get("NS", 2))
    T = bench.N_UTT * bench.UTT_LEN
    feats_h = bench.synth_feats(t, T, bench.SEED)
    for ns in (1, n_streams, 4):
        per = T // ns
        n_utt = bench.N_UTT // ns
        models = [P.PtmModel(t) for _ in range(ns)]
        streams = [torch.cuda.Stream() for _ in range(ns)]
        bufs = []
        for k in range(ns):
            f = torch.from_numpy(feats_h[k * per:(k + 1) * per]).to(dev)
            off = torch.arange(0, per + 1, bench.UTT_LEN, dtype=torch.int32, device=dev)
            sc = torch.empty((per, models[k].n_chain, 4), dtype=torch.int32, device=dev)
            cw = torch.empty((per, models[k].n_chain, 4), dtype=torch.uint8, device=dev)
            out = torch.empty((per, models[k].n_sen), dtype=torch.int16, device=dev)
            bufs.append((f, off, sc, cw, out))
        torch.cuda.synchronize()

        def step():
            for k in range(ns):
                f, off, sc, cw, out = bufs[k]
                capi.check(L.psgpu_ptm_score_batch_dev(models[k].h, C.c_void_p(f.data_ptr()), C.c_void_p(off.data_ptr()),
                                                       n_utt, per, None, None, C.c_void_p(sc.data_ptr()),
                                                       C.c_void_p(cw.data_ptr()), C.c_void_p(out.data_ptr()), None, 0,
                                                       C.c_void_p(streams[k].cuda_stream)), "score")
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        K = 50
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%d stream(s) x %d frames: %.2f M frames/s (%.1f us per 10k frames)" % (ns, per, T * K / dt / 1e6, 1e6 * dt / K))
        for m in models:
            m.close()


if __name__ == "__main__":
    main()
