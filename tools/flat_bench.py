#!/usr/bin/env python3
"""Throughput of the device flat-lexicon second pass (psgpu_fwdflat_search_dev) on replicas of a golden two-pass
trace: B utterances in one launch (one workgroup each).  FB_CASE = a tests/golden/fwdflat_trace_*.npz name,
FB_BATCHES = 1,64,512.  Prints frames/s and whether utterance 0's tables are the reference's."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import pocketsphinx_amd as P
    from test_flat_hostsim import flat_rows
    from test_oracle_flat import load_flat
    case = os.environ.get("FB_CASE", "goforward")
    g, st, fst = load_flat(case)
    lm = P.NGramTrieLM(fst) if "lm" not in st else None
    s = P.FwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"], lm=lm)
    rows = flat_rows(g, s.n_sen)
    T = rows.shape[0]
    d_rows = torch.from_numpy(rows).to("cuda:0")
    print("case %s, %d frames, first-pass table %d entries" % (case, T, g["bp1"].shape[0]))
    for B in [int(x) for x in os.environ.get("FB_BATCHES", "1,64,512").split(",")]:
        d_s = d_rows.repeat(B, 1)
        args = (d_s, [T] * B, [g["bp1"]] * B, [g["flat_w1_ssid"]] * B)
        r = s.search(*args, bp_cap=8192, bss_cap=1 << 17)          # warm-up (includes the host-side vocabulary build)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = s.search(*args, bp_cap=8192, bss_cap=1 << 17)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = r[0]["bp"].shape == g["bp"].shape and np.array_equal(r[0]["bp"], g["bp"])
        print("B=%d: %.4f s (call incl. host vocabulary build, hand-over copies, result read-back), %.0f frames/s, tables ok: %s"
              % (B, dt, B * T / dt, ok))


if __name__ == "__main__":
    main()
