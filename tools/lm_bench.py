#!/usr/bin/env python3
"""Throughput of the device language-model look-up (psgpu_lm_tg_score_dev) on the recorded queries of a
fixture, tiled; checks the answers of the first tile against the reference's."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import pocketsphinx_amd as P
    for name in os.environ.get("LB_CASES", "synthetic,turtle_decoder").split(","):
        g = np.load(os.path.join(ROOT, "tests", "golden", "lm_%s.npz" % name))
        lm = P.NGramTrieLM({k: g[k] for k in g.files})
        q = g["queries"]
        reps = max(1, (4 << 20) // len(q))
        d_q = torch.from_numpy(np.tile(q, (reps, 1))).cuda()
        sc, nu = lm.tg_score(d_q)
        torch.cuda.synchronize()
        ok = np.array_equal(sc[:len(q)].cpu().numpy(), g["scores"]) and np.array_equal(sc[-len(q):].cpu().numpy(), g["scores"])
        t0 = time.perf_counter()
        for _ in range(5):
            lm.tg_score(d_q)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("lm %s: %d look-ups in %.3f ms = %.1f M look-ups/s (order %d, %d words), answers ok: %s" % (
            name, len(d_q), 1e3 * dt, len(d_q) / dt / 1e6, lm.order, lm.n_words, ok))


if __name__ == "__main__":
    main()
