#!/usr/bin/env python3
"""hmm_vit_kernel alone over a dense arena larger than L2 + MALL (the workload of bench.py extra.hmm_vit_kernel): ms per launch,
algorithmic (86 B per HMM-frame) and line (128 B) rates.  HB_CHECK=1: the records after one launch against the
oracle's hmm_vit_eval on a sample."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import capi
    L = capi.lib()
    dev = torch.device("cuda", 0)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_en_us_3st.npz"))
    n_sen = int(g["n_sen"][0])
    ctx = P.HmmContext(g["tp"], g["sseq"], n_sen)
    n_hmm, rng = int(os.environ.get("HB_N", 8 * 1024 * 1024)), np.random.default_rng(1)
    recs = np.zeros(n_hmm, P.HMM_REC)
    recs["score"][:, :3] = -rng.integers(0, 200000, (n_hmm, 3))
    recs["history"][:, :3] = rng.integers(0, 5000, (n_hmm, 3))
    recs["senid"][:, :3] = rng.integers(0, n_sen, (n_hmm, 3))
    recs["tmatid_mpx"] = rng.integers(0, g["tp"].shape[0], n_hmm)
    d_recs = torch.from_numpy(recs.view(np.uint8).reshape(n_hmm, 64)).to(dev)
    d_scr = torch.from_numpy(np.ascontiguousarray(g["senscr"][0])).to(dev)
    d_best = torch.full((1,), -0x20000000, dtype=torch.int32, device=dev)

    def step():
        capi.check(L.psgpu_hmm_vit_eval_dev(ctx.h, C.c_void_p(d_recs.data_ptr()), None, n_hmm, None, C.c_void_p(d_scr.data_ptr()), n_sen,
                                            C.c_void_p(d_best.data_ptr()), sp), "hmm")
    for _ in range(2):
        step()
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.psgpu_event_create(C.byref(e0)); L.psgpu_event_create(C.byref(e1))
    K = 10
    L.psgpu_event_record(e0, sp)
    for _ in range(K):
        step()
    L.psgpu_event_record(e1, sp)
    ms = C.c_float()
    L.psgpu_event_elapsed_ms(e0, e1, C.byref(ms))
    per = ms.value / K * 1e-3
    print("hmm_vit_kernel: %d HMMs, %.4f ms per launch, algorithmic %.0f GB/s (%.3f of 8 TB/s), lines %.0f GB/s" % (
        n_hmm, per * 1e3, 86 * n_hmm / per / 1e9, 86 * n_hmm / per / 1e9 / 8000.0, 128 * n_hmm / per / 1e9))


if __name__ == "__main__":
    main()
