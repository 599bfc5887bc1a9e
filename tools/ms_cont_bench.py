#!/usr/bin/env python3
"""Batched multi-stream scorer on a synthetic fully continuous model (.cont. mapping) of
en-us size -- the workload of bench.py extra.ms_continuous, alone, for profiling:
  rocprofv3 --kernel-trace --stats -d gpurun_out/msprof -o ms -- python tools/ms_cont_bench.py"""
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
    n_sen = int(os.environ.get("MS_SEN", 5126)); n_den = int(os.environ.get("MS_DEN", 16)); LL = 39
    n_fr = int(os.environ.get("MS_FRAMES", 1600)); K = int(os.environ.get("MS_STEPS", 5))
    L = capi.lib()
    dev = torch.device("cuda", 0)
    z = np.load(os.path.join(ROOT, "tests", "golden", "en_us_ptm_tables.npz"))
    rng = np.random.default_rng(9)
    mt = dict(n_mgau=np.array([n_sen]), n_feat=np.array([1]), n_density=np.array([n_den]),
              n_sen=np.array([n_sen]), max_topn=np.array([4]), aw=np.array([1]),
              featlen=np.array([LL], np.int32),
              mean=rng.standard_normal(n_sen * n_den * LL).astype(np.float32),
              var=np.floor(np.exp(rng.uniform(0, 12, n_sen * n_den * LL))).astype(np.float32),
              det=np.floor(rng.uniform(-500000, 400000, (n_sen, 1, n_den))).astype(np.float32),
              pdf=rng.integers(0, 256, (n_sen, 1, n_den)).astype(np.uint8),
              sen2mgau=np.arange(n_sen, dtype=np.uint32), logadd=z["logadd8"],
              logadd_size=np.array([int(z["logadd8"].size)]), logadd_width=np.array([1]),
              log_zero=np.array([-524288]))
    ms = P.MsMgau(mt)
    f = torch.from_numpy(rng.standard_normal((n_fr, LL)).astype(np.float32)).to(dev)
    nl = n_fr * ms.n_mgau * ms.n_feat * ms.topn
    ids = torch.empty(nl, dtype=torch.int32, device=dev)
    dist = torch.empty(nl, dtype=torch.float32, device=dev)
    scr = torch.empty((n_fr, ms.n_sen), dtype=torch.int16, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def step():
        capi.check(L.psgpu_ms_score_batch_dev(ms.h, C.c_void_p(f.data_ptr()), n_fr, 
                                              C.c_void_p(ids.data_ptr()) if os.environ.get("MS_LISTS") else None,
                                              C.c_void_p(dist.data_ptr()) if os.environ.get("MS_LISTS") else None,
                                              C.c_void_p(scr.data_ptr()), sp), "ms")
    step()
    capi.check(L.psgpu_ms_batch_check(ms.h, sp), "check")
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.psgpu_event_create(C.byref(e0)); L.psgpu_event_create(C.byref(e1))
    L.psgpu_event_record(e0, sp)
    for _ in range(K):
        step()
    L.psgpu_event_record(e1, sp)
    ms_ = C.c_float()
    L.psgpu_event_elapsed_ms(e0, e1, C.byref(ms_))
    print("frames/s %.1f  ms/step %.4f  (%d senones x %d densities x %d dims, %d frames)" % (
        n_fr * K / (ms_.value * 1e-3), ms_.value / K, n_sen, n_den, LL, n_fr))


if __name__ == "__main__":
    main()
