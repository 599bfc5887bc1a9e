#!/usr/bin/env python3
"""Throughput of the device lexicon-tree search (psgpu_fwdtree_search_dev) on replicas of a golden
trace: B utterances in one launch (one workgroup each).  Prints frames/s and the per-utterance time."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import pocketsphinx_amd as P
    from test_search_gpu import _inputs
    # SB_CASE: a golden trace; SB_LM = trie: language scores from the model's trie on the device (the only way for
    # the medium_* traces) instead of the dense table
    case = os.environ.get("SB_CASE", "goforward")
    if case == "cmudict":
        # the full-vocabulary task (134,865 words): trace made on the spot by the compiled reference (oracle/_ref)
        import subprocess
        import tempfile
        from pocketsphinx_amd.tablefile import read_psgb
        ref = os.path.join(ROOT, "oracle", "_ref")
        out = os.path.join(tempfile.mkdtemp(), "big.psgb")
        subprocess.check_call([os.path.join(ref, "ref_dump"), "fwdtree", out, os.path.join(ref, "model", "en-us"),
                               os.path.join(ref, "data", "big.arpa"), os.path.join(ref, "data", "cmudict-en-us.dict"),
                               os.path.join(ref, "data", "goforward.raw"), "--", "fwdflat", "no", "bestpath", "no"])
        g = st = read_psgb(out)
    else:
        g = np.load(os.path.join(ROOT, "tests", "golden", "fwdtree_trace_%s.npz" % case))
        static = bytes(g["static"]).decode()
        st = np.load(os.path.join(ROOT, "tests", "golden", "fwdtree_static_%s.npz" % static))
        st = {k: st[k] for k in st.files}
    lm = None
    if os.environ.get("SB_LM", "dense") == "trie" or "lm" not in st:
        lmsrc = st if "lm" not in st else np.load(os.path.join(ROOT, "tests", "golden", "lm_%s.npz" % {
            "en_us_turtle": "turtle_decoder", "tidigits": "tidigits_decoder"}[static]))
        lm = P.NGramTrieLM({k: lmsrc[k] for k in (lmsrc.files if hasattr(lmsrc, "files") else lmsrc)})
    print("case %s, %d words, %d tree nodes, LM: %s" % (case, int(g["par"][3]), int(g["par"][4] + g["par"][5]), "trie" if lm else "dense"))
    # PSGPU_FWDTREE_LAYOUT=slab in the environment forces the device-memory layout (psgpu_fwdtree_layout)
    s = P.FwdtreeSearch(st, g["par"], lm=lm)
    print("layout:", "LDS" if s.lds_layout() else "slab")
    rows, pen = _inputs(g, s.n_sen)
    import ctypes as C
    from pocketsphinx_amd import capi
    dev = torch.device("cuda", 0)
    T = rows.shape[0]
    d_rows = torch.from_numpy(rows).to(dev); d_pen1 = torch.from_numpy(pen).to(dev)
    for B in [int(x) for x in os.environ.get("SB_BATCHES", "1,64,512").split(",")]:
        # every utterance reads the same device rows (scr offsets repeat): inputs resident, as in a pipeline
        d_s = d_rows.repeat(B, 1); d_p = d_pen1.repeat(B, 1)
        nb = B
        uo = torch.from_numpy((np.arange(nb + 1) * T).astype(np.int32)).to(dev)
        bp_cap, bss_cap = 8192, 1 << 17
        bp = torch.zeros((nb, 10, bp_cap), dtype=torch.int32, device=dev); bss = torch.zeros((nb, bss_cap), dtype=torch.int32, device=dev)
        idx = torch.zeros((nb, T + 2), dtype=torch.int32, device=dev); step = torch.zeros((nb, T, 4), dtype=torch.int32, device=dev)
        res = torch.zeros((nb, 8), dtype=torch.int32, device=dev)
        p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
        sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def run():
            capi.check(capi.lib().psgpu_fwdtree_search_dev(s.h, p(d_s), C.c_int64(s.n_sen), p(d_p), p(uo), nb, T, bp_cap, bss_cap,
                                                           p(bp), p(bss), p(idx), p(step), p(res), 0, 0, None, sp), "search")
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = int(os.environ.get("SB_REPS", "2"))
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nbp = int(res[0, 0].item())
        ok = nbp == g["bp"].shape[0] and np.array_equal(bp[0, :, :nbp].cpu().numpy().T, g["bp"])
        print("B=%d (%d launches of %d): %.4f s, %.0f frames/s, %.2f ms per launch, tables ok: %s" % (
            nb * reps, reps, nb, dt, nb * reps * T / dt, 1e3 * dt / reps, ok))


if __name__ == "__main__":
    main()
