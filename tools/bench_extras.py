#!/usr/bin/env python3
"""Secondary measurements of bench.py (never `value`): the scorer's PCIe-inclusive rate, the Viterbi-step kernel's own
roofline, the multi-stream / continuous / semi-continuous scorers, the audio-to-scores chain.  Called in-process by
bench.py at N = 1."""
import ctypes as C
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_UTT, UTT_LEN = 40, 250
HBM_PEAK_GBS = 8000.0


def extras(P, capi, L, model, t, feats_h, dev, sp):
    """Secondary measurements (never `value`): the PCIe-inclusive rate of the
    host-buffer entry point and the Viterbi-step kernel's own roofline."""
    import torch
    out = {}
    # (1) host buffers in, host buffers out: psgpu_ptm_score_batch (H2D + 2 kernels + D2H of 102 MB)
    sc = P.PtmMgau(model)
    lens = [UTT_LEN] * N_UTT
    sc.score_utts(feats_h, lens, want_topn=False)
    t0 = time.perf_counter()
    for _ in range(3):
        sc.score_utts(feats_h, lens, want_topn=False)
    out["pcie_inclusive_frames_per_s"] = round(3 * feats_h.shape[0] / (time.perf_counter() - t0), 1)
    # (2) hmm_vit_kernel over a dense arena larger than L2+MALL: B_v = 86 B per HMM-frame
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_en_us_3st.npz"))
        n_sen = int(g["n_sen"][0])
        ctx = P.HmmContext(g["tp"], g["sseq"], n_sen)
        n_hmm, rng = 8 * 1024 * 1024, np.random.default_rng(1)
        recs = np.zeros(n_hmm, P.HMM_REC)
        recs["score"][:, :3] = -rng.integers(0, 200000, (n_hmm, 3))
        recs["history"][:, :3] = rng.integers(0, 5000, (n_hmm, 3))
        recs["senid"][:, :3] = rng.integers(0, n_sen, (n_hmm, 3))
        recs["tmatid_mpx"] = rng.integers(0, g["tp"].shape[0], n_hmm)
        d_recs = torch.from_numpy(recs.view(np.uint8).reshape(n_hmm, 64)).to(dev)
        d_scr = torch.from_numpy(np.ascontiguousarray(g["senscr"][0])).to(dev)
        d_best = torch.full((1,), -0x20000000, dtype=torch.int32, device=dev)

        def step():
            capi.check(L.psgpu_hmm_vit_eval_dev(ctx.h, C.c_void_p(d_recs.data_ptr()), None, n_hmm, None,
                                                C.c_void_p(d_scr.data_ptr()), n_sen,
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
        out["hmm_vit_kernel"] = {
            "hmm_frames_per_s": round(n_hmm / per, 1), "n_hmm": n_hmm, "ms_per_launch": round(per * 1e3, 4),
            "roofline": {"bound": "hbm", "achieved": round(86 * n_hmm / per / 1e9, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(86 * n_hmm / per / 1e9 / HBM_PEAK_GBS, 5),
                         "line_traffic_GBs": round(128 * n_hmm / per / 1e9, 2)}}
        L.psgpu_event_destroy(e0); L.psgpu_event_destroy(e1)
        ctx.close()
    except Exception as e:          # secondary measurement: report, do not hide
        out["hmm_vit_kernel"] = {"error": str(e)}
    # (3) ms (multi-stream / continuous) scorer, BASELINE configs[3] flavour: en-us forced through the
    #     ms path (42 codebooks x 3 streams x 128 densities, float mixture weights re-quantised by the
    #     reference), 64 utterances x 50 frames, compallsen
    try:
        z = np.load(os.path.join(ROOT, "tests", "golden", "ms_en_us_tables.npz"))
        mt = {k: z[k] for k in z.files}
        ms = P.MsMgau(mt)
        n_fr = 64 * 50
        f = torch.from_numpy(np.ascontiguousarray(feats_h[:n_fr])).to(dev)
        nl = n_fr * ms.n_mgau * ms.n_feat * ms.topn
        ids = torch.empty(nl, dtype=torch.int32, device=dev)
        dist = torch.empty(nl, dtype=torch.float32, device=dev)
        scr = torch.empty((n_fr, ms.n_sen), dtype=torch.int16, device=dev)

        def mstep():
            capi.check(L.psgpu_ms_score_batch_dev(ms.h, C.c_void_p(f.data_ptr()), n_fr, C.c_void_p(ids.data_ptr()),
                                                  C.c_void_p(dist.data_ptr()), C.c_void_p(scr.data_ptr()), sp), "ms")
        mstep()
        capi.check(L.psgpu_ms_batch_check(ms.h, sp), "ms check")
        e0, e1 = C.c_void_p(), C.c_void_p()
        L.psgpu_event_create(C.byref(e0)); L.psgpu_event_create(C.byref(e1))
        K = 5
        L.psgpu_event_record(e0, sp)
        for _ in range(K):
            mstep()
        L.psgpu_event_record(e1, sp)
        ms_ = C.c_float()
        L.psgpu_event_elapsed_ms(e0, e1, C.byref(ms_))
        out["ms_scorer"] = {"frames_per_s": round(n_fr * K / (ms_.value * 1e-3), 1), "frames": n_fr,
                            "model": "en-us via ms (42 cb x 3 x 128, topn %d)" % ms.topn,
                            "ms_per_launch_pair": round(ms_.value / K, 4)}
        L.psgpu_event_destroy(e0); L.psgpu_event_destroy(e1)
        ms.close()
    except Exception as e:
        out["ms_scorer"] = {"error": str(e)}
    # (4) a fully continuous model of en-us size (BASELINE configs[3]: the only bundled continuous model,
    #     an4_ci_cont, has 102 one-density codebooks): 5126 senones x 16 densities x 39 dims, top-4,
    #     random parameters with the value ranges of real precomputed tables, 64 utterances x 250 frames
    try:
        rng = np.random.default_rng(9)
        n_sen, n_den, LL = 5126, 16, 39
        mt = dict(n_mgau=np.array([n_sen]), n_feat=np.array([1]), n_density=np.array([n_den]),
                  n_sen=np.array([n_sen]), max_topn=np.array([4]), aw=np.array([1]),
                  featlen=np.array([LL], np.int32),
                  mean=rng.standard_normal(n_sen * n_den * LL).astype(np.float32),
                  var=np.floor(np.exp(rng.uniform(0, 12, n_sen * n_den * LL))).astype(np.float32),
                  det=np.floor(rng.uniform(-500000, 400000, (n_sen, 1, n_den))).astype(np.float32),
                  pdf=rng.integers(0, 256, (n_sen, 1, n_den)).astype(np.uint8),
                  sen2mgau=np.arange(n_sen, dtype=np.uint32), logadd=t["logadd8"],
                  logadd_size=np.array([int(t["logadd8"].size)]), logadd_width=np.array([1]),
                  log_zero=np.array([-524288]))
        ms = P.MsMgau(mt)
        n_fr = 64 * 250                                      # BASELINE configs[3]: a batch of 64 utterances
        f = torch.from_numpy(rng.standard_normal((n_fr, LL)).astype(np.float32)).to(dev)
        nl = n_fr * ms.n_mgau * ms.n_feat * ms.topn
        ids = torch.empty(nl, dtype=torch.int32, device=dev)
        dist = torch.empty(nl, dtype=torch.float32, device=dev)
        scr = torch.empty((n_fr, ms.n_sen), dtype=torch.int16, device=dev)

        def cstep():
            capi.check(L.psgpu_ms_score_batch_dev(ms.h, C.c_void_p(f.data_ptr()), n_fr, None, None,
                                                  C.c_void_p(scr.data_ptr()), sp), "ms")
        cstep()
        capi.check(L.psgpu_ms_batch_check(ms.h, sp), "ms check")
        e0, e1 = C.c_void_p(), C.c_void_p()
        L.psgpu_event_create(C.byref(e0)); L.psgpu_event_create(C.byref(e1))
        K = 5
        L.psgpu_event_record(e0, sp)
        for _ in range(K):
            cstep()
        L.psgpu_event_record(e1, sp)
        ms_ = C.c_float()
        L.psgpu_event_elapsed_ms(e0, e1, C.byref(ms_))
        flop = n_sen * n_den * LL * 4
        out["ms_continuous"] = {"frames_per_s": round(n_fr * K / (ms_.value * 1e-3), 1), "frames": n_fr,
                                "model": "synthetic .cont. 5126 senones x 16 densities x 39 dims, topn 4",
                                "ms_per_launch_pair": round(ms_.value / K, 4),
                                "distance_tflops": round(flop * n_fr * K / (ms_.value * 1e-3) / 1e12, 2)}
        L.psgpu_event_destroy(e0); L.psgpu_event_destroy(e1)
        ms.close()
    except Exception as e:
        out["ms_continuous"] = {"error": str(e)}
    # (5) the whole device-side chain from audio: synthetic 16 kHz PCM (white noise
    #     plus a tiled tone burst, int16) -> MFCC front end -> 1s_c_d_dd features with batch CMN -> PTM
    #     senone scores, same batch shape as the headline (40 utterances x 250 frames)
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "mfcc_en_us_goforward.npz"))
        fe = P.FrontEnd({k: g[k] for k in g.files})
        par = [int(v) for v in g["par"]]
        fsz, fsh = par[0], par[1]
        n_samp = fsz + fsh * (UTT_LEN - 2)                  # UTT_LEN frames including the tail frame
        rng = np.random.default_rng(11)
        tt = np.arange(n_samp)
        pcm_h = np.concatenate([(2000 * rng.standard_normal(n_samp) + 6000 * np.sin(2 * np.pi * (200 + 37 * u) * tt / 16000.0)
                                 * (np.sin(2 * np.pi * 3 * tt / 16000.0) > 0)).astype(np.int16) for u in range(N_UTT)])
        soff = (np.arange(N_UTT + 1, dtype=np.int64) * n_samp)
        assert fe.n_frames(n_samp) == UTT_LEN
        Tn = N_UTT * UTT_LEN
        pcm = torch.from_numpy(pcm_h).to(dev)
        cep = torch.empty((Tn, fe.out_dim), dtype=torch.float32, device=dev)
        ft = torch.empty((Tn, 3 * fe.out_dim), dtype=torch.float32, device=dev)
        foff = torch.empty(N_UTT + 1, dtype=torch.int32, device=dev)
        tsc = torch.empty((Tn, model.n_chain, model.topn), dtype=torch.int32, device=dev)
        tcw = torch.empty((Tn, model.n_chain, model.topn), dtype=torch.uint8, device=dev)
        scr = torch.empty((Tn, model.n_sen), dtype=torch.int16, device=dev)
        L.psgpu_fe_process_utts_dev.argtypes = [C.c_void_p] * 10
        sarr = soff.ctypes.data_as(C.c_void_p)

        def fe_step():
            capi.check(L.psgpu_fe_process_utts_dev(fe.h, C.c_void_p(pcm.data_ptr()), sarr, N_UTT, None, None,
                                                   C.c_void_p(cep.data_ptr()), C.c_void_p(foff.data_ptr()), None, sp), "fe")

        def chain_step():
            fe_step()
            capi.check(L.psgpu_feat_1s_c_d_dd_dev(C.c_void_p(cep.data_ptr()), C.c_void_p(foff.data_ptr()), N_UTT,
                                                  fe.out_dim, C.c_void_p(ft.data_ptr()), sp), "feat")
            capi.check(L.psgpu_ptm_score_batch_dev(model.h, C.c_void_p(ft.data_ptr()), C.c_void_p(foff.data_ptr()), N_UTT, Tn,
                                                   None, None, C.c_void_p(tsc.data_ptr()), C.c_void_p(tcw.data_ptr()),
                                                   C.c_void_p(scr.data_ptr()), None, 0, sp), "score")
        e0, e1 = C.c_void_p(), C.c_void_p()
        L.psgpu_event_create(C.byref(e0)); L.psgpu_event_create(C.byref(e1))
        res = {}
        for name, fn in (("front_end", fe_step), ("pcm_to_scores", chain_step)):
            fn(); fn()
            K = 20
            L.psgpu_event_record(e0, sp)
            for _ in range(K):
                fn()
            L.psgpu_event_record(e1, sp)
            ms_ = C.c_float()
            L.psgpu_event_elapsed_ms(e0, e1, C.byref(ms_))
            res[name] = ms_.value / K
        out["pcm_pipeline"] = {"frames": Tn, "audio_s": round(N_UTT * n_samp / 16000.0, 2),
                               "front_end_ms": round(res["front_end"], 4),
                               "front_end_frames_per_s": round(Tn / (res["front_end"] * 1e-3), 1),
                               "pcm_to_scores_ms": round(res["pcm_to_scores"], 4),
                               "pcm_to_scores_frames_per_s": round(Tn / (res["pcm_to_scores"] * 1e-3), 1),
                               "xrt": round(res["pcm_to_scores"] * 1e-3 / (N_UTT * n_samp / 16000.0), 9),
                               "data": "synthetic 16 kHz int16 PCM (noise + gated tone), en-us front-end tables"}
        L.psgpu_event_destroy(e0); L.psgpu_event_destroy(e1)
        fe.close()
    except Exception as e:
        out["pcm_pipeline"] = {"error": str(e)}
    # (6) semi-continuous scorer, batched entry: tidigits model (4 streams x 256 densities, 4-bit clustered
    #     weights), 512 utterances x 100 frames (one wave per (utterance, stream) walks its frames in order)
    try:
        z = np.load(os.path.join(ROOT, "tests", "golden", "semi_tidigits_tables.npz"))
        g = np.load(os.path.join(ROOT, "tests", "golden", "senlog_tidigits_default.npz"))
        sm = P.SemiMgau({k: z[k] for k in z.files})
        n_u, u_len = 512, 100
        rng = np.random.default_rng(4)
        fh = np.ascontiguousarray(g["call_feat"][rng.integers(0, g["call_feat"].shape[0], n_u * u_len)], np.float32)
        f = torch.from_numpy(fh).to(dev)
        so = torch.arange(0, n_u * u_len + 1, u_len, dtype=torch.int32, device=dev)
        scr = torch.empty((n_u * u_len, sm.n_sen), dtype=torch.int16, device=dev)

        def sstep():
            capi.check(L.psgpu_semi_score_batch_dev(sm.m, C.c_void_p(f.data_ptr()), C.c_void_p(so.data_ptr()), n_u,
                                                    n_u * u_len, C.c_void_p(scr.data_ptr()), sp), "semi")
        sstep(); sstep()
        e0, e1 = C.c_void_p(), C.c_void_p()
        L.psgpu_event_create(C.byref(e0)); L.psgpu_event_create(C.byref(e1))
        K = 5
        L.psgpu_event_record(e0, sp)
        for _ in range(K):
            sstep()
        L.psgpu_event_record(e1, sp)
        ms_ = C.c_float()
        L.psgpu_event_elapsed_ms(e0, e1, C.byref(ms_))
        out["semi_scorer"] = {"frames_per_s": round(n_u * u_len * K / (ms_.value * 1e-3), 1), "frames": n_u * u_len,
                              "utterances": n_u, "model": "tidigits s2_semi (4 x 256, 4-bit weights, %d senones)" % sm.n_sen,
                              "ms_per_launch_pair": round(ms_.value / K, 4)}
        L.psgpu_event_destroy(e0); L.psgpu_event_destroy(e1)
        sm.close()
    except Exception as e:
        out["semi_scorer"] = {"error": str(e)}
    return out
