#!/usr/bin/env python3
"""A batch of live decoders (psgpu_decode_streams_*): LS_STREAMS utterances of LS_SEC seconds in progress at once, every stream fed
LS_CHUNK frames a step (10 = 100 ms of audio), one launch set a step; the feature vectors come from the host as a live caller's do
(computed once, on the device, from the benchmark's synthetic PCM).  Prints frames/s over all streams, the time of a step (= the
latency from a piece's arrival to every stream's updated hypothesis on the host), and whether the final hypotheses equal those of
ONE call over the same utterances (psgpu_decode_first_pass_feat).   LS_STREAMS=512 LS_SEC=30 LS_CHUNK=10 LS_FETCH=1"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _npz(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name))
    return {k: z[k] for k in z.files}


def main():
    n = int(os.environ.get("LS_STREAMS", "512")); sec = float(os.environ.get("LS_SEC", "30")); chunk = int(os.environ.get("LS_CHUNK", "10"))
    fetch = bool(int(os.environ.get("LS_FETCH", "1")))
    from pocketsphinx_amd import synth
    pcm_h = np.concatenate([synth.utterance(i % 64, sec) for i in range(n)])
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import capi
    dev = torch.device("cuda", 0)
    L = capi.lib()
    gt = _npz("fwdtree_trace_goforward.npz")
    p = P.DecodePipeline(_npz("mfcc_en_us_goforward.npz"), _npz("en_us_ptm_tables.npz"), _npz("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)
    pcm = torch.from_numpy(pcm_h).to(dev)
    soff = np.arange(n + 1, dtype=np.int64) * (pcm_h.size // n)
    # one call over whole utterances: the hypotheses to reproduce, and the feature vectors a live caller would hand over
    p.run_dev(pcm, soff)
    hn0, hyp0, res0 = p.fetch()
    v = p.view()
    T = int(v.total_frames) // n
    vl = 39
    feat = np.empty((n * T, vl), np.float32)
    capi.check(L.psgpu_memcpy_d2h(feat.ctypes.data_as(C.c_void_p), C.c_void_p(v.feat_dev), feat.nbytes, p._stream), "d2h")
    capi.check(L.psgpu_stream_sync(p._stream), "sync")
    feat = feat.reshape(n, T, vl)
    t_once = []
    for _ in range(2):
        t0 = time.perf_counter(); p.run_feat(feat.reshape(n * T, vl), [T] * n); p.fetch(want_hyp=False); t_once.append(time.perf_counter() - t0)
    # the same utterances as streams
    steps = (T + chunk - 1) // chunk
    pieces = [np.ascontiguousarray(feat[:, k * chunk:min((k + 1) * chunk, T)].reshape(-1, vl)) for k in range(steps)]
    cnts = [np.full(n, min((k + 1) * chunk, T) - k * chunk, np.int32) for k in range(steps)]
    n_obj = int(os.environ.get("LS_OBJECTS", "1"))
    if os.environ.get("LS_PCM"):
        # the streams fed with AUDIO (psgpu_decode_streams_step_pcm): chunk x 160 samples a stream a step -- one ps_process_raw call per live
        # decoder -- front end, live cepstral mean and feature window per stream on the device (no hypothesis to compare with: a live
        # decoder's features are not a whole-utterance decoder's; parity is tests/test_streams_pcm_gpu.py's)
        ns = pcm_h.size // n
        per = chunk * 160
        steps = (ns + per - 1) // per
        pcs = [np.ascontiguousarray(pcm_h.reshape(n, ns)[:, k * per:min((k + 1) * per, ns)].reshape(-1)) for k in range(steps)]
        cnt64 = [np.full(n, min((k + 1) * per, ns) - k * per, np.int64) for k in range(steps)]
        p.streams_pcm_begin(n, T + 8, chunk + 8, grow_feat=False)
        lat = []
        fin0 = np.zeros(n, np.uint8); fin1 = np.ones(n, np.uint8)
        gained = np.zeros(n, np.int32)
        t0 = time.perf_counter()
        for k in range(steps):
            ta = time.perf_counter()
            capi.check(L.psgpu_decode_streams_step_pcm(p.h, pcs[k].ctypes.data_as(C.c_void_p), cnt64[k].ctypes.data_as(C.c_void_p),
                                                       (fin1 if k == steps - 1 else fin0).ctypes.data_as(C.c_void_p), gained.ctypes.data_as(C.c_void_p),
                                                       p._stream), "step_pcm")
            if fetch or k == steps - 1:
                hn, hyp, res = p.fetch()
            lat.append(time.perf_counter() - ta)
        dt = time.perf_counter() - t0
        frames = int(res[:, 2].sum())
        first_ms, last_ms = 1e3 * lat[0], 1e3 * lat[-1]
        lat = np.sort(np.array(lat[1:]))
        print(json.dumps({"metric": "frames/s over all streams, live FROM AUDIO: %d streams fed %d samples a step" % (n, per), "value": round(frames / dt, 1),
                          "unit": "frames/s", "streams": n, "seconds_per_utterance": sec, "samples_per_step_per_stream": per, "steps": steps,
                          "ms_per_step": round(1e3 * dt / steps, 4), "step_ms_median": round(1e3 * float(np.median(lat)), 4),
                          "step_ms_p99": round(1e3 * float(lat[int(0.99 * (len(lat) - 1))]), 4), "first_step_ms": round(first_ms, 3),
                          "last_step_ms": round(last_ms, 3), "audio_ms_per_step": 10.0 * chunk, "xrt": round(dt / (n * sec), 8),
                          "frames_searched": frames, "status_nonzero": int((res[:, 3] != 0).sum()), "words_per_hyp": round(float(hn[:, 0].mean()), 1),
                          "what": "psgpu_decode_streams_step_pcm per piece: the reference's buffer counters walked on the host, H2D of the audio, front end + "
                                  "cmn_live + feature window per stream, batch scorer, phone loop, all streams' searches resumed, hypotheses to the host"}))
        p.close()
        return
    if n_obj == 2:
        # the streams split over TWO pipeline objects on streams of their own: one half's kernels run while the host fetches the other
        # half's hypotheses and hands over its next pieces
        from pocketsphinx_amd import decode as pdec
        h = n // 2
        q = P.DecodePipeline(_npz("mfcc_en_us_goforward.npz"), _npz("en_us_ptm_tables.npz"), _npz("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)
        objs = [(p, 0, h, pdec.dedicated_stream()), (q, h, n, pdec.dedicated_stream())]
        for o, a, b, st in objs:
            o.streams_begin(b - a, T + 8, chunk, stream=st)
        pcs = [[np.ascontiguousarray(feat[a:b, k * chunk:min((k + 1) * chunk, T)].reshape(-1, vl)) for k in range(steps)] for _, a, b, _ in objs]
        fin0 = np.zeros(n, np.uint8); fin1 = np.ones(n, np.uint8)
        lat = []; outs = [None, None]
        t0 = time.perf_counter()
        for k in range(steps):
            ta = time.perf_counter()
            for i, (o, a, b, st) in enumerate(objs):
                capi.check(L.psgpu_decode_streams_step(o.h, pcs[i][k].ctypes.data_as(C.c_void_p), cnts[k][a:b].ctypes.data_as(C.c_void_p),
                                                       (fin1 if k == steps - 1 else fin0)[a:b].ctypes.data_as(C.c_void_p), o._stream), "step")
            for i, (o, a, b, st) in enumerate(objs):
                outs[i] = o.fetch()
            lat.append(time.perf_counter() - ta)
        dt = time.perf_counter() - t0
        hn = np.concatenate([outs[0][0], outs[1][0]]); hyp = np.concatenate([outs[0][1], outs[1][1]]); res = np.concatenate([outs[0][2], outs[1][2]])
        searched = p.live_frames_searched() + q.live_frames_searched()
        q.close()
    else:
        p.streams_begin(n, T + 8, chunk)
        lat = []
        fin0 = np.zeros(n, np.uint8); fin1 = np.ones(n, np.uint8)
        t0 = time.perf_counter()
        for k in range(steps):
            ta = time.perf_counter()
            capi.check(L.psgpu_decode_streams_step(p.h, pieces[k].ctypes.data_as(C.c_void_p), cnts[k].ctypes.data_as(C.c_void_p),
                                                   (fin1 if k == steps - 1 else fin0).ctypes.data_as(C.c_void_p), p._stream), "step")
            if fetch or k == steps - 1:
                hn, hyp, res = p.fetch()
            lat.append(time.perf_counter() - ta)
        dt = time.perf_counter() - t0
        searched = p.live_frames_searched()
    same = bool(np.array_equal(hn, hn0) and np.array_equal(res[:, :5], res0[:, :5])
                and all(np.array_equal(hyp[u, :hn[u, 0]], hyp0[u, :hn0[u, 0]]) for u in range(n)))
    first_ms, last_ms = 1e3 * lat[0], 1e3 * lat[-1]
    lat = np.sort(np.array(lat[1:]))
    out = {"metric": "frames/s over all streams, live: %d streams fed %d frames a step" % (n, chunk), "value": round(n * T / dt, 1), "unit": "frames/s",
           "streams": n, "seconds_per_utterance": sec, "frames_per_step_per_stream": chunk, "steps": steps,
           "ms_per_step": round(1e3 * dt / steps, 4), "step_ms_median": round(1e3 * float(np.median(lat)), 4),
           "step_ms_p99": round(1e3 * float(lat[int(0.99 * (len(lat) - 1))]), 4),
           "first_step_ms": round(first_ms, 3), "last_step_ms": round(last_ms, 3), "audio_ms_per_step": 10.0 * chunk, "xrt": round(dt / (n * sec), 8),
           "hypotheses_fetched_every_step": fetch, "pipeline_objects": n_obj, "frames_searched": searched, "frames": n * T,
           "one_call_over_the_same_features_s": round(min(t_once), 4), "one_call_frames_per_s": round(n * T / min(t_once), 1),
           "final_hypotheses_equal_the_one_call_decode": same, "status_nonzero": int((res[:, 3] != 0).sum()),
           "what": "psgpu_decode_streams_step per piece: H2D of the pieces' features, batch scorer, phone loop, window copy, all streams' "
                   "searches resumed, hypotheses of every stream to the host"}
    print(json.dumps(out))
    p.close()


if __name__ == "__main__":
    main()
