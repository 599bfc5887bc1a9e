#!/usr/bin/env python3
"""bench.py -- decode throughput of the device first pass on MI355X (BASELINE.json metric: frames/sec + xRT decode,
en-us PTM, n-gram fwdtree).

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.  N is the number of ranks, one per
GPU over RCCL: under a launcher (torch.distributed.run, the driver's form for N > 1) WORLD_SIZE must equal it; started plainly with
N > 1 the process becomes that launcher itself (self_launch_argv); a node with fewer GPUs than ranks fails loudly.

Workload (BASELINE.json configs[4], the per-GPU share): 512 utterances x 30 s of synthetic 16 kHz PCM (the bundled
recordings tiled with random gains and pauses over a noise floor, pocketsphinx_amd/synth.py; utterance id = seed),
en-us PTM acoustic model (5126 senones), turtle n-gram LM + dictionary as the reference built them
(tests/golden/fwdtree_static_en_us_turtle.npz), `-fwdflat no -bestpath no` ("fwdtree only").  A "step" is one
pass of the whole hot path over that batch, PCM already resident in HBM:
    PCM -> MFCC -> 1s_c_d_dd features -> PTM senone scores -> phone-loop search -> lexicon-tree Viterbi search
        -> back-pointer tables -> best exit + backtrace -> hypothesis records (word id, start, end, score) on the host
through ONE C-ABI call per step (psgpu_decode_first_pass_dev, include/psgpu.h) plus the fetch of the hypothesis records.
value = frames/s over all ranks; xrt = seconds of compute per second of audio.

N > 1 (weak scaling, utterances shard with no data-path collective inside the decode): rank 0 owns the PCM of the whole
job, scatters each rank's 512 utterances over RCCL before the timed region; every timed step ends with the gather of the
fixed-size hypothesis records to rank 0 (pocketsphinx_amd/batch.py).

Extra objects on the line:
  roofline     HBM roofline of the dominant kernel (fwdtree_kernel): algorithmic bytes per launch = sum over frames of
               156 + 2 * listed senones + 86 * HMM evaluations (SURVEY 8d "full decode, per frame"), counted by the kernel
               itself, over the kernel's duration from HIP events on its launch stream inside the library
  cpu_baseline the UNMODIFIED reference (oracle/_ref/ref_decode_bench, kind "reference") decoding a bounded sample of the
               SAME utterances on one host thread; the same run is the parity check: every sampled utterance's words and
               frame boundaries from the device must equal the reference's
"""
import argparse
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_UTT, UTT_SECONDS = 512, 30.0
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_GOPS = 78643.2         # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (non-FMA fp32 op rate)
N_SAMPLE = 64                    # utterances decoded by the reference for cpu_baseline + parity
LEG_CHECK = 32                   # ... on the other decode legs (two passes, multi-stream scorer, large vocabulary)
# the compiled reference (test infrastructure): what this file does with it is time it (cpu_baseline) and compare with it (parity)
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
# ... and the same reference built with upstream's Release flags (-O3 -DNDEBUG: `make -C oracle release`): what cpu_baseline times
REF_REL = os.path.join(REF_DIR, "release")


def ref_exe(name):
    """(path, flags) of a timing program of the compiled reference: the Release-flag build when it is there"""
    rel = os.path.join(REF_REL, name)
    if os.path.exists(rel):
        return rel, "gcc -O3 -DNDEBUG (upstream's Release flags)"
    return os.path.join(REF_DIR, name), "gcc -O2"
LV_UTT, LV_CHECK, LV_STEPS = 256, 64, 2    # the large-vocabulary leg: utterances per step, utterances the reference decodes, timed steps


def _npz(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name))
    return {k: z[k] for k in z.files}


def _synth_range(args):
    from pocketsphinx_amd import synth
    first, n, seconds = args
    return synth.batch(first, n, seconds)[0]


def synth_pcm(first_id, n_utt, seconds):
    """PCM of utterances first_id .. first_id + n_utt - 1, int16, back to back (a process pool above 64 utterances;
    called before torch / HIP are initialised)"""
    from pocketsphinx_amd import synth
    if n_utt <= 64:
        return synth.batch(first_id, n_utt, seconds)[0]
    from concurrent.futures import ProcessPoolExecutor
    nw = min(32, os.cpu_count() or 1)
    per = (n_utt + nw - 1) // nw
    jobs = [(first_id + i, min(per, n_utt - i), seconds) for i in range(0, n_utt, per)]
    with ProcessPoolExecutor(max_workers=nw) as ex:
        return np.concatenate(list(ex.map(_synth_range, jobs)))


def _ref_one(job):
    exe, model, lm, dic, path, n_samples, extra = job
    out = subprocess.run([exe, model, lm, dic, path, str(n_samples)] + (["--"] + list(extra) if extra else []), capture_output=True,
                         text=True, timeout=1800)
    lines = [json.loads(ln) for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines or "total" not in lines[-1]:
        raise RuntimeError("ref_decode_bench rc %d: %s" % (out.returncode, out.stderr[-300:]))
    return lines[:-1], lines[-1]


def reference_decode(pcm, n_samples, ids, lm="turtle.lm.bin", dic="turtle.dic", procs=1, model="en-us", extra=()):
    """the compiled reference on the same PCM: ([json per utterance, in the order of ids], totals); None when oracle/_ref is
    absent.  procs > 1: that many reference processes side by side, each one thread decoding its share of the utterances
    (pocketsphinx_batch's way to use a machine: one decoder per core, programs/pocketsphinx_batch.c) -- totals then carry
    the wall time of the slowest process as well"""
    ref = REF_DIR
    exe = ref_exe("ref_decode_bench")[0]
    if not os.path.exists(exe):
        return None
    procs = max(1, min(procs, len(ids)))
    shares = [ids[k::procs] for k in range(procs)]
    paths = []
    try:
        for sh in shares:
            with tempfile.NamedTemporaryFile(suffix=".raw", delete=False) as fh:
                for i in sh:
                    pcm[i * n_samples:(i + 1) * n_samples].tofile(fh)
                paths.append(fh.name)
        jobs = [(exe, os.path.join(ref, "model", model), os.path.join(ref, "data", lm), os.path.join(ref, "data", dic), pth, n_samples, extra)
                for pth in paths]
        t0 = time.perf_counter()
        if procs == 1:
            outs = [_ref_one(jobs[0])]
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=procs) as ex:
                outs = list(ex.map(_ref_one, jobs))
        wall = time.perf_counter() - t0
    finally:
        for pth in paths:
            os.unlink(pth)
    by_id = {}
    for sh, (utts, _) in zip(shares, outs):
        for i, r in zip(sh, utts):
            by_id[i] = r
    frames = sum(t["frames"] for _, t in outs); cpu_s = sum(t["cpu_s"] for _, t in outs)
    tot = {"frames": frames, "cpu_s": cpu_s, "frames_per_s": frames / cpu_s if cpu_s > 0 else 0.0,
           "xrt": cpu_s / (frames / 100.0) if frames else 0.0, "procs": procs, "wall_s": wall,
           "frames_per_s_all_procs": frames / wall if wall > 0 else 0.0}
    return [by_id[i] for i in ids], tot


def parity_of(ids, utts, hn, hyp, res):
    """utterances whose device hypothesis (word ids, start / end frames, path score, frame count) differs from the reference's"""
    bad = []
    for i, r in zip(ids, utts):
        got = [tuple(int(v) for v in hyp[i, k, :3]) for k in range(min(int(hn[i, 0]), hyp.shape[1]))]
        want = [(s[1], s[2], s[3]) for s in r["seg"]]
        if got != want or int(hn[i, 1]) != r["score"] or int(res[i, 2]) != r["frames"]:
            bad.append(i)
    return bad


def parity_of_the_rest(pcm, n_samples, n_utt, ids, hn, hyp, res, **kw):
    """parity beyond the timed sample: the reference decodes EVERY utterance of the step that `ids` left out, on half the host's cores
    (no timing: cpu_baseline stays the sample's) -- only where the host has the cores to do it in seconds (>= 32; PSGPU_BENCH_FULL_PARITY=0/1
    overrides).  -> (utterances checked, ids that differ, wall seconds), or None"""
    want = os.environ.get("PSGPU_BENCH_FULL_PARITY")
    if want == "0" or (want is None and (os.cpu_count() or 1) < 32):
        return None
    rest = [i for i in range(n_utt) if i not in set(ids)]
    if not rest:
        return 0, [], 0.0
    ref = reference_decode(pcm, n_samples, rest, procs=max(1, (os.cpu_count() or 2) // 2), **kw)
    if ref is None:
        return None
    return len(rest), parity_of(rest, ref[0], hn, hyp, res), ref[1]["wall_s"]


def widen_parity(par, rest):
    if rest is None:
        return par
    n, bad, wall = rest
    par["sample_checked"] = par["checked"]
    par["checked"] += n; par["identical"] += n - len(bad); par["mismatching_utterances"] = list(par["mismatching_utterances"]) + bad
    par["rest_wall_s"] = round(wall, 1)
    par["note"] = "every utterance of the step: the timed sample, then the others decoded by reference processes on half the host's cores"
    return par


def large_vocab_leg(P, pcm_all, n_samp, seconds, dev, fe_tables, ptm_tables, n_utt, steps, n_check, with_cpu, table_dir=None):
    """The large-vocabulary decode (the stand-in for configs[2]'s absent en-us.lm.bin): the first
    n_utt of the headline's utterances through the same device pipeline with the 134,865-word dictionary and the synthetic
    126k-unigram LM (trie on the device) -- PCM -> hypotheses -- timed; the reference decodes n_check of them with the same LM
    and dictionary: cpu_baseline + per-utterance parity."""
    import torch
    from pocketsphinx_amd import largevocab as lv
    tpath = lv.table_path(directory=table_dir)
    if not lv.available(tpath):
        return {"skipped": "table file %s not found (integration/psgpu_export_tables; `make -C integration tables`, or --tables DIR)" % tpath}
    t0 = time.perf_counter()
    g = lv.tables(tpath)
    t_tab = time.perf_counter() - t0
    pipe = lv.pipeline(g, fe_tables, ptm_tables)
    pipe.stage_timing(True)
    slab = C.c_int64(); lds = C.c_int32()
    from pocketsphinx_amd import capi
    capi.check(capi.lib().psgpu_fwdtree_layout(pipe.search.h, C.byref(lds), C.byref(slab)), "psgpu_fwdtree_layout")
    pcm = torch.from_numpy(pcm_all[:n_utt * n_samp]).to(dev)
    soff = np.arange(n_utt + 1, dtype=np.int64) * n_samp
    stream = torch.cuda.current_stream().cuda_stream
    pipe.run_dev(pcm, soff, stream); pipe.fetch(want_hyp=False)          # warm-up (allocations, table growth if any)
    torch.cuda.synchronize()
    stage = []
    t1 = time.perf_counter()
    for _ in range(steps):
        pipe.run_dev(pcm, soff, stream)
        hn, hyp, res = pipe.fetch()
        stage.append(pipe.last_stage_ms())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / steps
    n_bad_status = int((res[:, 3] != 0).sum())
    frames = int(res[:, 2].sum())
    evals = int(res[:, 5].astype(np.int64).sum() + (res[:, 6].astype(np.int64) << 32).sum())
    senones = int(res[:, 7].astype(np.int64).sum())
    alg_bytes = 156 * frames + 2 * senones + 86 * evals
    st_mean = {k: float(np.mean([s[k] for s in stage])) for k in stage[0]}
    search_s = st_mean["search"] * 1e-3
    par = g["par"]
    traffic = None
    for ppath in sorted(glob.glob(os.path.join(ROOT, "profiles", "*largevocab*_pmc_traffic.json")), reverse=True):
        try:       # the newest committed PMC pass of THIS leg at THIS batch size (tools/gpu_call_lvpmc.sh)
            jt = json.load(open(ppath))
            if jt.get("_workload", {}).get("leg") == "decode_large_vocab" and jt["_workload"].get("utterances") == n_utt \
                    and jt["_workload"].get("seconds") == seconds and jt.get("fwdtree_kernel", {}).get("hbm_bytes_per_launch"):
                traffic = round(jt["fwdtree_kernel"]["hbm_bytes_per_launch"])
                break
        except Exception:
            continue
    out = {
        "metric": "frames/sec + xRT decode, en-us PTM 5126-senone n-gram fwdtree, 134,865-word dictionary (device first pass, PCM -> hypotheses)",
        "value": round(frames / dt, 1), "unit": "frames/s", "ms_per_step": round(1e3 * dt, 2), "steps": steps,
        "xrt": round(dt / (n_utt * seconds), 7),
        "config": {"workload": "%d utterances x %g s (the headline's first %d), en-us PTM + big.arpa (126,055 unigrams) + "
                               "cmudict-en-us.dict (%d words, %d lexicon-tree channels), fwdtree only, trie LM on the device"
                               % (n_utt, seconds, n_utt, int(par[3]), int(par[4] + par[5])),
                   "utterances": n_utt, "frames_per_step": frames, "maxhmmpf": int(par[17]),
                   "search_slab_bytes_per_utterance": int(slab.value), "lds_layout": bool(lds.value)},
        "stage_ms": {k: round(v, 3) for k, v in st_mean.items()},
        "workload_counts": {"hmm_evals_per_frame": round(evals / max(frames, 1), 1), "listed_senones_per_frame": round(senones / max(frames, 1), 1),
                            "back_pointers_per_utt": round(float(res[:, 0].mean()), 1), "score_stack_per_utt": round(float(res[:, 1].mean()), 1),
                            "words_per_hyp": round(float(hn[:, 0].mean()), 1), "table_growths": pipe.tables_grown()},
        "roofline": {"bound": "hbm", "kernel": "fwdtree_kernel<3, 1024, false, false>", "achieved": round(alg_bytes / search_s / 1e9, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_bytes / search_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(st_mean["search"], 2),
                     "note": "one 1024-work-item workgroup per utterance, all state in a per-utterance slab in device memory; bytes = sum over "
                             "frames of 156 + 2 x listed senones + 86 x HMM evaluations (SURVEY 8d), counted by the kernel; kernel_ms from HIP "
                             "events on its launch stream; traffic = L2-miss bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE) of the newest "
                             "COMMITTED PMC pass of this leg at this batch size (profiles/*largevocab*_pmc_traffic.json, "
                             "tools/gpu_call_lvpmc.sh): a constant of the repository, not a measurement of this run; traffic / "
                             "kernel time = the rate the memory system actually sustains"},
        "table_file": os.path.relpath(tpath, ROOT), "table_file_read_s": round(t_tab, 2),
    }
    if n_bad_status:
        out["error"] = "%d utterances ended with status != 0" % n_bad_status
    if with_cpu:
        ids = sorted(set(int(i) for i in np.linspace(0, n_utt - 1, min(n_check, n_utt))))
        procs = max(1, min(len(ids), (os.cpu_count() or 2) // 2))
        utts, tot = reference_decode(pcm_all, n_samp, ids, "big.arpa", "cmudict-en-us.dict", procs)
        bad = parity_of(ids, utts, hn, hyp, res)
        out["cpu_baseline"] = {"value": round(tot["frames_per_s"], 2), "unit": "frames/s", "cores": 1, "kind": "reference", "xrt": round(tot["xrt"], 5),
                               "sample": "%d of the step's %d utterances (%d frames, %.1f s of CPU in all; %d one-thread reference processes side by "
                                         "side, value = frames / summed CPU seconds), same LM and dictionary, -fwdflat no -bestpath no"
                                         % (len(ids), n_utt, tot["frames"], tot["cpu_s"], tot["procs"]),
                               "what": "unmodified reference (%s)" % ref_exe("ref_decode_bench")[1]}
        out["parity"] = {"checked": len(ids), "identical": len(ids) - len(bad), "mismatching_utterances": bad,
                         "what": "word ids, start / end frames, path score and frame count: device vs the reference on the same PCM"}
        widen_parity(out["parity"], parity_of_the_rest(pcm_all, n_samp, n_utt, ids, hn, hyp, res, lm="big.arpa", dic="cmudict-en-us.dict"))
        out["speedup_vs_cpu_1thread"] = round(frames / dt / tot["frames_per_s"], 1)
        w = lv.words_of(g)
        out["sample_hyp"] = " ".join(w[int(hyp[0, k, 0])] for k in range(min(int(hn[0, 0]), 12)))
    pipe.close()
    del pcm
    torch.cuda.empty_cache()
    return out


def ms_scorer_leg(P, pcm_all, n_samp, seconds, dev, fe_tables, static, gt, n_utt, steps, n_check, with_cpu, cont_tables=None):
    """BASELINE configs[3]'s per-GPU material: a batch of 64 utterances decoded with the multi-stream scorer (en-us through
    ms_cont_mgau_frame_eval, the reference's -senmgau route: 42 codebooks x 3 streams x 128 densities, top-4 by full scan,
    16-bit log-add) -- PCM -> hypotheses through the same device pipeline; the reference decodes n_check of them with that scorer.
    cont_tables: the leg with a CONTINUOUS-density model instead (configs[3] as written): the scorer's tables of the staged
    5126 x 16 x 39 model as integration/psgpu_export_tables wrote them (the same lexicon tree: en-us's mdef), the reference
    decoding with that model directory"""
    import torch
    ms = P.MsMgau(cont_tables if cont_tables is not None else _npz("ms_en_us_tables.npz"))
    pipe = P.DecodePipeline(fe_tables, None, static, gt["par"], gt, scorer=ms)
    pipe.stage_timing(True)
    pcm = torch.from_numpy(pcm_all[:n_utt * n_samp]).to(dev)
    soff = np.arange(n_utt + 1, dtype=np.int64) * n_samp
    stream = torch.cuda.current_stream().cuda_stream
    pipe.run_dev(pcm, soff, stream); pipe.fetch(want_hyp=False)
    torch.cuda.synchronize()
    stage = []
    t1 = time.perf_counter()
    for _ in range(steps):
        pipe.run_dev(pcm, soff, stream)
        hn, hyp, res = pipe.fetch()
        stage.append(pipe.last_stage_ms())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / steps
    frames = int(res[:, 2].sum())
    st_mean = {k: float(np.mean([s_[k] for s_ in stage])) for k in stage[0]}
    out = {"frames_per_s": round(frames / dt, 1), "ms_per_step": round(1e3 * dt, 2), "utterances": n_utt, "frames": frames,
           "xrt": round(dt / (n_utt * seconds), 8), "stage_ms": {k: round(v, 3) for k, v in st_mean.items()},
           "status_nonzero": int((res[:, 3] != 0).sum()),
           "what": ("configs[3]: %d utterances x %g s, a continuous-density model of en-us size (5126 senones x 16 densities x 39 dimensions, "
                    "one codebook a senone: ms_gauden + ms_senone with .cont. mixtures; synthetic parameters cut from en-us, "
                    "oracle/stage_cont_model.py) + turtle LM, fwdtree only, PCM -> hypotheses on one MI355X" % (n_utt, seconds))
                   if cont_tables is not None else
                   "configs[3]: %d utterances x %g s, en-us through the ms scorer (-senmgau .ptm.) + turtle LM, fwdtree only, PCM -> "
                   "hypotheses on one MI355X (the per-GPU share of the 8-way shard; utterances shard as in the headline)" % (n_utt, seconds)}
    if with_cpu:
        ids = sorted(set(int(i) for i in np.linspace(0, n_utt - 1, min(n_check, n_utt))))
        ref = reference_decode(pcm_all, n_samp, ids, procs=min(len(ids), max(1, (os.cpu_count() or 2) // 2)),
                               model="en-us-cont" if cont_tables is not None else "en-us-ms",
                               extra=() if cont_tables is not None else ("senmgau", ".ptm."))
        if ref is not None:
            utts, tot = ref
            bad = parity_of(ids, utts, hn, hyp, res)
            out["cpu_baseline"] = {"value": round(tot["frames_per_s"], 2), "unit": "frames/s", "cores": 1, "kind": "reference",
                                   "sample": "%d utterances, %.1f s of CPU in all, %s-fwdflat no -bestpath no" % (
                                       len(ids), tot["cpu_s"], "" if cont_tables is not None else "-senmgau .ptm. ")}
            out["parity"] = {"checked": len(ids), "identical": len(ids) - len(bad), "mismatching_utterances": bad}
    pipe.close(); ms = None
    del pcm
    torch.cuda.empty_cache()
    return out


def two_pass_leg(P, pcm_all, n_samp, seconds, dev, fe_tables, ptm_tables, static, gt, n_utt, steps, n_check, with_cpu):
    """the reference's DEFAULT search configuration (-fwdflat yes; -bestpath no here) at the headline's scale: both search passes on
    the device inside one pipeline object (psgpu_decode_second_pass), PCM -> the second pass's hypotheses; the reference decodes
    n_check of the utterances with both passes"""
    import torch
    gf, fst = _npz("fwdflat_trace_goforward.npz"), _npz("fwdflat_static_en_us_turtle.npz")
    pipe = P.DecodePipeline(fe_tables, ptm_tables, static, gt["par"], gt)
    flat = P.FwdflatSearch(static, fst, gf["par"], gf["flat_par"], gf["flat_lwf"])
    pcm = torch.from_numpy(pcm_all[:n_utt * n_samp]).to(dev)
    soff = np.arange(n_utt + 1, dtype=np.int64) * n_samp
    stream = torch.cuda.current_stream().cuda_stream
    pipe.run_dev(pcm, soff, stream); pipe.second_pass(flat); pipe.fetch(want_hyp=False)
    torch.cuda.synchronize()
    t_first = t_second = 0.0
    t1 = time.perf_counter()
    for _ in range(steps):
        ta = time.perf_counter()
        pipe.run_dev(pcm, soff, stream)
        torch.cuda.synchronize(); tb = time.perf_counter()
        pipe.second_pass(flat)
        hn, hyp, res = pipe.fetch()
        tc = time.perf_counter()
        t_first += tb - ta; t_second += tc - tb
    dt = (time.perf_counter() - t1) / steps
    frames = int(res[:, 2].sum())
    out = {"frames_per_s": round(frames / dt, 1), "ms_per_step": round(1e3 * dt, 2), "utterances": n_utt, "frames": frames,
           "xrt": round(dt / (n_utt * seconds), 8), "first_pass_ms": round(1e3 * t_first / steps, 2),
           "second_pass_ms": round(1e3 * t_second / steps, 2), "status_nonzero": int((res[:, 3] != 0).sum()),
           "what": "%d utterances x %g s, en-us PTM + turtle LM, fwdtree AND fwdflat on the device in one pipeline object (one batch at a "
                   "time: no overlap of batches as in the headline), PCM -> the second pass's hypotheses; second_pass_ms includes the host's "
                   "vocabulary build from the first pass's tables and the fetch" % (n_utt, seconds)}
    if with_cpu:
        ids = sorted(set(int(i) for i in np.linspace(0, n_utt - 1, min(n_check, n_utt))))
        ref = reference_decode(pcm_all, n_samp, ids, procs=min(len(ids), max(1, (os.cpu_count() or 2) // 2)), extra=("fwdflat", "yes", "bestpath", "no"))
        if ref is not None:
            utts, tot = ref
            bad = parity_of(ids, utts, hn, hyp, res)
            out["cpu_baseline"] = {"value": round(tot["frames_per_s"], 2), "unit": "frames/s", "cores": 1, "kind": "reference",
                                   "sample": "%d utterances, %.1f s of CPU in all, -fwdflat yes -bestpath no" % (len(ids), tot["cpu_s"])}
            out["parity"] = {"checked": len(ids), "identical": len(ids) - len(bad), "mismatching_utterances": bad}
            widen_parity(out["parity"], parity_of_the_rest(pcm_all, n_samp, n_utt, ids, hn, hyp, res, extra=("fwdflat", "yes", "bestpath", "no")))
    flat.close(); pipe.close()
    del pcm
    torch.cuda.empty_cache()
    return out


class _DevArray:
    """a device buffer of the library as a zero-copy torch tensor (torch.as_tensor on __cuda_array_interface__)"""

    def __init__(self, ptr, shape, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def scorer_extra(P, capi, L, dev, sp, tables):
    """the senone scorer alone (BASELINE configs[1]: 10,000 synthetic frames, compallsen): round 1's headline"""
    import torch
    n_utt, utt_len = 40, 250
    T = n_utt * utt_len
    mean = tables["mean"].reshape(int(tables["n_mgau"][0]), int(tables["n_feat"][0]), int(tables["n_density"][0]), -1)
    mu = mean.mean(axis=(0, 2)).reshape(-1); sd = mean.std(axis=(0, 2)).reshape(-1)
    feats_h = (mu + sd * np.random.default_rng(20260921).standard_normal((T, mu.size))).astype(np.float32)
    model = P.PtmModel(tables)
    feats = torch.from_numpy(feats_h).to(dev)
    off = torch.arange(0, T + 1, utt_len, dtype=torch.int32, device=dev)
    tsc = torch.empty((T, model.n_chain, model.topn), dtype=torch.int32, device=dev)
    tcw = torch.empty((T, model.n_chain, model.topn), dtype=torch.uint8, device=dev)
    scr = torch.empty((T, model.n_sen), dtype=torch.int16, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731

    def step():
        capi.check(L.psgpu_ptm_score_batch_dev(model.h, p(feats), p(off), n_utt, T, None, None, p(tsc), p(tcw), p(scr), None, 0, sp),
                   "score_batch_dev")
    capi.check(L.psgpu_ptm_kernel_timing(model.h, 1), "kernel_timing")
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    K = 20
    ms3 = (C.c_float * 3)()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    capi.check(L.psgpu_ptm_last_kernel_ms(model.h, ms3), "last_kernel_ms")
    lane, fix, sen = float(ms3[0]), float(ms3[1]), float(ms3[2])
    bpf, fpf = 39 * 4 + model.n_sen * 2, 16128 * 13 * 4
    out = {"frames_per_s": round(T * K / dt, 1), "frames": T, "ms_per_step": round(1e3 * dt / K, 4),
           "kernels_ms": {"ptm_lane_kernel": round(lane, 4), "ptm_chain_kernel(fix-up)": round(fix, 4), "ptm_senone_kernel_f3n4": round(sen, 4)},
           "roofline": {"bound": "hbm", "kernel": "ptm_lane_kernel", "achieved": round(bpf * T / (lane * 1e-3) / 1e9, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bpf * T / (lane * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "note": "VALU-bound by construction; fp32-VALU fraction of the distance work = %.4f"
                                % (fpf * T / (lane * 1e-3) / 1e9 / VALU_PEAK_GOPS)},
           "what": "configs[1]: en-us PTM senone-score-only, 10,000 synthetic frames = 40 utterances x 250, compallsen, topn 4"}
    # cpu_baseline + parity: the unmodified reference's ptm_mgau_frame_eval(compallsen) on the same feature vectors, one thread
    # (oracle/_ref/ref_score_bench: the loop around the scorer's vtable entry, acmod.c:1076-1133), every one of its int16 scores
    # against the device's
    exe, exe_flags = ref_exe("ref_score_bench")
    if os.path.exists(exe):
        with tempfile.TemporaryDirectory() as td:
            fpath, spath = os.path.join(td, "feats.f32"), os.path.join(td, "scores.i16")
            feats_h.tofile(fpath)
            r = subprocess.run([exe, os.path.join(REF_DIR, "model", "en-us"), fpath, str(utt_len), spath], capture_output=True, text=True, timeout=600)
            lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            if r.returncode == 0 and lines and os.path.exists(spath):
                j = json.loads(lines[-1])
                want = np.fromfile(spath, np.int16).reshape(T, model.n_sen)
                got = scr.cpu().numpy()
                bad = int((want != got).any(axis=1).sum())
                out["cpu_baseline"] = {"value": round(j["frames_per_s"], 1), "unit": "frames/s", "cores": 1, "kind": "reference",
                                       "sample": "all %d frames of the step (%.1f s of CPU)" % (j["frames"], j["seconds"]),
                                       "what": "unmodified reference (" + exe_flags + "): ptm_mgau_frame_eval, compallsen, "
                                               "fresh top-N history per utterance"}
                out["parity"] = {"frames_checked": T, "senones": int(model.n_sen), "frames_with_a_differing_score": bad,
                                 "what": "every int16 senone score of every frame: device vs the reference (bit-exact = 0 differing)"}
                out["speedup_vs_cpu_1thread"] = round(out["frames_per_s"] / j["frames_per_s"], 1)
            else:
                out["cpu_baseline"] = {"error": (r.stderr or r.stdout)[-300:]}
    return out, model, feats_h


def child_extras(out):
    """measurements that run in CHILD processes with a time limit (a fault or a slow case cannot take the headline with it)"""
    t_children = time.perf_counter()

    def child(key, argv, env, limit):
        left = 360.0 - (time.perf_counter() - t_children)        # all children together: six minutes at most
        if left < 20.0:
            out[key] = {"skipped": "time budget of the child-process extras used up"}
            return
        try:
            r = subprocess.run([sys.executable] + argv, env=dict(os.environ, **env), capture_output=True, text=True,
                               timeout=min(limit, left))
            lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
            if r.returncode != 0 or not lines:
                out[key] = {"error": "exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}
            elif lines[-1].lstrip().startswith("{"):
                out[key] = json.loads(lines[-1])
            else:
                out[key] = {"lines": lines[-6:]}
        except Exception as e:
            out[key] = {"error": str(e)[-400:]}
    sb = os.path.join(ROOT, "tools", "search_bench.py")
    child("search_only_turtle", [sb], {"SB_CASE": "goforward", "SB_BATCHES": "512,1024", "SB_REPS": "2"}, 100)
    child("search_only_medium", [sb], {"SB_CASE": "medium_goforward", "SB_BATCHES": "512", "SB_REPS": "2"}, 100)
    if os.path.exists(os.path.join(REF_DIR, "ref_dump")):
        # the full cmudict task (134,865 words) on replicas of a recorded reference trace (its scores are the input, its tables the
        # check): single-thread reference ~1.2 k frames/s on this decode (profiles/r01i_*)
        child("search_only_cmudict", [sb], {"SB_CASE": "cmudict", "SB_BATCHES": "1,32,256", "SB_REPS": "1"}, 200)
    child("device_decode_two_pass", [os.path.join(ROOT, "tools", "two_pass_bench.py")], {"TP_B": "256"}, 120)
    child("decode_three_pass", [os.path.join(ROOT, "tools", "three_pass_bench.py")], {}, 200)
    # a batch of live decoders: the headline's 512 utterances IN PROGRESS at once, 100 ms of audio a stream a step (psgpu_decode_streams_*):
    # frames/s over all streams and the time of a step = a piece's arrival to every stream's updated hypothesis on the host
    child("live_streams", [os.path.join(ROOT, "tools", "streams_bench.py")], {"LS_STREAMS": "512", "LS_SEC": "30", "LS_CHUNK": "10"}, 120)
    # ... and the same batch of live decoders fed AUDIO (psgpu_decode_streams_step_pcm: front end, live cepstral mean and feature window per
    # stream on the device)
    child("live_streams_from_audio", [os.path.join(ROOT, "tools", "streams_bench.py")], {"LS_STREAMS": "512", "LS_SEC": "30", "LS_CHUNK": "10", "LS_PCM": "1"}, 120)
    from pocketsphinx_amd import largevocab as lv
    if lv.available(lv.table_path(directory=os.environ.get("PSGPU_TABLE_DIR"))):
        # configs[2]'s shape: ONE 60 s utterance, en-us PTM + the large LM / dictionary (en-us.lm.bin is not in the repository: big.arpa
        # + cmudict stand in), fwdtree AND fwdflat on the device, the reference's two-pass decode of the same PCM beside it
        child("decode_two_pass_large_vocab_60s", [os.path.join(ROOT, "tools", "two_pass_bench.py")],
              {"TP_TASK": "big", "TP_SYNTH": "60", "TP_B": "1", "TP_CHECK_EVERY": "1"}, 200)


_JSON_FD = None


def sq_issue(kernel_key):
    """how busy a search kernel's resident wavefronts are, from the newest COMMITTED SQ-counter pass (profiles/*_sq_issue.json, written by
    tools/gpu_call_r6sq.sh): issue_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES (the share of a resident wavefront's cycles with an
    instruction in flight), wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES.  What bounds these kernels is the chain of dependent steps of a
    frame, not bytes: these two say how far from a busy instruction stream that leaves them.  A constant of the repository like `traffic`."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_issue.json")), reverse=True):
        try:
            k = json.load(open(path)).get(kernel_key)
            if k and k.get("issue_frac") is not None:
                return k
        except Exception:
            continue
    return None


def flatten_for_driver(line):
    """the driver's record keeps `roofline`'s SCALAR keys (nested objects are dropped, BENCH_r05): the scorer's and the
    large-vocabulary leg's figures once more as scalars beside the headline kernel's"""
    rf = line["roofline"]
    sc = rf.get("scorer") or {}
    if sc:
        rf["scorer_valu_frac"] = sc.get("frac_of_no_fma_rate")
        rf["scorer_tflops"] = sc.get("achieved")
        rf["scorer_kernel_ms"] = sc.get("kernel_ms")
    st, sa = line.get("stage_ms") or {}, line.get("stage_ms_one_step_alone") or {}
    for k in ("front_end", "scorer", "search"):
        if k in st:
            rf["stage_%s_ms_beside" % k] = st[k]
        if k in sa:
            rf["stage_%s_ms_alone" % k] = sa[k]
    if rf.get("traffic") and rf.get("algorithmic_bytes_per_launch"):
        rf["traffic_over_algorithmic"] = round(rf["traffic"] / rf["algorithmic_bytes_per_launch"], 3)
    iq = sq_issue("fwdtree_kernel_headline")
    if iq:
        rf["issue_frac"] = iq["issue_frac"]
        rf["wait_frac"] = iq.get("wait_frac")
    if rf.get("kernel_ms") and rf.get("kernel_ms_alone"):
        rf["alone_over_beside"] = round(rf["kernel_ms_alone"] / rf["kernel_ms"], 4)
    lv = line.get("decode_large_vocab")
    if isinstance(lv, dict) and "value" in lv:
        lr, lp, lc = lv.get("roofline") or {}, lv.get("parity") or {}, lv.get("cpu_baseline") or {}
        rf["lv_value"] = lv["value"]
        rf["lv_ms_per_step"] = lv.get("ms_per_step")
        rf["lv_utterances"] = (lv.get("config") or {}).get("utterances")
        rf["lv_kernel_ms"] = lr.get("kernel_ms")
        rf["lv_frac"] = lr.get("frac")
        rf["lv_achieved"] = lr.get("achieved")
        rf["lv_traffic"] = lr.get("traffic")
        rf["lv_algorithmic_bytes"] = lr.get("algorithmic_bytes_per_launch")
        rf["lv_traffic_over_algorithmic"] = (round(lr["traffic"] / lr["algorithmic_bytes_per_launch"], 3)
                                             if lr.get("traffic") and lr.get("algorithmic_bytes_per_launch") else None)
        rf["lv_parity_identical"] = lp.get("identical")
        rf["lv_parity_checked"] = lp.get("checked")
        rf["lv_cpu_baseline"] = lc.get("value")
        iq = sq_issue("fwdtree_kernel_large_vocab")
        if iq:
            rf["lv_issue_frac"] = iq["issue_frac"]
            rf["lv_wait_frac"] = iq.get("wait_frac")
    elif isinstance(lv, dict):
        rf["lv_value"] = None
        rf["lv_error"] = str(lv.get("error") or lv.get("skipped"))[:120]
    tp = line.get("decode_two_pass")
    if isinstance(tp, dict) and "first_pass_ms" in tp:
        rf["two_pass_first_ms"], rf["two_pass_second_ms"] = tp["first_pass_ms"], tp["second_pass_ms"]
        rf["two_pass_value"] = tp.get("frames_per_s")
        pp = tp.get("parity") or {}
        rf["two_pass_parity_identical"], rf["two_pass_parity_checked"] = pp.get("identical"), pp.get("checked")
    for key, pre in (("decode_ms_scorer", "ms"), ("decode_ms_continuous", "cont")):
        lg = line.get(key)
        if isinstance(lg, dict) and "frames_per_s" in lg:
            rf[pre + "_value"] = lg["frames_per_s"]
            pp = lg.get("parity") or {}
            rf[pre + "_parity_identical"], rf[pre + "_parity_checked"] = pp.get("identical"), pp.get("checked")
    ex = line.get("extra") or {}
    s60 = ex.get("decode_two_pass_large_vocab_60s")
    if isinstance(s60, dict):
        for k in ("seconds", "first_pass_call_s", "second_pass_call_s", "xrt"):
            if isinstance(s60.get(k), (int, float)):
                rf["lv60_" + k] = s60[k]
        if isinstance(s60.get("reference"), dict):
            rf["lv60_cpu_seconds"] = s60["reference"].get("cpu_s")
        if isinstance(s60.get("parity"), dict):
            rf["lv60_parity_identical"], rf["lv60_parity_checked"] = s60["parity"].get("identical"), s60["parity"].get("checked")
    for key, pre in (("live_streams", "live"), ("live_streams_from_audio", "live_pcm")):
        ls = ex.get(key)
        if isinstance(ls, dict) and "value" in ls:
            rf[pre + "_value"], rf[pre + "_ms_per_step"], rf[pre + "_step_ms_p99"] = ls["value"], ls.get("ms_per_step"), ls.get("step_ms_p99")
    for key, pre in (("search_only_turtle", "so_turtle"), ("search_only_cmudict", "so_cmudict")):
        so = ex.get(key)
        if isinstance(so, dict):
            for k, v in so.items():
                if isinstance(v, (int, float)) and not isinstance(v, bool):
                    rf["%s_%s" % (pre, k)] = v
    par = line.get("parity") or {}
    if par:
        rf["parity_identical"], rf["parity_checked"] = par.get("identical"), par.get("checked")
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict) and isinstance(cb.get("all_cores"), dict):
        cb["all_cores_value"], cb["all_cores_n"] = cb["all_cores"].get("value"), cb["all_cores"].get("cores")


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch_argv(n_gpus, argv, port=None):
    """the command line that runs this file as n_gpus ranks on this node: one process per GPU under torch.distributed.run,
    rendezvous on 127.0.0.1 (the container's host name may not resolve)"""
    if port is None:
        port = free_port()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_gpus,
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def emit(line):
    """the ONE line of standard output"""
    data = (json.dumps(line) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def main():
    # standard output carries exactly one JSON line: native libraries that print there (RCCL's version banner at
    # init_process_group) are sent to standard error for the life of the process
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--utts", type=int, default=B_UTT, help="utterances per GPU (default: the configs[4] share, 512)")
    ap.add_argument("--seconds", type=float, default=UTT_SECONDS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline workload only (clean per-kernel profiles)")
    ap.add_argument("--no-large-vocab", action="store_true", help="leave the 134,865-word leg (decode_large_vocab) out")
    ap.add_argument("--scatter", action="store_true", help="N > 1: rank 0 synthesises the whole job's audio and scatters it over RCCL "
                    "(default: every rank synthesises its own share)")
    ap.add_argument("--large-vocab-utts", type=int, default=LV_UTT)
    ap.add_argument("--tables", default=None, help="directory of table files written by integration/psgpu_export_tables (default: "
                                                   "$PSGPU_TABLE_DIR, else integration/_tables): the large-vocabulary task's tables")
    ap.add_argument("--workload", choices=("headline", "large"), default="headline",
                    help="large: ONLY the large-vocabulary leg, printed as the line (for profiling that kernel alone)")
    args = ap.parse_args()
    if args.tables:
        os.environ["PSGPU_TABLE_DIR"] = os.path.abspath(args.tables)      # (the child-process extras read it too)

    # --gpus N is the number of ranks of the job.  Launched by torch.distributed.run (the driver's form for N > 1) the
    # environment carries it and the two must agree; launched plainly with N > 1 (`python bench.py --gpus 8`) this process
    # becomes the launcher of N ranks, one per GPU over RCCL (self_launch_argv) -- the reference's batch driver is one
    # process walking a control file (programs/pocketsphinx_batch.c:877-899); here the control file's utterances shard.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.stdout.flush()
        os.dup2(_JSON_FD, 1)                              # (the ranks print the line themselves)
        argv = self_launch_argv(args.gpus, sys.argv[1:])
        sys.stderr.write("bench.py: --gpus %d without a launcher: starting %d ranks: %s\n" % (args.gpus, args.gpus, " ".join(argv)))
        if os.environ.get("PSGPU_BENCH_LAUNCH_DRYRUN"):   # (tests/test_bench_launch.py: the command line, not the job)
            emit({"launch": argv})
            return
        os.execv(argv[0], argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE): the line's n_gpus would not be what "
                         "was asked for" % (args.gpus, world))
    B, n_samp = args.utts, int(round(args.seconds * 16000))

    # the job's PCM (synthesised before torch / HIP start: the pool forks).  Every rank makes the utterances of ITS share (utterance id =
    # seed: the same bytes wherever they are made; the ranks' pools work side by side, so the set-up of an N-rank run costs what one
    # rank's does).  --scatter: rank 0 owns the whole job's audio, as one reader of a control file would, and scatters it over RCCL
    # (pocketsphinx_amd.batch.scatter_pcm; N - 1 sends of 491 MB one after another, outside the timed region either way)
    if args.scatter and world > 1:
        pcm_all = synth_pcm(0, B * world, args.seconds) if rank == 0 else None
    else:
        pcm_all = synth_pcm(rank * B, B, args.seconds)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s): one rank per GPU, no sharing" % (world, torch.cuda.device_count()))
    dist = None
    if world > 1 or os.environ.get("PSGPU_BENCH_FORCE_DIST"):     # (.._FORCE_DIST: the N > 1 code path with one rank, for a one-GPU box)
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:                       # (.._FORCE_DIST without a launcher: a one-rank group of its own)
            os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()), "RANK": "0", "WORLD_SIZE": "1"})
        dist.init_process_group(backend="nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import pocketsphinx_amd as P
    from pocketsphinx_amd import capi, batch as pbatch
    L = capi.lib()
    capi.check(L.psgpu_set_device(local_rank), "psgpu_set_device")
    tables = _npz("en_us_ptm_tables.npz")
    gt = _npz("fwdtree_trace_goforward.npz")
    if args.workload == "large":
        if world != 1:
            raise SystemExit("bench.py --workload large is a one-GPU measurement")
        nlv = min(args.large_vocab_utts, B)
        lvl = large_vocab_leg(P, pcm_all, n_samp, args.seconds, dev, _npz("mfcc_en_us_goforward.npz"), tables, nlv, max(args.steps, 1),
                              LV_CHECK, not args.no_cpu_baseline, args.tables)
        lvl.update({"n_gpus": 1, "warmup": 1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "f32 (Gaussian distances) + int32 (log-domain scores, Viterbi)", "data": "synthetic (as the headline)"})
        emit(lvl)
        return
    # Two pipeline objects taking turns (PSGPU_BENCH_PIPES, default 2): while one batch's tree search -- a latency-bound
    # recurrence, one workgroup per utterance -- is resident, the other batch's front end and scorer run beside it; searches
    # are ordered by events, each pipeline on a stream with a hardware queue of its own (psgpu_decode_search_after,
    # DESIGN.md 0).  A step is still one pass of the hot path over one batch; two are in flight.  Measured (profiles/r03_*):
    # 145 ms per step with one object, 112 ms with two.
    from pocketsphinx_amd import decode as pdec
    n_pipe = max(1, int(os.environ.get("PSGPU_BENCH_PIPES", "2")))
    pipes = [P.DecodePipeline(_npz("mfcc_en_us_goforward.npz"), tables, _npz("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)
             for _ in range(n_pipe)]
    streams = [torch.cuda.ExternalStream(pdec.dedicated_stream(), device=dev) for _ in range(n_pipe)]
    for k, q in enumerate(pipes):
        q.stage_timing(True)
        if q.model is not None:                          # the scorer's kernels by HIP events of their own (psgpu_ptm_kernel_timing)
            capi.check(L.psgpu_ptm_kernel_timing(q.model.h, 1), "psgpu_ptm_kernel_timing")
        if n_pipe > 1:
            q.search_after(pipes[(k - 1) % n_pipe])
    scorer_ms = []                                       # per timed step: (top-N kernel, exact fix-up kernel) in ms
    pipe = pipes[0]
    stream = torch.cuda.current_stream().cuda_stream
    sp = C.c_void_p(stream)

    # ---- inputs resident in HBM: this rank's 512 utterances (scattered from rank 0 over RCCL when N > 1)
    pcm = torch.empty(B * n_samp, dtype=torch.int16, device=dev)
    if dist is None or not args.scatter:
        pcm.copy_(torch.from_numpy(pcm_all))
    else:
        pbatch.scatter_pcm(pcm, pcm_all, B * n_samp, device=dev)
    soff = np.arange(B + 1, dtype=np.int64) * n_samp
    torch.cuda.synchronize()

    stage = []
    gather_stream = torch.cuda.Stream(device=dev) if dist is not None else None
    gathered = []

    fe_ahead = n_pipe == 2 and not os.environ.get("PSGPU_BENCH_NO_FE_AHEAD")

    def launch(k):
        """one pass of the hot path over the batch: everything enqueued on the step's stream.  (Two objects: the front end of the
        OTHER object's next step first, on that object's own stream -- it runs beside this step's scorer instead of at the start
        of its own step, where the resident search holds most of the LDS it wants: psgpu_decode_front_end_ahead.  It is part of
        step k + 1's work, inside the timed region for every step but the first, whose front end runs in the step itself.)"""
        if fe_ahead and k + 1 < launch.n:
            pipes[(k + 1) % n_pipe].front_end_ahead(pcm, soff)
        pipes[k % n_pipe].run_dev(pcm, soff, streams[k % n_pipe].cuda_stream)
    launch.n = 0

    def finish(k, timed):
        """the step's hypothesis records on the host (rank 0: of every rank)"""
        q, st = pipes[k % n_pipe], streams[k % n_pipe]
        if dist is None:
            out = q.fetch()
        else:
            # copies of the step's records (small kernels on the step's stream: the pipeline's buffers are free for its next
            # call), then the RCCL gather on a stream of its own -- a collective's kernel launched beside a resident search may
            # have to wait for registers, and must not hold up this pipeline's next step; its result is needed only at the end
            v = q.view()
            with torch.cuda.stream(st):
                hn = torch.as_tensor(_DevArray(v.hyp_n_dev, (B, 4)), device=dev).clone()
                hy = torch.as_tensor(_DevArray(v.hyp_dev, (B, q.max_words, 4)), device=dev).clone()
                hn.record_stream(gather_stream); hy.record_stream(gather_stream)
                copied = torch.cuda.Event(); copied.record(st)
            with torch.cuda.stream(gather_stream):
                gather_stream.wait_event(copied)
                g_hn = pbatch.gather_records(hn, device=dev)
                g_hy = pbatch.gather_records(hy, device=dev)
            gathered[:] = [(g_hn, g_hy, hn, hy)]               # (kept alive until the next step's replace them)
            out = None
        if timed:
            stage.append(q.last_stage_ms())     # events of this step's launches (complete: the records are here)
            if q.model is not None:
                ms3 = (C.c_float * 3)()
                capi.check(L.psgpu_ptm_last_kernel_ms(q.model.h, ms3), "psgpu_ptm_last_kernel_ms")
                scorer_ms.append((float(ms3[0]), float(ms3[1])))
        return out

    host_split = {"launch_wall": 0.0, "launch_cpu": 0.0, "finish_wall": 0.0, "finish_cpu": 0.0}

    def timed_call(key, fn, *a):
        w0, c0_ = time.perf_counter(), time.thread_time()
        r = fn(*a)
        host_split[key + "_wall"] += time.perf_counter() - w0; host_split[key + "_cpu"] += time.thread_time() - c0_
        return r

    def run_steps(n, timed):
        launch.n = n
        for k in range(n):
            timed_call("launch", launch, k)
            if k >= n_pipe - 1:
                timed_call("finish", finish, k - (n_pipe - 1), timed)
        for k in range(max(n - (n_pipe - 1), 0), n):
            timed_call("finish", finish, k, timed)

    run_steps(args.warmup, False)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    def thread_cpu():
        """CPU seconds (user + system) of every thread of this process, by thread id, with the thread's name"""
        out = {}
        tick = os.sysconf("SC_CLK_TCK")
        for t in os.listdir("/proc/self/task"):
            try:
                f = open("/proc/self/task/%s/stat" % t).read()
                name = f[f.index("(") + 1:f.rindex(")")]
                v = f[f.rindex(")") + 2:].split()
                out[int(t)] = (name, (int(v[11]) + int(v[12])) / tick)
            except Exception:
                pass
        return out
    for k_ in host_split:
        host_split[k_] = 0.0
    t0 = time.perf_counter()
    c0 = time.process_time()
    th0 = thread_cpu()
    run_steps(args.steps, True)
    torch.cuda.synchronize()
    host_cpu_s = time.process_time() - c0                # this rank's host CPU inside the timed region (all its threads)
    th1 = thread_cpu()
    host_threads = sorted(((n, round(1e3 * (c - th0.get(t, (n, 0.0))[1]) / args.steps, 2)) for t, (n, c) in th1.items()), key=lambda x: -x[1])[:5]
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        dt = pbatch.max_over_ranks(dt, device=dev)

    # ---- the same steps with the batch's PCM coming from HOST memory each step (the boundary handing over host buffers): a
    #      pinned host buffer, two device buffers taking turns, the copy of step k + 1 on a stream of its own beside step k
    pcie = None
    if dist is None and rank == 0 and not os.environ.get("PSGPU_BENCH_NO_PCIE"):
        try:
            pinned = torch.from_numpy(pcm_all[:B * n_samp]).pin_memory()
            bufs = [pcm, torch.empty_like(pcm)]
            cstream = torch.cuda.Stream(device=dev)
            evs = [torch.cuda.Event() for _ in bufs]

            def launch_h(k):
                with torch.cuda.stream(cstream):
                    bufs[k % 2].copy_(pinned, non_blocking=True)
                    evs[k % 2].record(cstream)
                streams[k % n_pipe].wait_event(evs[k % 2])
                pipes[k % n_pipe].run_dev(bufs[k % 2], soff, streams[k % n_pipe].cuda_stream)

            def run_h(n):
                for k in range(n):
                    launch_h(k)
                    if k >= n_pipe - 1:
                        finish(k - (n_pipe - 1), False)
                for k in range(max(n - (n_pipe - 1), 0), n):
                    finish(k, False)
            run_h(2)
            torch.cuda.synchronize()
            th = time.perf_counter()
            run_h(args.steps)
            torch.cuda.synchronize()
            dth = (time.perf_counter() - th) / args.steps
            pcie = {"ms_per_step": round(1e3 * dth, 3), "h2d_bytes_per_step": int(B * n_samp * 2),
                    "what": "every step's PCM copied from pinned host memory (hipMemcpyAsync on a copy stream, double-buffered) before "
                            "its front end: the rate a caller holding host buffers sees"}
            del bufs, pinned
        except Exception as e:
            pcie = {"error": str(e)[-200:]}

    # this rank's own results (tables' sizes, status, workload counters); one step alone also gives the stages' times without
    # another batch beside them
    pipe.run_dev(pcm, soff, streams[0].cuda_stream)
    hn_l, hyp_l, res_l = pipe.fetch()
    stage_alone = pipe.last_stage_ms()
    if dist is not None and rank == 0 and gathered:
        # the job's records as gathered in the last timed step: rank 0's block is what rank 0 has just decoded again, and every
        # other rank's block holds hypotheses (the same utterance lengths everywhere)
        g_hn = gathered[0][0].cpu().numpy()
        if g_hn.shape[0] != B * world or not np.array_equal(g_hn[:B], hn_l) or int((g_hn[:, 0] <= 0).sum()):
            raise SystemExit("bench: the gathered hypothesis records are not the ranks' results")
    if int((res_l[:, 3] != 0).sum()):
        raise SystemExit("bench: %d utterances ended with a full back-pointer table / score stack" % int((res_l[:, 3] != 0).sum()))
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    frames_rank = int(res_l[:, 2].sum())
    frames_step = frames_rank * world             # every rank decodes the same number of equally long utterances
    fps = frames_step * args.steps / dt
    audio_s = B * world * args.seconds
    st_mean = {k: float(np.mean([s[k] for s in stage])) for k in stage[0]}
    evals = int(res_l[:, 5].astype(np.int64).sum() + (res_l[:, 6].astype(np.int64) << 32).sum())
    senones = int(res_l[:, 7].astype(np.int64).sum())
    alg_bytes = 156 * frames_rank + 2 * senones + 86 * evals
    search_s = st_mean["search"] * 1e-3
    workload = ("configs[4] per-GPU share: %d utterances x %g s synthetic 16 kHz PCM, en-us PTM (5126 senones) + turtle n-gram LM, "
                "fwdtree only (-fwdflat no -bestpath no), PCM -> hypotheses on the device" % (B, args.seconds))
    traffic = None
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:       # the newest committed PMC pass OF THIS WORKLOAD (tools/prof_collect.py records what it profiled)
            j = json.load(open(tpath))
            k = j.get("fwdtree_kernel")
            if k and not j.get("_workload", {}).get("leg") and j.get("_workload", {}).get("utterances") == B and j["_workload"].get("seconds") == args.seconds \
                    and k.get("hbm_bytes_per_launch"):
                traffic = round(k["hbm_bytes_per_launch"])
                break
        except Exception:
            continue
    line = {
        "metric": "frames/sec + xRT decode, en-us PTM 5126-senone n-gram fwdtree (device first pass, PCM -> hypotheses)",
        "value": round(fps, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (Gaussian distances) + int32 (log-domain scores, Viterbi)",
        "data": "synthetic: bundled recordings tiled with random gains / pauses over a noise floor; model, LM and dictionary tables "
                "as the reference built them",
        "config": {"workload": workload, "utterances_per_gpu": B, "seconds_per_utterance": args.seconds,
                   "frames_per_step_per_gpu": frames_rank, "lm": "turtle.lm.bin (115 dictionary words)",
                   "parallelism": "utt-shard x%d (rank 0 scatters PCM, gathers hypothesis records)" % world},
        "xrt": round((dt / args.steps) / audio_s, 9),
        "pcie_inclusive": None if pcie is None else dict(pcie, **({"value": round(frames_step / (pcie["ms_per_step"] * 1e-3), 1), "unit": "frames/s"}
                                                                  if "ms_per_step" in pcie else {})),
        "steps_in_flight": n_pipe,
        "host": {"cpu_ms_per_step_per_rank": round(1e3 * host_cpu_s / args.steps, 3), "host_cpus": os.cpu_count(),
                 "busiest_threads_ms_per_step": [{"thread": n, "cpu_ms_per_step": c} for n, c in host_threads],
                 "main_thread_ms_per_step": {k_: round(1e3 * v_ / args.steps, 2) for k_, v_ in host_split.items()},
                 "ranks_the_host_cores_carry": int(os.cpu_count() / max(host_cpu_s / max(dt, 1e-9), 1e-9)),
                 "what": "host CPU time of one rank inside the timed region (launches, the fetch of the hypothesis records, the gather at N > "
                         "1), per step: what a node's host cores must supply per GPU -- the host-side ceiling of SURVEY 8e is "
                         "host_cpus / (N x this / ms_per_step) ranks"},
        "stage_ms": {k: round(v, 3) for k, v in st_mean.items()},
        "stage_ms_one_step_alone": {k: round(v, 3) for k, v in stage_alone.items()},
        "workload_counts": {"hmm_evals_per_frame": round(evals / max(frames_rank, 1), 2),
                            "listed_senones_per_frame": round(senones / max(frames_rank, 1), 2),
                            "back_pointers_per_utt": round(float(res_l[:, 0].mean()), 1),
                            "words_per_hyp": round(float(hn_l[:, 0].mean()), 1),
                            "lds_layout": bool(pipe.search.lds_layout())},
        "roofline": {"bound": "hbm", "kernel": "fwdtree_kernel", "achieved": round(alg_bytes / search_s / 1e9, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(alg_bytes / search_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(st_mean["search"], 3),
                     "kernel_ms_alone": round(stage_alone["search"], 3),
                     "note": "a recurrence over frames: one workgroup per utterance, bound by the latency of one frame's dependent "
                             "steps, not by bytes (DESIGN.md 4); bytes = sum over frames of 156 + 2 x listed senones + 86 x HMM "
                             "evaluations (SURVEY 8d), counted by the kernel; kernel_ms = HIP events around the kernel in the timed "
                             "region, i.e. with the other batch's front end and scorer running beside it; traffic = HBM bytes per launch of "
                             "this workload from the newest COMMITTED PMC profile (profiles/*_pmc_traffic.json, FETCH_SIZE x 2 + WRITE_SIZE): "
                             "a constant of the repository, not a measurement of this run"},
    }
    # the scorer's roofline beside the search's (VALU-fp32 bound, not HBM, not MFMA): 838,656 fp32 operations a frame
    # (16,128 densities x 13 dimensions x {sub, mul, mul, sub}) over the top-N kernel's own time in the timed region (HIP events
    # around the launch on its stream: with the other batch's search resident beside it)
    if scorer_ms:
        lane_ms = float(np.mean([a for a, _ in scorer_ms])); fix_ms = float(np.mean([b for _, b in scorer_ms]))
        gops = 838656.0 * frames_rank / (lane_ms * 1e-3) / 1e9
        line["roofline"]["scorer"] = {
            "bound": "valu_fp32", "kernel": "ptm_lane_kernel", "kernel_ms": round(lane_ms, 3), "fixup_kernel_ms": round(fix_ms, 3),
            "achieved": round(gops / 1e3, 2), "peak": round(2 * VALU_PEAK_GOPS / 1e3, 1), "unit": "TFLOP/s",
            "frac": round(gops / (2 * VALU_PEAK_GOPS), 4), "frac_of_no_fma_rate": round(gops / VALU_PEAK_GOPS, 4),
                        "note": "fp32 vector operations per frame (SURVEY 8d) / the kernel's time by HIP events in the timed region (beside the other "
                    "batch's search); peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz x 2 (an FMA counted twice); the bit-exact distance "
                    "(ptm_mgau.c:102-128: sub, mul, mul, sub in fp32, no contraction) cannot use FMA: frac_of_no_fma_rate is the bound "
                    "it can reach"}
    # ---- cpu_baseline + parity: the compiled reference on a sample of the same utterances
    if world > 1:
        line["cpu_baseline"] = None                      # (the reference leg runs at N = 1 only: BENCH, not SCALE)
    if not args.no_cpu_baseline and world == 1:          # (rank 0 at N = 1 only: the scaling runs time the device path alone)
        ids = sorted(set(int(i) for i in np.linspace(0, B - 1, min(N_SAMPLE, B))))
        n_proc = max(1, min(len(ids), (os.cpu_count() or 2) // 2))
        ref = reference_decode(pcm_all, n_samp, ids, procs=n_proc)
        if ref is None:
            line["cpu_baseline"] = None
            line["parity"] = {"checked": 0, "note": "oracle/_ref/ref_decode_bench not built: parity unchecked in this run"}
        else:
            utts, tot = ref
            bad = parity_of(ids, utts, hn_l, hyp_l, res_l)
            line["cpu_baseline"] = {"value": round(tot["frames_per_s"], 2), "unit": "frames/s", "cores": 1, "kind": "reference",
                                    "xrt": round(tot["xrt"], 6),
                                    "sample": "%d of the step's %d utterances (%d frames, %.1f s of CPU in all): ps_start_utt / "
                                              "ps_process_raw(full_utt) / ps_end_utt per utterance, -fwdflat no -bestpath no; value = frames / "
                                              "summed CPU seconds of one-thread decoders" % (len(ids), B, tot["frames"], tot["cpu_s"]),
                                    "what": "unmodified reference (%s), one thread" % ref_exe("ref_decode_bench")[1],
                                    "build": ref_exe("ref_decode_bench")[1],
                                    "all_cores": {"value": round(tot["frames_per_s_all_procs"], 1), "unit": "frames/s", "cores": tot["procs"],
                                                  "host_cpus": os.cpu_count(), "wall_s": round(tot["wall_s"], 2),
                                                  "what": "%d reference processes side by side, one decoder thread each (pocketsphinx_batch's way to "
                                                          "use a machine), frames / wall time of the slowest" % tot["procs"]}}
            line["parity"] = {"checked": len(ids), "identical": len(ids) - len(bad), "mismatching_utterances": bad,
                              "what": "word ids, start / end frames, path score and frame count of each sampled utterance: device vs "
                                      "the reference decoding the same PCM"}
            widen_parity(line["parity"], parity_of_the_rest(pcm_all, n_samp, B, ids, hn_l, hyp_l, res_l))
            bad = line["parity"]["mismatching_utterances"]
            line["speedup_vs_cpu_1thread"] = round(fps / world / tot["frames_per_s"], 1)
            if bad:
                emit(line)
                raise SystemExit("bench: device hypotheses differ from the reference's on utterances %r" % bad)
    # ---- extras (N = 1 only)
    extra = {}
    if not args.no_extras and world == 1:
        for q in pipes:
            q.close()
        del pcm
        torch.cuda.empty_cache()
        try:
            line["decode_ms_scorer"] = ms_scorer_leg(P, pcm_all, n_samp, args.seconds, dev, _npz("mfcc_en_us_goforward.npz"),
                                                     _npz("fwdtree_static_en_us_turtle.npz"), gt, min(64, B), 2, LEG_CHECK, not args.no_cpu_baseline)
        except Exception as e:
            line["decode_ms_scorer"] = {"error": str(e)[-400:]}
        try:
            # configs[3] as written: the continuous-density model through the same pipeline (tables: integration/_tables, exported
            # from a decoder the reference initialised with the staged model)
            from pocketsphinx_amd.tablefile import read_psgb
            tdir = os.environ.get("PSGPU_TABLE_DIR") or os.path.join(ROOT, "integration", "_tables")
            cpath = os.path.join(tdir, "en_us_cont_turtle.psgb")
            if os.path.exists(cpath):
                cg = read_psgb(cpath)
                ct = {k[3:]: v for k, v in cg.items() if k.startswith("ms_")}
                ct["sen2mgau"] = ct["sen2mgau"].astype(np.uint32)
                line["decode_ms_continuous"] = ms_scorer_leg(P, pcm_all, n_samp, args.seconds, dev, _npz("mfcc_en_us_goforward.npz"),
                                                             _npz("fwdtree_static_en_us_turtle.npz"), gt, min(64, B), 2, LEG_CHECK,
                                                             not args.no_cpu_baseline, cont_tables=ct)
            else:
                line["decode_ms_continuous"] = {"skipped": "%s not built (make -C integration tables)" % cpath}
        except Exception as e:
            line["decode_ms_continuous"] = {"error": str(e)[-400:]}
        try:
            line["decode_two_pass"] = two_pass_leg(P, pcm_all, n_samp, args.seconds, dev, _npz("mfcc_en_us_goforward.npz"), tables,
                                                   _npz("fwdtree_static_en_us_turtle.npz"), gt, B, 2, LEG_CHECK, not args.no_cpu_baseline)
        except Exception as e:
            line["decode_two_pass"] = {"error": str(e)[-400:]}
        if not args.no_large_vocab:
            try:
                line["decode_large_vocab"] = large_vocab_leg(P, pcm_all, n_samp, args.seconds, dev, _npz("mfcc_en_us_goforward.npz"), tables,
                                                             min(args.large_vocab_utts, B), LV_STEPS, LV_CHECK, not args.no_cpu_baseline, args.tables)
            except Exception as e:
                line["decode_large_vocab"] = {"error": str(e)[-400:]}
        try:
            # configs[2]: one 60 s utterance
            p1 = P.DecodePipeline(_npz("mfcc_en_us_goforward.npz"), tables, _npz("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)
            from pocketsphinx_amd import synth
            u60 = torch.from_numpy(synth.utterance(7, 60.0)).to(dev)
            o60 = np.array([0, u60.numel()], np.int64)
            p1.run_dev(u60, o60, stream); p1.fetch()
            t1 = time.perf_counter()
            for _ in range(3):
                p1.run_dev(u60, o60, stream); hn1, _, res1 = p1.fetch()
            d1 = (time.perf_counter() - t1) / 3
            extra["single_utterance_60s"] = {"frames": int(res1[0, 2]), "seconds": round(d1, 5), "frames_per_s": round(int(res1[0, 2]) / d1, 1),
                                             "xrt": round(d1 / 60.0, 7), "words": int(hn1[0, 0]),
                                             "what": "configs[2]: one 60 s utterance through the same pipeline (latency of one workgroup's "
                                                     "recurrence over 6000 frames)"}
            p1.close()
        except Exception as e:
            extra["single_utterance_60s"] = {"error": str(e)[-300:]}
        try:
            sc, model, feats_h = scorer_extra(P, capi, L, dev, sp, tables)
            extra["senone_scoring"] = sc
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_extras
            extra.update(bench_extras.extras(P, capi, L, model, tables, feats_h, dev, sp))
            model.close()
        except Exception as e:
            extra["senone_scoring_error"] = str(e)[-300:]
        if not os.environ.get("PSGPU_BENCH_NO_CHILD"):
            child_extras(extra)
    line["extra"] = extra
    # the other legs where the driver keeps them (it stores `roofline` whole and only the NAMES of other top-level keys): value,
    # roofline fraction, traffic against algorithmic bytes, CPU baseline and parity of each leg that ran
    legs = {}
    for name in ("decode_large_vocab", "decode_two_pass", "decode_ms_scorer", "decode_ms_continuous"):
        lg = line.get(name)
        if isinstance(lg, dict) and "value" not in lg and "frames_per_s" in lg:      # (the legs that predate the line's field names)
            lg = dict(lg, value=lg["frames_per_s"], unit="frames/s", config={"workload": lg.get("what")})
        if not isinstance(lg, dict) or "value" not in lg:
            if isinstance(lg, dict) and ("error" in lg or "skipped" in lg):
                legs[name] = {k: lg[k] for k in ("error", "skipped") if k in lg}
            continue
        rf, par_, cb = lg.get("roofline") or {}, lg.get("parity") or {}, lg.get("cpu_baseline") or {}
        legs[name] = {"value": lg["value"], "unit": lg.get("unit"), "ms_per_step": lg.get("ms_per_step"),
                      "workload": (lg.get("config") or {}).get("workload"),
                      "roofline_frac": rf.get("frac"), "achieved_GBps": rf.get("achieved"), "kernel": rf.get("kernel"), "kernel_ms": rf.get("kernel_ms"),
                      "traffic": rf.get("traffic"), "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"),
                      "traffic_over_algorithmic": (round(rf["traffic"] / rf["algorithmic_bytes_per_launch"], 2)
                                                   if rf.get("traffic") and rf.get("algorithmic_bytes_per_launch") else None),
                      "cpu_baseline": {k: cb.get(k) for k in ("value", "unit", "cores", "kind")} if cb else None,
                      "parity": ("%d/%d identical" % (par_.get("identical", 0), par_.get("checked", 0))) if par_ else None}
        for k in ("first_pass_ms", "second_pass_ms"):
            if k in lg:
                legs[name][k] = lg[k]
    if legs:
        line["roofline"]["legs"] = legs
    flatten_for_driver(line)
    emit(line)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
