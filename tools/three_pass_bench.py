#!/usr/bin/env python3
"""The reference's DEFAULT configuration -- lexicon-tree pass, flat-lexicon pass, lattice best path (-fwdflat yes -bestpath yes,
src/config_macro.h) -- through the additive batch call of the C binding (integration/psgpu_decode_batch.c,
PSGPU_BATCH_DEVICE_FIRST_PASS + PSGPU_DEVICE_SECOND_PASS=1): both search passes of the whole batch on the MI355X, the second pass's
tables injected into the workers' decoders, ngram_search_finish -> ps_lattice_bestpath (ngram_search.c:782, ps_lattice.c:1216) on
the host threads.  Timed on TP3_B utterances of TP3_SECONDS each (the benchmark's generator); a sample of TP3_CHECK utterances is
decoded again by the same program with its comparison on: every result against a fresh unmodified CPU decoder's.  Prints ONE JSON
line.  Run by bench.py in a child process.  TEST / MEASUREMENT tool: the program it runs (oracle/_ref/batch_api_check) links the
compiled reference, which is the point -- the drop-in binding is what is measured."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref")


def run(files, workers, timing_only, extra_env=None):
    env = dict(os.environ, PSGPU_DEVICE_SECOND_PASS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    if timing_only:
        env["BATCH_CHECK_TIMING_ONLY"] = "1"
    else:
        env.pop("BATCH_CHECK_TIMING_ONLY", None)
    argv = [os.path.join(REF, "batch_api_check"), os.path.join(REF, "model", "en-us"), os.path.join(REF, "data", "turtle.lm.bin"),
            os.path.join(REF, "data", "turtle.dic"), str(workers), "16"] + files + ["--", "fwdflat", "yes", "bestpath", "yes"]
    p = subprocess.run(argv, capture_output=True, text=True, timeout=900, env=env)
    if os.environ.get("PSGPU_BATCH_TIMING"):              # (the binding's own account of a call, integration/psgpu_decode_batch.c)
        sys.stderr.write("".join(ln + "\n" for ln in p.stderr.splitlines() if ln.startswith("psgpu_")))
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    if not lines:
        raise RuntimeError("batch_api_check rc %d: %s" % (p.returncode, (p.stderr or p.stdout)[-400:]))
    return json.loads(lines[-1]), p.returncode


def main():
    from pocketsphinx_amd import synth
    B = int(os.environ.get("TP3_B", "128")); seconds = float(os.environ.get("TP3_SECONDS", "30")); n_check = int(os.environ.get("TP3_CHECK", "32"))
    workers = int(os.environ.get("TP3_WORKERS", str(max(1, min(32, (os.cpu_count() or 2) // 2)))))
    if not os.path.exists(os.path.join(REF, "batch_api_check")):
        print(json.dumps({"skipped": "oracle/_ref/batch_api_check not built"}))
        return
    with tempfile.TemporaryDirectory() as td:
        files = []
        for i in range(B):
            f = os.path.join(td, "u%04d.raw" % i)
            synth.utterance(i, seconds).tofile(f)
            files.append(f)
        t, _ = run(files, workers, True)
        ids = sorted(set(int(x) for x in np.linspace(0, B - 1, min(n_check, B))))
        c, rc = run([files[i] for i in ids], min(workers, len(ids)), False)
    out = {"frames_per_s": t["frames_per_s"], "utterances": B, "frames": t["frames"], "seconds_per_utterance": seconds, "batch_s": t["batch_s"],
           "xrt": round(t["batch_s"] / (B * seconds), 8), "host_threads": workers,
           "parity": {"checked": c["B"], "identical": c["B"] - c["mismatch_batch"], "mismatch_reversed_order": c["mismatch_reversed"],
                      "mismatch_one_at_a_time": c["mismatch_single"],
                      "what": "hypothesis string, path score, frame count and every segment (word, frames, acoustic / language score, "
                              "back-off) of each sampled utterance: the batch call vs a fresh unmodified CPU decoder"},
           "cpu_baseline": {"value": round(c["frames"] / c["cpu_s"], 1) if c["cpu_s"] > 0 else None, "unit": "frames/s", "cores": 1, "kind": "reference",
                            "sample": "%d utterances, %.1f s of CPU in all (fresh decoder each, ps_init included), -fwdflat yes -bestpath yes" % (c["B"], c["cpu_s"])},
           "what": "the reference's default three passes on %d x %g s: fwdtree and fwdflat of the whole batch on the device (one pipeline object), "
                   "lattice generation + best path on %d host threads over the injected tables, results through the reference's own "
                   "ps_get_hyp / ps_seg_iter (integration/psgpu_decode_batch.c)" % (B, seconds, workers)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
