#!/usr/bin/env python3
"""Condense the rocprofv3 CSV outputs of tools/gpu_round.sh into two small
files that are committed under profiles/:

  profiles/<tag>_kernel_stats.txt   `rocprofv3 --kernel-trace --stats` summary
  profiles/<tag>_pmc_traffic.json   per-kernel HBM bytes per launch from the
                                    FETCH_SIZE / WRITE_SIZE passes, raw and
                                    corrected as MI355X_MICROARCH.md prescribes
                                    (units: the counters are in KB; on gfx950
                                    FETCH_SIZE reports 1/2 of the bytes of wide
                                    coalesced reads -> x2; WRITE_SIZE
                                    uncalibrated, reported as is)

The PMC averages are taken over the dispatches of ONE workload only (the bench is profiled with --no-extras, and
dispatches are filtered by grid size where a kernel is launched with several): the file records which workload
(`_workload`), and bench.py uses a file's number only for that workload.

usage: prof_collect.py gpurun_out/<tag> <tag> [utterances seconds]
"""
import csv
import glob
import json
import os
import sys


def find(root, pat):
    return sorted(glob.glob(os.path.join(root, "**", pat), recursive=True))


def col(row, *names):
    for n in names:
        for k in row:
            if k and k.strip().lower() == n.lower():
                return row[k]
    return None


def short(name):
    n = name.replace("void ", "")
    for ch in "<(":
        if ch in n:
            n = n.split(ch)[0]
    return n.strip()


def main():
    root, tag = sys.argv[1], sys.argv[2]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(repo, "gpurun_out", "profiles_" + tag)
    os.makedirs(prof, exist_ok=True)

    # ---- kernel stats
    cmd = ("bench.py --workload large --steps 1 --no-cpu-baseline --utts N --large-vocab-utts N (tools/gpu_call_lvpmc.sh)" if len(sys.argv) >= 6
           else "bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1")
    lines = ["# rocprofv3 --kernel-trace --stats -- python " + cmd,
             "# (durations in microseconds)",
             "%-8s %-14s %-12s %-8s %-10s %-10s %s" % ("calls", "total_us", "avg_us", "pct", "min_us", "max_us", "kernel")]
    stats = {}
    for f in find(os.path.join(root, "stats"), "*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            name = col(r, "Name")
            calls = int(col(r, "Calls"))
            tot = float(col(r, "TotalDurationNs")) / 1e3
            avg = float(col(r, "AverageNs")) / 1e3
            pct = float(col(r, "Percentage"))
            mn = float(col(r, "MinNs") or 0) / 1e3
            mx = float(col(r, "MaxNs") or 0) / 1e3
            stats[short(name)] = {"calls": calls, "avg_us": avg}
            lines.append("%-8d %-14.3f %-12.3f %-8.3f %-10.3f %-10.3f %s" % (calls, tot, avg, pct, mn, mx, name[:120]))
    # register / LDS footprint per kernel from the trace
    regs = {}
    for f in find(os.path.join(root, "stats"), "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k = short(col(r, "Kernel_Name") or "")
            if k in regs:
                continue
            regs[k] = {x: col(r, x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                              "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    lines.append("")
    lines.append("# per-kernel launch shape (first dispatch)")
    for k, v in regs.items():
        lines.append("%s: %s" % (k, v))
    txt = "\n".join(lines) + "\n"
    open(os.path.join(prof, "%s_kernel_stats.txt" % tag), "w").write(txt)
    print(txt)

    # ---- PMC traffic
    res = {}
    for sub, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        acc, grids = {}, {}
        for f in find(os.path.join(root, sub), "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if (col(r, "Counter_Name") or "") != cname:
                    continue
                k = short(col(r, "Kernel_Name") or "")
                acc.setdefault(k, []).append(float(col(r, "Counter_Value")))
                grids.setdefault(k, []).append(col(r, "Grid_Size"))
        for k, v in acc.items():
            if not (k.startswith("ptm_") or k.startswith("hmm_") or k.startswith("psgpu") or k.startswith("semi_")
                    or k.startswith("ms_") or k.startswith("fwdtree_") or k.startswith("phone_loop") or k.startswith("fe_")
                    or k.startswith("feat_") or k.startswith("fwdflat_")):
                continue
            # one workload only: the dispatches with the most common grid size
            gs = grids.get(k, [])
            if gs and len(set(gs)) > 1:
                top = max(set(gs), key=gs.count)
                v = [x for x, g_ in zip(v, gs) if g_ == top]
                res.setdefault(k, {})["grid_size_kept"] = top
            # steady state: drop the first (cold) dispatch
            vv = v[1:] if len(v) > 1 else v
            res.setdefault(k, {})[cname + "_KB_avg_per_launch"] = sum(vv) / len(vv)
            res[k]["n_launches_" + cname] = len(v)
    for k, d in res.items():
        f_kb = d.get("FETCH_SIZE_KB_avg_per_launch")
        w_kb = d.get("WRITE_SIZE_KB_avg_per_launch")
        d["hbm_read_bytes_corrected"] = None if f_kb is None else f_kb * 1024.0 * 2.0
        d["hbm_write_bytes"] = None if w_kb is None else w_kb * 1024.0
        if f_kb is not None and w_kb is not None:
            d["hbm_bytes_per_launch"] = d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]
    if len(sys.argv) >= 5:
        res["_workload"] = {"utterances": int(sys.argv[3]), "seconds": float(sys.argv[4]),
                            "command": "bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"}
        if len(sys.argv) >= 6:                               # a leg's own pass (tools/gpu_call_lvpmc.sh): bench.py matches on the leg's name
            res["_workload"]["leg"] = sys.argv[5]
            res["_workload"]["command"] = ("bench.py --workload large --steps 1 --no-cpu-baseline --utts N --large-vocab-utts N "
                                           "(N = utterances; tools/gpu_call_lvpmc.sh)")
    res["_note"] = ("FETCH_SIZE/WRITE_SIZE are reported by rocprofv3 in KB; read bytes doubled per "
                    "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncalibrated")
    json.dump(res, open(os.path.join(prof, "%s_pmc_traffic.json" % tag), "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
