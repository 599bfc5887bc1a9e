#!/usr/bin/env python3
"""Read a rocprofv3 --kernel-trace CSV of tools/overlap_probe.py and say which kernels ran beside which: per kernel name the
number of dispatches, mean duration, and the mean fraction of its run time during which a fwdtree_kernel dispatch of ANOTHER
queue was running.   usage: overlap_trace.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys


def short(name):
    n = name.replace("void ", "")
    for ch in "<(":
        if ch in n:
            n = n.split(ch)[0]
    return n.strip()


def main():
    rows = []
    for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            rows.append((short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?") + "/s" + r.get("Stream_Id", "?")))
    rows.sort(key=lambda x: x[1])
    t0 = rows[0][1]
    srch = [(s, e, q) for k, s, e, q in rows if k == "fwdtree_kernel"]
    agg = {}
    for k, s, e, q in rows:
        ov = 0
        for ss, ee, qq in srch:
            if (ss, ee, qq) != (s, e, q):
                ov += max(0, min(e, ee) - max(s, ss))
        a = agg.setdefault(k, [0, 0, 0])
        a[0] += 1; a[1] += e - s; a[2] += ov
    print("%-28s %6s %12s %10s" % ("kernel", "calls", "avg_us", "beside_search"))
    for k, (n, d, ov) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print("%-28s %6d %12.1f %9.1f%%" % (k, n, d / n / 1e3, 100.0 * ov / max(d, 1)))
    print("# timeline of the big kernels (ms from the first dispatch): name queue start end")
    for k, s, e, q in rows:
        if e - s > 500000:
            print("  %-24s q%-6s %9.2f %9.2f" % (k, q, (s - t0) / 1e6, (e - t0) / 1e6))


if __name__ == "__main__":
    main()
