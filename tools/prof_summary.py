#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into small text/JSON files that can
be committed under profiles/.

usage: prof_summary.py stats DB OUT.txt          kernel-trace --stats summary
       prof_summary.py pmc DB [DB...] OUT.json   per-kernel average counter values
"""
import json
import sqlite3
import sys


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --stats (durations in us)\n")
        fh.write("%-8s %-14s %-12s %-8s %s\n" % ("calls", "total_us", "avg_us", "pct", "kernel"))
        for name, calls, tot, avg, pct in rows:
            fh.write("%-8d %-14.3f %-12.3f %-8.3f %s\n" % (calls, tot, avg, pct, name[:160]))
    print(open(out).read())


def pmc(dbs, out):
    res = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
             "group by kernel_name, counter_name")
        for k, c, v, n in cur.execute(q):
            short = k.split("(")[0].replace("void ", "").split("<")[0]
            if not short.startswith("ptm_") and not short.startswith("psgpu") and not short.startswith("hmm_"):
                continue
            res.setdefault(short, {})[c] = {"avg": v, "n": n}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
