#!/bin/bash
# the search kernel's phase profile ALONE (tools/search_bench.py) and BESIDE the other batch's stages (bench.py's headline loop):
# which phases of a frame grow when front end and scorer run on the same compute units
set -u
TAG=${1:-profbeside}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export PSGPU_LIB_PATH=$PWD/pocketsphinx_amd/libpsgpu_prof.so
PSGPU_BENCH_NO_PCIE=1 PSGPU_BENCH_PIPES=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/alone.json" 2> "$OUT/alone.txt"
PSGPU_BENCH_NO_PCIE=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 > "$OUT/beside.json" 2> "$OUT/beside.txt"
python - <<PY
import re
def last(fn):
    t = open(fn).read().split("fwdtree_kernel profile:")
    return "fwdtree_kernel profile:" + t[-1] if len(t) > 1 else ""
a, b = last("$OUT/alone.txt"), last("$OUT/beside.txt")
def rows(t):
    d = {}
    for l in t.splitlines():
        m = re.match(r"\s+(\d+) (.{48})\s+(\d+) cycles/frame", l)
        if m: d[int(m.group(1))] = (m.group(2).strip(), int(m.group(3)))
    return d
ra, rb = rows(a), rows(b)
print(a.splitlines()[0]); print(b.splitlines()[0])
for k in ra:
    if k in rb: print("%2d %-48s alone %7d beside %7d  %+6d" % (k, ra[k][0], ra[k][1], rb[k][1], rb[k][1] - ra[k][1]))
PY
