#!/usr/bin/env python3
"""tools/build_variant.py NAME [-Dflag ...] -- an A/B build of the product library that differs in psgpu_search.hip's compile flags
only: pocketsphinx_amd/libpsgpu_NAME.so (objects of the other sources are the product build's).  A measuring tool: run a bench
with PSGPU_LIB_PATH=pocketsphinx_amd/libpsgpu_NAME.so."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pocketsphinx_amd import capi  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
SRC = "psgpu_search.hip"
if extra and extra[0].startswith("--src="):          # (another source's flags: --src=psgpu_flat.hip)
    SRC, extra = extra[0][6:], extra[1:]
capi.build_library()
bdir = os.path.join(capi.PKG_DIR, "_build")
vdir = os.path.join(capi.PKG_DIR, "_build_var"); os.makedirs(vdir, exist_ok=True)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-unused-value", "-Wno-unused-result", "-fPIC",
         "-I" + os.path.join(ROOT, "include")]
src = os.path.join(capi.CSRC, SRC)
obj = os.path.join(vdir, "%s_%s.o" % (SRC.split(".")[0], name))
ff = [f for f in capi.FILE_FLAGS.get(SRC, []) if not any(e.startswith("-O") for e in extra) or not f.startswith("-O")]
subprocess.check_call([hipcc] + flags + ff + extra + ["-c", src, "-o", obj])
objs = [os.path.join(bdir, s + ".o") for s in capi.SOURCES if s != SRC] + [obj]
out = os.path.join(capi.PKG_DIR, "libpsgpu_%s.so" % name)
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
