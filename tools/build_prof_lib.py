#!/usr/bin/env python3
"""Builds pocketsphinx_amd/libpsgpu_prof.so: the product sources with -DPSGPU_FT_PROFILE (per-phase cycle counters in the
tree-search kernel, printed to stderr after every search).  A measuring tool, never loaded by the package: run a bench
with PSGPU_LIB_PATH=pocketsphinx_amd/libpsgpu_prof.so to use it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pocketsphinx_amd import capi  # noqa: E402

print(capi.build_library(force=False, extra_flags=["-DPSGPU_FT_PROFILE"], lib_path=os.path.join(capi.PKG_DIR, "libpsgpu_prof.so"),
                         build_dir=os.path.join(capi.PKG_DIR, "_build_prof")))
