#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Stage a multi-density model for the ms scorer under
oracle/_ref/model/en-us-ms (the only bundled continuous model, an4_ci_cont, has
ONE density per codebook): the en-us mdef / means / variances with a float
mixture_weights file de-quantised from the en-us sendump,
to be used with `-senmgau .ptm.`.

usage: stage_ms_model.py EN_US_MODEL_DIR TABLES.npz OUT_DIR
"""
import os
import shutil
import struct
import sys

import numpy as np


def write_s3_mixw(path, w):
    """S3 mixture_weights file (senone_mixw_read, ms_senone.c:134-267): text header,
    byte-order magic, n_sen n_feat n_cw n_total, float32 [sen][feat][cw]."""
    w = np.ascontiguousarray(w, np.float32)
    with open(path, "wb") as fh:
        fh.write(b"s3\nversion 1.0\nendhdr\n")
        fh.write(struct.pack("<I4i", 0x11223344, w.shape[0], w.shape[1], w.shape[2], w.size))
        fh.write(w.tobytes())


def stage(src, tables, dst):
    os.makedirs(dst, exist_ok=True)
    for f in ("mdef", "means", "variances", "transition_matrices", "feat.params", "noisedict"):
        if not os.path.exists(os.path.join(dst, f)):
            shutil.copy(os.path.join(src, f), dst)
    mw = os.path.join(dst, "mixture_weights")
    if not os.path.exists(mw):
        t = np.load(tables)
        q = t["mixw"].astype(np.float64)                       # [feat][cw][sen], -log_{1.0001}(w) >> 10
        w = np.power(1.0001, -(q * 1024.0))                     # back to probabilities
        write_s3_mixw(mw, np.transpose(w, (2, 0, 1)))
    return dst


if __name__ == "__main__":
    stage(sys.argv[1], sys.argv[2], sys.argv[3])
