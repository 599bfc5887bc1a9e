#!/usr/bin/env python3
"""TEST / BENCHMARK INFRASTRUCTURE.  Stage a fully continuous acoustic model of en-us size under
oracle/_ref/model/en-us-cont for BASELINE configs[3] (the ms_gauden / ms_senone path with `.cont.` mixtures at scale; the
only bundled continuous model, an4_ci_cont, has 102 senones of ONE density): 5126 codebooks (one per senone) x 1 stream x
16 densities x 39 dimensions, with en-us's own mdef, transition matrices and feature parameters (minus -svspec: one stream).
Synthetic but speech-shaped: senone s takes, per stream of the en-us phonetically-tied model, the 16 codewords of its
codebook that its mixture weights like best; density k = the three streams' k-th codewords side by side, weight = the
product of their weights (renormalised).  Deterministic.

usage: stage_cont_model.py EN_US_MODEL_DIR TABLES.npz OUT_DIR"""
import os
import shutil
import struct
import sys

import numpy as np

N_DEN = 16


def read_gauden(path):
    """S3 gaussian parameter file (gauden_param_read, ms_gauden.c:112-250): text header, byte-order magic, n_mgau n_feat
    n_density, veclen[n_feat], n, float32 [mgau][feat][density][veclen]"""
    b = open(path, "rb").read()
    o = b.index(b"endhdr\n") + 7
    assert struct.unpack_from("<I", b, o)[0] == 0x11223344
    n_mgau, n_feat, n_den = struct.unpack_from("<3i", b, o + 4)
    veclen = struct.unpack_from("<%di" % n_feat, b, o + 16)
    n = struct.unpack_from("<i", b, o + 16 + 4 * n_feat)[0]
    assert len(set(veclen)) == 1 and n == n_mgau * n_den * sum(veclen)
    a = np.frombuffer(b, np.float32, n, o + 20 + 4 * n_feat)
    return a.reshape(n_mgau, n_feat, n_den, veclen[0])


def write_gauden(path, a):
    a = np.ascontiguousarray(a, np.float32)            # [mgau][feat][density][veclen]
    with open(path, "wb") as fh:
        fh.write(b"s3\nversion 1.0\nendhdr\n")
        fh.write(struct.pack("<I3i", 0x11223344, a.shape[0], a.shape[1], a.shape[2]))
        fh.write(struct.pack("<%di" % a.shape[1], *([a.shape[3]] * a.shape[1])))
        fh.write(struct.pack("<i", a.size))
        fh.write(a.tobytes())


def write_s3_mixw(path, w):
    """senone_mixw_read, ms_senone.c:134-267: n_sen n_feat n_cw n_total, float32 [sen][feat][cw]"""
    w = np.ascontiguousarray(w, np.float32)
    with open(path, "wb") as fh:
        fh.write(b"s3\nversion 1.0\nendhdr\n")
        fh.write(struct.pack("<I4i", 0x11223344, w.shape[0], w.shape[1], w.shape[2], w.size))
        fh.write(w.tobytes())


def stage(src, tables, dst):
    os.makedirs(dst, exist_ok=True)
    if all(os.path.exists(os.path.join(dst, f)) for f in ("means", "variances", "mixture_weights", "mdef", "feat.params")):
        return dst
    for f in ("mdef", "transition_matrices", "noisedict"):
        shutil.copy(os.path.join(src, f), dst)
    with open(os.path.join(dst, "feat.params"), "w") as fh:
        for ln in open(os.path.join(src, "feat.params")):
            if not ln.startswith("-svspec"):
                fh.write(ln)
    mean, var = read_gauden(os.path.join(src, "means")), read_gauden(os.path.join(src, "variances"))      # [cb][3][128][13]
    t = np.load(tables)
    q = t["mixw"].astype(np.int32)                         # [feat][cw][sen]: -log_{1.0001}(w) >> 10
    s2c = t["sen2cb"].astype(np.int64)
    n_sen = q.shape[2]
    order = np.argsort(q, axis=1, kind="stable")[:, :N_DEN, :]          # [feat][k][sen]: the k-th best codeword of each stream
    m = np.empty((n_sen, 1, N_DEN, 39), np.float32); v = np.empty_like(m)
    w = np.ones((n_sen, 1, N_DEN), np.float64)
    sen = np.arange(n_sen)
    for f in range(3):
        cw = order[f]                                       # [k][sen]
        for k in range(N_DEN):
            m[:, 0, k, 13 * f:13 * f + 13] = mean[s2c, f, cw[k], :]
            v[:, 0, k, 13 * f:13 * f + 13] = var[s2c, f, cw[k], :]
            w[:, 0, k] *= np.power(1.0001, -(q[f, cw[k], sen].astype(np.float64) * 1024.0))
    w /= w.sum(axis=2, keepdims=True)
    write_gauden(os.path.join(dst, "means"), m)
    write_gauden(os.path.join(dst, "variances"), v)
    write_s3_mixw(os.path.join(dst, "mixture_weights"), w)
    return dst


if __name__ == "__main__":
    print(stage(sys.argv[1], sys.argv[2], sys.argv[3]))
