#!/usr/bin/env python3
"""Debug aid (GPU box): one synthetic utterance through the device pipeline against a `ref_dump fwdtree` trace of the
compiled reference on the same PCM: phone-loop penalties, the scores of each frame's active senones, per-frame best
scores / back-pointer counts -- prints the first frame where each diverges."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))


class _DevArray:
    def __init__(self, ptr, shape, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def main():
    import torch
    import pocketsphinx_amd as P
    from pocketsphinx_amd import synth
    from psgb import read_psgb
    from test_oracle_golden import _load
    uid = int(os.environ.get("DBG_UTT", "0")); seconds = float(os.environ.get("DBG_SECONDS", "30"))
    pcm = synth.utterance(uid, seconds)
    ref = os.path.join(ROOT, "oracle", "_ref")
    tmp = tempfile.mkdtemp()
    pcm.tofile(os.path.join(tmp, "u.raw"))
    subprocess.check_call([os.path.join(ref, "ref_dump"), "fwdtree", os.path.join(tmp, "u.psgb"), os.path.join(ref, "model", "en-us"),
                           os.path.join(ref, "data", "turtle.lm.bin"), os.path.join(ref, "data", "turtle.dic"), os.path.join(tmp, "u.raw"),
                           "--", "fwdflat", "no", "bestpath", "no"])
    g = read_psgb(os.path.join(tmp, "u.psgb"))
    T = int(g["n_frame"][0])
    gt = _load("fwdtree_trace_goforward.npz")
    import pso
    p = P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), pso.load_tables(), _load("fwdtree_static_en_us_turtle.npz"), gt["par"], gt)
    reps = int(os.environ.get("DBG_REPS", "1"))
    dev = torch.device("cuda", 0)
    n_sen, n_ci, plw = 5126, int(g["pl_par"][0]), int(g["pl_par"][5])
    first = None
    for rep in range(reps):          # nondeterminism hunt: the same utterance several times, everything compared with the first run
        p.run([pcm] * int(os.environ.get("DBG_COPIES", "1")))
        hn, hyp, res = p.fetch()
        v = p.view()
        Tn = int(v.total_frames)
        pen_r = torch.as_tensor(_DevArray(v.penalties_dev, (Tn, n_ci)), device=dev).cpu().numpy()
        rows_r = torch.as_tensor(_DevArray(v.rows_dev, (Tn, n_sen), "<i2"), device=dev).cpu().numpy()
        feat_r = torch.as_tensor(_DevArray(v.feat_dev, (Tn, 39), "<f4"), device=dev).cpu().numpy()
        if first is None:
            first = (pen_r, rows_r, feat_r, res.copy())
        else:
            print("run %d: n_bp %s  feat equal %s  rows equal %s (%d rows differ)  penalties equal %s (%d frames differ, first %s)" % (
                rep, res[:, 0], np.array_equal(feat_r, first[2]), np.array_equal(rows_r, first[1]),
                int((rows_r != first[1]).any(axis=1).sum()), np.array_equal(pen_r, first[0]),
                int((pen_r != first[0]).any(axis=1).sum()), np.nonzero((pen_r != first[0]).any(axis=1))[0][:3]))
    p.run([pcm])
    hn, hyp, res = p.fetch()
    v = p.view()
    print("frames", T, "device", res[0, 2], "bp ref", g["bp"].shape[0], "device", res[0, 0], "bss ref", g["bscore_stack"].shape[0], "device", res[0, 1])
    pen = torch.as_tensor(_DevArray(v.penalties_dev, (T, n_ci)), device=dev).cpu().numpy()
    rows = torch.as_tensor(_DevArray(v.rows_dev, (T, n_sen), "<i2"), device=dev).cpu().numpy()
    step = torch.as_tensor(_DevArray(v.step_dev, (T, 4)), device=dev).cpu().numpy()
    # what the phone loop saw: the un-normalised scores of the CI senones (for an offline replay of the reference's loop)
    ci_sen = np.unique(np.asarray(_load("fwdtree_static_en_us_turtle.npz")["sseq"])[np.asarray(g["pl_ssid"])].reshape(-1))
    out = os.path.join(ROOT, "gpurun_out", "debug_long_utt%d.npz" % uid)
    np.savez_compressed(out, ci_sen=ci_sen, ci_raw=rows[:, ci_sen], pen_dev=pen, step_pen=g["step_pen"][:T], pl_par=g["pl_par"],
                        pl_weight=g["pl_weight"], pl_ssid=g["pl_ssid"], pl_tmat=g["pl_tmat"])
    print("wrote", out)
    got_pen = pen[np.minimum(np.arange(T) + plw, T - 1)]
    bad = np.nonzero((got_pen != g["step_pen"][:T]).any(axis=1))[0]
    print("penalties: %d frames differ; first %s" % (bad.size, bad[:5]))
    if bad.size:
        f = bad[0]
        print("  frame", f, "device", got_pen[f][:12], "ref", g["step_pen"][f][:12])
    off, act, scr = g["step_act_off"], g["step_act"], g["step_scr"]
    nbad = 0
    for f in range(T):
        a = act[off[f]:off[f + 1]]
        if a.size == 0:
            continue
        mine = rows[f, a].astype(np.int32)
        mine = (mine - mine.min()).astype(np.int16)
        if not np.array_equal(mine, scr[off[f]:off[f + 1]]):
            if nbad < 3:
                d = np.nonzero(mine != scr[off[f]:off[f + 1]])[0]
                print("  scores differ at frame", f, "n_active", a.size, "first senones", a[d[:5]], mine[d[:5]], scr[off[f]:off[f + 1]][d[:5]])
            nbad += 1
    print("active-senone scores: %d frames differ" % nbad)
    refstep = np.stack([g["step_best"], g["step_lpbest"], g["step_bpidx"]], axis=1)[:T]
    bad = np.nonzero((step[:T, :3] != refstep).any(axis=1))[0]
    print("per-frame (best, lpbest, bpidx): %d frames differ; first %s" % (bad.size, bad[:5]))
    if bad.size:
        f = bad[0]
        print("  frame", f, "device", step[f], "ref", refstep[f], " previous frame device", step[f - 1], "ref", refstep[f - 1])


if __name__ == "__main__":
    main()
