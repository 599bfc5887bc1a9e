"""Per-call latency of psgpu_ptm_frame_eval (fresh = top-N + senone kernels, reuse = senone only)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pso, pocketsphinx_amd as P
t = pso.load_tables()
g = np.load(os.path.join(pso.GOLDEN_DIR, "senlog_default.npz"))
m = P.PtmModel(t); st = P.PtmState(m, 7)
off = g["call_act_off"]
calls = []
for c in range(600):
    na = int(g["call_nact"][c])
    calls.append((g["call_feat"][c], int(g["call_frame"][c]), None if na < 0 else g["call_act"][off[c]:off[c]+na], int(g["call_frame_idx"][c])))
for rep in range(2):
    tf = tr = 0.0; nf = nr = 0
    for feat, fr, act, fi in calls:
        t0 = time.perf_counter()
        st.frame_eval(feat, fr, active=act, compallsen=False, frame_idx=fi)
        dt = time.perf_counter() - t0
        if fr >= fi: tf += dt; nf += 1
        else: tr += dt; nr += 1
print("fresh calls: %d, %.1f us each; reuse calls: %d, %.1f us each (incl. ~3 us of ctypes)" % (nf, 1e6*tf/nf, nr, 1e6*tr/max(nr,1)))
