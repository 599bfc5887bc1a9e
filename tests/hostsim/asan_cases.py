"""Driven by tools/hostsim_asan.sh: search kernels on the simulator, built with AddressSanitizer, output buffers sized
to what each golden needs (+ the kernels' documented head-room)."""
import os

import simlib
simlib.LIB = os.environ["PSGPU_SIM_LIB"]; simlib.DEPS = []

import pso  # noqa: E402
from test_flat_hostsim import check_flat, flat_rows  # noqa: E402
from test_oracle_flat import load_flat  # noqa: E402
from test_oracle_golden import _load  # noqa: E402
from test_search_gpu import _check, _inputs  # noqa: E402


def caps(g):
    return dict(bp_cap=int(g["bp"].shape[0]) + 64, bss_cap=int(g["bscore_stack"].shape[0]) + 128)


for case, mode in [("goforward", "slab"), ("goforward", "lds"), ("goforward_maxhmmpf60_maxwpf3", "lds"), ("man_ah_2934za", "lds"), ("medium_numbers_maxwpf8", "lds"),
                   ("goforward_maxhmmpf60_maxwpf3", "slab"), ("man_ah_2934za", "slab"), ("medium_numbers_maxwpf8", "slab")]:
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    os.environ["PSGPU_FWDTREE_LAYOUT"] = mode
    s = simlib.SimFwdtreeSearch(st, g["par"], lm=simlib.SimLm(st) if "lm" not in st else None)
    rows, pen = _inputs(g, s.n_sen)
    _check(s.search(rows, pen, [rows.shape[0]], handover={}, **caps(g))[0], g, case)
    print("tree search", case, "mode", mode, "clean")
# the search resumed between calls (the state saved and restored), and on a window of rows (frames already searched dropped: a
# read of one is out of bounds here)
for case, mode in [("goforward", "lds"), ("goforward", "slab"), ("man_ah_2934za", "lds")]:
    g = _load("fwdtree_trace_%s.npz" % case)
    st = _load("fwdtree_static_%s.npz" % bytes(g["static"]).decode())
    os.environ["PSGPU_FWDTREE_LAYOUT"] = mode
    s = simlib.SimFwdtreeSearch(st, g["par"])
    rows, pen = _inputs(g, s.n_sen)
    T = rows.shape[0]
    _check(s.search(rows, pen, [T], cuts=[7, 8, T // 2], lag=4, **caps(g))[0], g, case)
    r, _ = simlib.search_windows(s, rows, pen, [9, 10, T // 2, T], 4, **caps(g))
    _check(r, g, case)
    print("tree search resumed / on windows", case, "mode", mode, "clean")
# two streams on one handle, windows back to back in one buffer, a restart in between
ga, gb = _load("fwdtree_trace_goforward.npz"), _load("fwdtree_trace_numbers.npz")
st = _load("fwdtree_static_en_us_turtle.npz")
for mode in ("lds", "slab"):
    os.environ["PSGPU_FWDTREE_LAYOUT"] = mode
    s = simlib.SimFwdtreeSearch(st, ga["par"])
    a, b = _inputs(ga, s.n_sen), _inputs(gb, s.n_sen)
    Ta, Tb = a[0].shape[0], b[0].shape[0]
    sched = [[(50, False, None), (20, False, None)], [(Ta, True, None), (131, False, None)], [(70, False, b), (200, False, None)],
             [(Tb, True, None), (Tb, True, None)]]
    out, _ = simlib.search_streams(s, [a, b], sched, 6, bp_cap=int(gb["bp"].shape[0]) + 64, bss_cap=int(gb["bscore_stack"].shape[0]) + 128)
    _check(out[0], gb, "stream 0"); _check(out[1], gb, "stream 1")
    print("tree search, two streams with a restart, mode", mode, "clean")
for case in ["goforward", "man_ah_2934za", "medium_numbers"]:
    g, st, fst = load_flat(case)
    s = simlib.SimFwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"], lm=simlib.SimLm(fst) if "lm" not in st else None)
    check_flat(s.search(flat_rows(g, s.n_sen), [int(g["flat_n_steps"][0])], [g["bp1"]], [g["flat_w1_ssid"]], **caps(g))[0], g, case)
    print("flat search", case, "clean")
g, st, fst = load_flat("numbers")
s = simlib.SimFwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"])
check_flat(s.search(g["flat_feat"], [g["flat_feat"].shape[0]], [g["bp1"]], [g["flat_w1_ssid"]], ptm_tables=pso.load_tables(),
                    topn_seed=g["flat_ptm_seed"], **caps(g))[0], g, "numbers")
print("flat search scoring its own senones clean")

# the second pass with the batch scorer's lists at hand (lazy lists, open entries by a wavefront) and with its LDS queues cut
# down to almost nothing (every frame through the slab fall-backs)
import ctypes  # noqa: E402
from test_flat_hostsim import _all_density_lists  # noqa: E402
t = pso.load_tables()
g, st, fst = load_flat("goforward")
lists = _all_density_lists(t, g["flat_feat"])
lists = (lists[0], lists[1], lists[2] | (__import__("numpy").random.default_rng(3).random(lists[2].shape) < 0.3).astype("uint8"))
s = simlib.SimFwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"])
check_flat(s.search(g["flat_feat"], [g["flat_feat"].shape[0]], [g["bp1"]], [g["flat_w1_ssid"]], ptm_tables=t, topn_seed=g["flat_ptm_seed"],
                    lists=lists, **caps(g))[0], g, "goforward, lists")
print("flat search taking the batch scorer's lists clean")
for name, val in (("psgpu_sim_ff_exit_cap", 1), ("psgpu_sim_ff_el_cap", 3), ("psgpu_sim_ff_awl_regs", 0), ("psgpu_sim_ff_pair_rows", 1), ("psgpu_sim_ff_slice_chunk", 2), ("psgpu_sim_ff_force_walk", 1)):
    knob = ctypes.c_int.in_dll(simlib.lib(), name)
    old = knob.value
    knob.value = val
    check_flat(s.search(g["flat_feat"], [g["flat_feat"].shape[0]], [g["bp1"]], [g["flat_w1_ssid"]], ptm_tables=t, topn_seed=g["flat_ptm_seed"],
                        lists=lists, **caps(g))[0], g, "goforward, lists, %s" % name)
    knob.value = old
    print("flat search with", name, "=", val, "clean")

# tables too small: the kernels must stop with status 1 and stay inside the buffers
g = _load("fwdtree_trace_numbers.npz")
st = _load("fwdtree_static_en_us_turtle.npz")
for mode in ("slab", "lds"):
    os.environ["PSGPU_FWDTREE_LAYOUT"] = mode
    s = simlib.SimFwdtreeSearch(st, g["par"])
    rows, pen = _inputs(g, s.n_sen)
    for kw in (dict(bp_cap=300, bss_cap=1 << 16), dict(bp_cap=4096, bss_cap=900)):
        r = s.search(rows, pen, [rows.shape[0]], **kw)[0]
        assert r["status"] == 1 and r["n_frame"] < rows.shape[0], (mode, kw, r["status"], r["n_frame"])
    print("tree search mode", mode, "full tables: status 1, clean")
g, st, fst = load_flat("numbers")
s = simlib.SimFwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"])
for kw in (dict(bp_cap=300, bss_cap=1 << 16), dict(bp_cap=4096, bss_cap=900)):
    r = s.search(flat_rows(g, s.n_sen), [int(g["flat_n_steps"][0])], [g["bp1"]], [g["flat_w1_ssid"]], **kw)[0]
    assert r["status"] == 1 and r["n_frame"] < int(g["flat_n_steps"][0]), (kw, r["status"], r["n_frame"])
print("flat search full tables: status 1, clean")
