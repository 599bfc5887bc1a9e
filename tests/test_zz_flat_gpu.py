"""GPU parity of the flat-lexicon second-pass kernel (psgpu_fwdflat_*, the ngram_fwdflat_search replacement,
SURVEY 8a row 18) against dumps of the unmodified reference (`ref_dump fwdflat`, the goldens of
tests/test_oracle_flat.py): handed the first pass's back-pointer table and the senone scores the reference's
second pass was handed frame by frame, the kernel must produce its back-pointer table (ten columns),
right-context score stack, frame marks and per-frame best score / back-pointer count -- bit for bit, including
the float-weighted language scores.  (Sorts last on purpose: written in a session that had no GPU minutes left;
the kernel source is checked on the CPU by tests/test_flat_hostsim.py.)"""
import numpy as np
import pytest

from test_flat_hostsim import check_flat, flat_rows
from test_oracle_flat import FLAT_CASES, load_flat

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", FLAT_CASES)
def test_fwdflat_kernel_matches_reference(case):
    import pocketsphinx_amd as P
    g, st, fst = load_flat(case)
    lm = P.NGramTrieLM(fst) if "lm" not in st else None
    s = P.FwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"], lm=lm)
    r = s.search(flat_rows(g, s.n_sen), [int(g["flat_n_steps"][0])], [g["bp1"]], [g["flat_w1_ssid"]])[0]
    check_flat(r, g, case)
    s.close()


def test_fwdflat_kernel_batch_of_utterances():
    import pocketsphinx_amd as P
    loaded = [load_flat(n) for n in ("goforward", "numbers")]
    g0, st, fst = loaded[0]
    s = P.FwdflatSearch(st, fst, g0["par"], g0["flat_par"], g0["flat_lwf"])
    gs = [loaded[0][0], loaded[1][0], loaded[0][0]]
    rows = [flat_rows(g, s.n_sen) for g in gs]
    out = s.search(np.concatenate(rows), [r.shape[0] for r in rows], [g["bp1"] for g in gs], [g["flat_w1_ssid"] for g in gs])
    for r, g in zip(out, gs):
        check_flat(r, g, "batch")
    s.close()


@pytest.mark.parametrize("case", ["goforward", "numbers", "man_ah_2934za"])
def test_two_passes_chained_on_the_device(case):
    """tree search -> flat search, the hand-over (back-pointer table, result record, single-phone ssids) staying in
    device buffers; the second pass must end with the reference's pass-2 tables"""
    import pocketsphinx_amd as P
    from test_oracle_golden import _load
    from test_search_gpu import _inputs
    g, st, fst = load_flat(case)
    g1 = _load("fwdtree_trace_%s.npz" % case)
    s1 = P.FwdtreeSearch(st, g1["par"])
    rows1, pen1 = _inputs(g1, s1.n_sen)
    h = {}
    r1 = s1.search(rows1, pen1, [rows1.shape[0]], handover=h)[0]
    assert np.array_equal(r1["bp"], g["bp1"]) and np.array_equal(h["w1_ssid"][0].cpu().numpy(), g["flat_w1_ssid"])
    s2 = P.FwdflatSearch(st, fst, g["par"], g["flat_par"], g["flat_lwf"])
    r2 = s2.search(flat_rows(g, s2.n_sen), [int(g["flat_n_steps"][0])], h)[0]
    check_flat(r2, g, case)
    s1.close(); s2.close()
