import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


# Collection order of the GPU suite (the driver runs it with -x): the fast bit-exact kernel-level files first -- each row
# of SURVEY section 8 has its golden test there -- then the pipelines, and the slow subprocess suites of the reference-side
# binding last, so that one late failure cannot hide the row-level evidence.  Unlisted files keep their alphabetical place
# in the middle group.
_FIRST = ["test_ptm_gpu", "test_ptm_frame_gpu", "test_semi_gpu", "test_ms_gpu", "test_hmm_gpu", "test_fe_gpu", "test_feat_gpu",
          "test_lm_gpu", "test_search_gpu", "test_zz_search_layouts_gpu", "test_zz_flat_gpu", "test_random_models_gpu",
          "test_scorers_pipeline_gpu", "test_largevocab_gpu", "test_decode_pipeline_gpu", "test_batch_dist_gpu"]
_LAST = ["test_dropin_gpu", "test_zz_asan_gpu"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _FIRST:
            return (0, _FIRST.index(name))
        if name in _LAST:
            return (2, _LAST.index(name))
        return (1, 0)
    items.sort(key=rank)            # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def tables():
    import pso
    return pso.load_tables()


def run_isolated(module, func, *args, timeout=900):
    """Run tests/<module>.<func>(*args) in a child Python process and fail with its output if it does not exit cleanly.
    For GPU tests of kernels that have not run on a device yet: a memory fault there kills the process that launched
    the kernel, and it must not be the pytest process (which would take every other test's result with it)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path[:0] = [%r, %r]; import %s as m; m.%s(*%r)" % (here, os.path.dirname(here), module, func, tuple(args)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "%s.%s%r exited with %d\n%s\n%s" % (module, func, tuple(args), r.returncode, r.stdout[-1500:], r.stderr[-3000:])


def make_big_flat_trace(out_dir):
    """`ref_dump fwdflat` of the compiled reference on the large-vocabulary task (134,865 words; ~30 MB, ~17 s to make);
    None when oracle/_ref is not built"""
    import os
    import subprocess
    import sys
    import pso
    ref = pso.REF_DIR
    need = [os.path.join(ref, "ref_dump"), os.path.join(ref, "data", "big.arpa"), os.path.join(ref, "data", "cmudict-en-us.dict")]
    if not all(os.path.exists(p) for p in need):
        return None
    out = os.path.join(str(out_dir), "big_flat.psgb")
    subprocess.check_call([need[0], "fwdflat", out, os.path.join(ref, "model", "en-us"), need[1], need[2],
                           os.path.join(ref, "data", "goforward.raw"), "--", "fwdflat", "yes", "bestpath", "no"], timeout=900)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    from psgb import read_psgb
    return read_psgb(out)


@pytest.fixture(scope="session")
def big_flat_trace(tmp_path_factory):
    g = make_big_flat_trace(tmp_path_factory.mktemp("bigflat"))
    if g is None:
        pytest.fail("oracle/_ref (compiled reference + staged data) not built: run __graft_entry__.build() where /root/reference is present")
    return g
