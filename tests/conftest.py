import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


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
