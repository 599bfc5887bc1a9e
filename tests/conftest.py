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
