"""The launcher's dispatch table, walked against the oracle (tests/dispatch_matrix.py): once in this process with the default knobs,
and once per dispatch knob in a child program -- the non-temporal twins, the two-step shapes on small launches, the 64-wide
tiles and the single-sweep kernel's other workgroup shapes are what a default call picks only at sizes no test can afford to
check against a CPU oracle.  Together with the size-gated fold tests this is what lets tests/test_zz_kernel_coverage.py demand that
EVERY kernel of the shipped library has been run against the oracle."""
import os
import subprocess
import sys

import pytest

import dispatch_matrix
import oracle
from test_gpu_parity import ta  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dispatch_matrix_default_knobs(ta):
    lines = []
    n = dispatch_matrix.run_matrix(ta, oracle, "default", log=lines.append)
    assert n >= 300, (n, lines[-3:])


@pytest.mark.parametrize("profile", sorted(dispatch_matrix.PROFILE_ENV))
def test_dispatch_matrix_under_knob(ta, profile):
    # a child program (the knobs are read once per process); its launches are counted through TOYNI_LAUNCH_LOG (tests/conftest.py)
    # (tests/_hooks: the child dumps its launched-kernel list at exit for the coverage guard)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "_hooks")]),
               **dispatch_matrix.PROFILE_ENV[profile])
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dispatch_matrix.py"), profile], capture_output=True, text=True,
                         timeout=540, env=env, cwd=ROOT)
    assert res.returncode == 0 and f"MATRIX OK profile={profile}" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]
