import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Every kernel symbol the library launches during this session, in this process and in every child process a test starts (other
    # dispatch knobs, ranks, compiled C++ hosts), ends up in this file: the library keeps an in-memory list (toyni_launched_kernels,
    # include/toyni_hip.h section 4) and writes nothing itself; Python children dump the list at exit through tests/_hooks/sitecustomize.py
    # (on PYTHONPATH below), the compiled hosts through tests/cpp/launch_dump.hpp.  tests/test_zz_kernel_coverage.py reads the file, plus
    # this process's own list, at the end of the session.
    hooks = os.path.join(ROOT, "tests", "_hooks")
    if hooks not in os.environ.get("PYTHONPATH", "").split(os.pathsep):
        os.environ["PYTHONPATH"] = hooks + (os.pathsep + os.environ["PYTHONPATH"] if os.environ.get("PYTHONPATH") else "")
    if "TOYNI_LAUNCH_LOG" not in os.environ:
        import tempfile
        fd, path = tempfile.mkstemp(prefix="toyni_launch_log_", suffix=".txt")
        os.close(fd)
        os.environ["TOYNI_LAUNCH_LOG"] = path
        config._toyni_launch_log_owned = path


def pytest_unconfigure(config):
    path = getattr(config, "_toyni_launch_log_owned", None)
    if path and os.path.exists(path):
        os.unlink(path)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "vectors.json")) as f:
        return json.load(f)
