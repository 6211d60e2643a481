"""Test-session hook for CHILD Python processes (tests/conftest.py puts this directory on PYTHONPATH): at exit, append the kernel
symbols this process launched through libtoyni_hip.so (toyni_launched_kernels: an in-memory list; the library itself writes no file)
to $TOYNI_LAUNCH_LOG, where tests/test_zz_kernel_coverage.py counts them.  Test infrastructure only; does nothing unless the
variable is set and toyni_amd was imported."""
import atexit
import os
import sys


def _toyni_dump_launched_kernels():
    path = os.environ.get("TOYNI_LAUNCH_LOG")
    mod = sys.modules.get("toyni_amd._lib")
    if not path or mod is None:
        return
    try:
        names = mod.launched_kernels()
        if names:
            with open(path, "a") as f:
                f.write("".join(n + "\n" for n in names))
    except Exception:   # a child that failed before the library was usable has nothing to report
        pass


atexit.register(_toyni_dump_launched_kernels)
