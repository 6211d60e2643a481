"""Kernel-coverage guard: every __global__ symbol of the SHIPPED libtoyni_hip.so must have been launched by the GPU
test session that is just finishing -- so a kernel that the launcher picks only beyond some size, alignment or knob cannot ship
without having been through a parity test.  (This file sorts last on purpose: pytest runs files in name order.)

How: the library files the symbol of every kernel at the first launch of each launch site in an in-memory list
(toyni_launched_kernels; it writes no file itself).  Child processes of the session -- other dispatch knobs, ranks, the compiled
C++ hosts -- dump their list at exit into $TOYNI_LAUNCH_LOG through the test harness (tests/_hooks/sitecustomize.py, which
tests/conftest.py puts on PYTHONPATH; tests/cpp/launch_dump.hpp), so they count too.  The kernels the binary contains are the `Function Name:` remarks of the build that produced it
(toyni_amd/lib/libtoyni_hip.resources.txt, written by __graft_entry__.build_hip).

ALLOWED_UNLAUNCHED would list instantiations that exist without a path to them, each with its reason; it is empty."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (regex on the mangled symbol, reason).  EMPTY since round 4: the no-prefetch twins that used to be listed here were deleted
# (profiles/r04_ab_notwins.txt), as were the A/B-only fold shapes and the unreachable non-temporal twins of the small row shapes.
ALLOWED_UNLAUNCHED = []


def shipped_kernels():
    import __graft_entry__ as entry
    entry.build_hip()
    remarks = open(entry.RESOURCES).read()
    return sorted(set(re.findall(r"Function Name: (\S+)", remarks)))


def test_kernel_list_of_the_build_is_present_and_plausible():
    names = shipped_kernels()
    assert len(names) >= 180 and any("fri_fold_stream_kernel" in n for n in names) and any("ntt_pass3_kernel" in n for n in names)
    for pat, _ in ALLOWED_UNLAUNCHED:
        assert any(re.search(pat, n) for n in names), f"allowlist entry matches nothing any more: {pat}"


@pytest.mark.gpu
def test_every_shipped_kernel_was_launched_by_this_session(request):
    # judged only when the whole GPU suite ran in this session (a -k / single-file run proves nothing about coverage)
    ran = {os.path.basename(str(item.fspath)) for item in request.session.items}
    gpu_files = set()
    for f in glob.glob(os.path.join(ROOT, "tests", "test_*.py")):
        src = open(f).read()
        if re.search(r"^pytestmark = pytest\.mark\.gpu|^@pytest\.mark\.gpu|^@gpu$", src, flags=re.M):
            gpu_files.add(os.path.basename(f))
    missing_files = sorted(gpu_files - ran)
    if missing_files:
        pytest.skip(f"partial session (no tests from {missing_files}): coverage is judged on full `-m gpu` runs")
    log = os.environ.get("TOYNI_LAUNCH_LOG")
    assert log and os.path.exists(log), "tests/conftest.py sets TOYNI_LAUNCH_LOG for the session"
    launched = {l.strip() for l in open(log) if l.strip()}
    from toyni_amd import _lib
    launched |= set(_lib.launched_kernels())
    never = [k for k in shipped_kernels() if k not in launched and not any(re.search(p, k) for p, _ in ALLOWED_UNLAUNCHED)]
    assert not never, "kernels of the shipped library that no GPU test launched:\n  " + "\n  ".join(never)
