"""Register-budget guard for the hand-written kernels (CPU only: hipcc cross-compiles gfx950 and reports per-kernel resources).

The 1024-point pass kernels live at the edge of the 128-VGPR budget that four waves per SIMD allow; a spill (scratch) there
costs ~20 % of the headline and nothing functional fails when it happens -- it once appeared merely because a second template
instantiation of the same pass shape was removed (the inliner then makes different choices).  So the budget is a test."""
import os
import re

import __graft_entry__ as entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_spills_and_occupancy_targets_hold():
    # the remarks of the build that produced the shipped library (written by build_hip next to it); a stale or missing record
    # means the library is rebuilt here, so the test always judges the binary that the other tests load
    entry.build_hip()
    if not os.path.exists(entry.RESOURCES) or os.path.getmtime(entry.RESOURCES) < os.path.getmtime(entry.LIB) - 5:
        entry.build_hip(force=True)
    remarks = open(entry.RESOURCES).read()
    assert "loop not unrolled" not in remarks             # a stage loop that stays rolled indexes registers dynamically -> scratch
    blocks = re.split(r"remark: [^\n]*Function Name: ", remarks)[1:]
    seen = 0
    for b in blocks:
        name = b.split(" ")[0]
        if not any(k in name for k in ("ntt_pass_kernel", "ntt_lds_kernel", "ntt_pass3_kernel", "ntt_pass3s_kernel", "ntt_row2048_kernel", "ntt_row4096_kernel")):
            continue

        def field(key):
            m = re.search(key + r": (\d+)", b)
            assert m, (name, key)
            return int(m.group(1))

        seen += 1
        allowed = 0   # (rounds 1-4 tolerated 20 B of scratch in the 8-row LDS-kernel shapes of n = 2^11 / 2^12: those shapes are gone)
        assert field(r"ScratchSize \[bytes/lane\]") <= allowed, f"{name} spills {field(r'ScratchSize .bytes/lane.')} bytes per lane"
        assert field("VGPRs") <= 128, name
        assert field(r"Occupancy \[waves/SIMD\]") >= 4, name     # 16 waves per CU: what the pipelined kernels are tuned for
    assert seen >= 90
