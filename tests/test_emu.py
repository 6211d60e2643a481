"""Steps the gfx950 kernel bodies (toyni_amd/csrc/ntt_kernels.hpp) on the CPU against the oracle:
every pass shape of the dispatch table, 1-, 2- and 3-pass plans, ragged batches, forward / inverse /
round trip, the interleaved (Ext, AoS) forms of the same passes, and the structured FRI fold.  CPU only; the shipped library contains none of this."""
import subprocess

import __graft_entry__ as entry


def test_kernel_bodies_match_oracle_on_cpu():
    exe = entry.build_emu()
    # sizes 2^0..2^14 (1- and 2-pass), plus 2^20 (the 1024 x 1024 headline split; 8-wide tiles at batch 1, 16-wide at
    # batch 4, 32-wide at batch 16 -- the streaming configuration) and 2^21 (3-pass)
    # sLOGxG: one transform split over G emulated ranks (slab pass / relayout / row transforms, include/toyni_hip.h 2b):
    # 2^20 over 2 (1024-point first pass, 8-wide tiles) and 2^21 over 4 (3-pass plan); 2^13, 2^14 run by default
    # pN: launches of at most 2^N 32-wide tiles run the three-step latency shapes (Pass3); the default (6) covers the single
    # transforms above, "p-1" then repeats 2^16 / 2^18 / 2^20 (and their LDE / slab forms) on the two-step shapes alone
    res = subprocess.run([exe, "14", "16", "18", "19", "20", "20x2", "20x4", "20x16", "21", "s20x2", "s21x4", "s18x4",
                          # lLOGxZ: low-degree extension with 2^Z-fold implied zero padding -- every remaining first-pass shape
                          # ((4,4) at 2^16, (5,4) at 2^18, (5,5) at 2^20) across zero fractions, incl. blow-ups > 32
                          "l16x1", "l16x4", "l16x5", "l16x8", "l18x3", "l18x5", "l18x9", "l20x1", "l20x3", "l20x4", "l20x7", "l20x10", "l24x5",
                          # eLOGxV: V Ext vectors (AoS) through the interleaved passes -- plain, coset, LDE by 32 and by 4; 2^20 x 4 vectors reaches
                          # the 32-wide 1024-point shapes, a lone vector the 16-wide ones (sizes up to 2^14 run in the default loop)
                          "e16", "e18x2", "e20", "e20x4", "e21",
                          "p-1", "16", "18", "20", "s20x2", "l16x5", "l18x3", "l20x5", "l20x10",
                          # wN: launches of >= 2^N 32-wide tiles take the 64-wide shapes of the 128/256/512-point passes; w0 = always
                          "w0", "13", "14x3", "15", "16", "17x2", "18", "19", "21", "22", "s21x4", "s18x2", "e14x3", "e16x2", "e18",
                          # Q1: n = 2^21 / 2^22 through their two-pass latency plans (2048-point three-step shapes), plain / coset / LDE
                          "w10", "p6", "Q1", "21", "22", "l21x5", "l21x3", "l22x5", "l21x10", "l22x11",
                          # (round 5) the STREAMING 2048-point shapes of the same plans (16-wide tiles, 32 elements per thread, from 2^7 32-wide
                          # tiles' worth: four transforms of 2^21, two of 2^22): closing row pass (stepped wave by wave through steps 1 and 2 --
                          # it has no barrier there), the column pass as the first pass of an LDE in every zero fraction, coset forms
                          "21x4", "b4", "l21x5", "l21x2", "b2", "l22x5", "l22x1", "l22x2", "l22x8",
                          # and the interleaved (Ext) form of the streaming closing pass: a lone vector (already 2^7 32-wide tiles' worth),
                          # plain / coset / LDE by 32 and 4 (two vectors, and zero fractions 3 and 4 of the column shape: once by hand,
                          # `emu_ntt 0 Q1 e21x2 l22x3 l22x4`; the GPU tests cover them against the oracle)
                          "e21", "Q0"],   # (once by hand: `emu_ntt 0 Q1 e22` -- the LDE of Ext vectors to 2^22 through the interleaved 2048-point column shape, 28 s)
                         capture_output=True, text=True, timeout=1800)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-1000:]
    assert "ALL OK" in res.stdout
    import re
    m = re.search(r"tiles stepped: one-step (\d+), two-step (\d+), three-step (\d+)", res.stdout)
    assert m and int(m.group(2)) > 0 and int(m.group(3)) > 0, "both the two-step and the three-step shapes must have run"


def test_kernel_bodies_are_memory_safe_under_asan_ubsan():
    # every tile / LDS / table index of every pass shape stays in bounds (global buffers are exactly n * batch words,
    # the LDS array exactly LDS_WORDS): 2^0..2^11 with ragged batches (2^11 also through the single-sweep LDS kernel in all its
    # workgroup shapes), 2^13 / 2^15 (single-sweep, 32-row tiles), 2^20 (8-wide 1024-point tiles), 2^21 (3 passes)
    exe = entry.build_emu_sanitized()
    res = subprocess.run([exe, "11", "13", "15", "16", "18", "20", "21", "s16x2", "s21x8", "l16x5", "l18x2", "l20x6", "e16", "e20", "p-1", "16", "20", "l20x6",
                          "w0", "14", "16", "18", "21", "e14x3", "e18", "w10", "p6", "Q1", "21", "l21x5", "l22x5", "Q0"],   # l22x5: both streaming 2048-point shapes
                         capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "ALL OK" in res.stdout
