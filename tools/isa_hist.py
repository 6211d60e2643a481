#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S listing (diagnostic): tools/isa_hist.py build/toyni.s <symbol substring> [--loop]
Prints VALU / SALU / VMEM / LDS counts per mnemonic, weighted by the measured issue cost of profiles/r02_microbench.txt."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.rstrip().split(":")[0].endswith("j") or (l.startswith("_Z") and pat in l and ": " in l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
# cost model (cycles per wave-instruction per SIMD at >= 2 waves / SIMD), profiles/r02_microbench.txt
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_mov_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_add_f32", "v_mul_f32"}
hist = collections.Counter()
for l in body:
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if m:
        hist[re.sub(r"_e32$|_e64$", "", m.group(1))] += 1
tot = collections.Counter()
cyc = 0.0
for k, v in hist.items():
    cls = "VALU" if k.startswith("v_") else "SALU" if k.startswith("s_") else "LDS" if k.startswith("ds_") else "VMEM" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
    tot[cls] += v
    if cls == "VALU":
        cyc += v * (2.5 if k in FAST else 4.75 if k.startswith("v_mad_u64") else 4.0)
print(f"{lines[start][:120]}")
print("totals:", dict(tot), f"VALU issue cycles (model) {cyc:.0f}")
for k, v in hist.most_common(40):
    print(f"  {v:6d} {k}")
