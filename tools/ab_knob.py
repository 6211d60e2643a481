#!/usr/bin/env python3
"""A/B of an environment knob over several (log_n, batch) workloads: runs bench.py (value only) alternately under each setting.
  python tools/ab_knob.py TOYNI_WIDE_TILES 10 99 -- 24:64 21:512 22:256 27:8 16:16384"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
knob = sys.argv[1]
sep = sys.argv.index("--")
values, specs = sys.argv[2:sep], sys.argv[sep + 1:]
for spec in specs:
    log_n, batch = spec.split(":")
    row = []
    for rep in range(2):
        for v in values:
            env = dict(os.environ, **{knob: v})
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-kernel-events", "--no-extras", "--no-cpu-baseline", "--steps", "5",
                                  "--warmup", "2", "--log-n", log_n, "--batch", batch], capture_output=True, text=True, env=env, timeout=300).stdout
            line = [l for l in out.splitlines() if l.startswith("{")]
            row.append((v, json.loads(line[-1])["value"] if line else float("nan")))
    print(f"n=2^{log_n} batch={batch}: " + "  ".join(f"{knob}={v}: {val / 1e9:7.1f} Gel/s" for v, val in row), flush=True)
