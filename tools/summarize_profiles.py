#!/usr/bin/env python3
"""Turn gpurun_out/profiles_<tag>/ (tools/collect_profiles.sh) into the tracked summaries under profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats table, verbatim
  profiles/<tag>_counters.json         mean per dispatch of every collected counter, per kernel
  profiles/<tag>_traffic.json          HBM bytes per launch of each NTT pass kernel:
                                       (2 * FETCH_SIZE + WRITE_SIZE) * 1024  -- FETCH_SIZE is in KiB and on gfx950 reports
                                       half of a coalesced stream (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
bench.py reads <tag>_traffic.json for roofline.traffic when its workload matches."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
log_n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
suffix = "" if log_n == 20 else f"_2p{log_n}"          # profiles/<tag>_traffic_2p24.json etc. for the other headline size
src = f"gpurun_out/profiles_{tag}{suffix}"
os.makedirs("profiles", exist_ok=True)


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]  # gpurun_out accumulates runs: keep the latest only


def counters(sub):
    files = newest(f"{src}/{sub}/*/*counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": len(next(iter(cs.values())))} for k, cs in agg.items()}


stats = newest(f"{src}/stats/*/*kernel_stats.csv")
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_kernel_stats{suffix}.csv")
allc = {}
for sub in ("fetch", "write", "sq", "lds"):
    for k, v in counters(sub).items():
        allc.setdefault(k, {}).update(v)
json.dump(allc, open(f"profiles/{tag}_counters{suffix}.json", "w"), indent=1, sort_keys=True)

durations = {}
if stats:
    for r in csv.DictReader(open(stats[0])):
        durations[r["Name"]] = float(r["AverageNs"])
traffic = {"log_n": log_n, "batch_per_gpu": batch, "command": f"python3 bench.py --steps 3 --warmup 1 --log-n {log_n} --batch {batch} --no-extras --no-cpu-baseline",
           "timing_command": f"python3 bench.py --log-n {log_n} --batch {batch} --no-extras --no-cpu-baseline (default --steps 20 --warmup 3): rocprof_avg_ns",
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch", "kernels": {}}
for k, v in allc.items():
    if ("ntt_pass_kernel" in k or "ntt_pass3s_kernel" in k) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:   # (the streaming passes; not the latency shapes of lone transforms)
        traffic["kernels"][k] = {
            "fetch_size_kib_raw": v["FETCH_SIZE"], "write_size_kib": v["WRITE_SIZE"],
            "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
            "rocprof_avg_ns": durations.get(k),
        }
json.dump(traffic, open(f"profiles/{tag}_traffic{suffix}.json", "w"), indent=1, sort_keys=True)
hf = f"{src}/csrc_sha256.txt"
if os.path.exists(hf):
    traffic["csrc_sha256"] = open(hf).read().strip()
    json.dump(traffic, open(f"profiles/{tag}_traffic{suffix}.json", "w"), indent=1, sort_keys=True)
bj = f"{src}/bench_under_rocprof.json"
if os.path.exists(bj) and os.path.getsize(bj):
    # The bench line printed INSIDE the --stats run.  Its roofline.traffic was read from whatever profile was committed when the run
    # started -- by construction the previous collection, not the one this very run produces -- so those fields (and their STALE note
    # after a source change) say nothing about this library and are dropped; the traffic of THIS library is {tag}_traffic{suffix}.json.
    line = json.loads(open(bj).read().strip().splitlines()[-1])
    for obj in [line.get("roofline")] + [v.get("roofline") for v in (line.get("extras") or {}).values() if isinstance(v, dict)]:
        if isinstance(obj, dict):
            obj.pop("traffic", None)
            obj.pop("traffic_source", None)
    line["note_on_this_file"] = "bench.py under rocprofv3 --kernel-trace --stats (profiled runs clock lower); traffic fields removed, see the traffic file of the same tag"
    if os.path.exists(hf):
        line["csrc_sha256"] = open(hf).read().strip()
    json.dump(line, open(f"profiles/{tag}_bench_under_rocprof{suffix}.json", "w"))
print(json.dumps(traffic, indent=1)[:1500])
