#!/usr/bin/env python3
"""gpurun_out/profiles_<tag>_side/ (tools/collect_side_profiles.sh) -> profiles/<tag>_fold_stats.csv (rocprofv3 --stats rows of the
fold kernels + HBM bytes per launch from FETCH_SIZE / WRITE_SIZE, gfx950 fetch correction x2) and profiles/<tag>_latency_stats.csv
(per-kernel averages of the single-transform launches)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = f"gpurun_out/profiles_{tag}_side"


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1] if files else None


def stats_rows(sub, keep):
    f = newest(f"{src}/{sub}/*/*kernel_stats.csv")
    rows = []
    if f:
        for r in csv.DictReader(open(f)):
            if keep(r["Name"]):
                rows.append(r)
    return rows


def counter_mean(sub, counter, keep):
    f = newest(f"{src}/{sub}/*/*counter_collection.csv")
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and keep(r["Kernel_Name"]):
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v), max(v)) for k, v in agg.items()}


fold = lambda name: "fri_fold_kernel" in name or "fri_fold_stream_kernel" in name  # noqa: E731
rows = stats_rows("fold_stats", fold)
fetch, write = counter_mean("fold_fetch", "FETCH_SIZE", fold), counter_mean("fold_write", "WRITE_SIZE", fold)
with open(f"profiles/{tag}_fold_stats.csv", "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs", "fetch_size_kib_raw_mean", "write_size_kib_mean", "fetch_size_kib_raw_max",
                "write_size_kib_max", "hbm_bytes_largest_launch=(2*FETCH+WRITE)*1024"])
    for r in rows:
        f_, w_ = fetch.get(r["Name"]), write.get(r["Name"])
        w.writerow([r["Name"], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], f_[0] if f_ else "", w_[0] if w_ else "",
                    f_[2] if f_ else "", w_[2] if w_ else "", (2 * f_[2] + w_[2]) * 1024 if f_ and w_ else ""])
if os.path.exists(f"{src}/csrc_sha256.txt"):
    json.dump({"csrc_sha256": open(f"{src}/csrc_sha256.txt").read().strip()}, open(f"profiles/{tag}_fold_stats.meta.json", "w"))
# explicit-point fold: VALU instructions per launch of the 2^24 layer (a counter instead of a hand count)
xs = lambda name: "fri_fold_xs" in name  # noqa: E731
xs_valu, xs_waves = counter_mean("fold_xs_valu", "SQ_INSTS_VALU", xs), counter_mean("fold_xs_valu", "SQ_WAVES", xs)
if xs_valu:
    obj = {"layer_log": 24, "kernels": {k: {"SQ_INSTS_VALU": v[0], "dispatches": v[1], "SQ_WAVES": xs_waves.get(k, (None,))[0],
                                            "lane_ops_per_input_element": v[0] * 64.0 / (1 << 24)} for k, v in xs_valu.items()}}
    if os.path.exists(f"{src}/csrc_sha256.txt"):
        obj["csrc_sha256"] = open(f"{src}/csrc_sha256.txt").read().strip()
    json.dump(obj, open(f"profiles/{tag}_counters_fold_xs.json", "w"), indent=1)
    print(json.dumps(obj, indent=1))
lat = stats_rows("latency_stats", lambda name: "ntt_pass" in name)
with open(f"profiles/{tag}_latency_stats.csv", "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs"])
    for r in sorted(lat, key=lambda r: r["Name"]):
        w.writerow([r["Name"], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
print(open(f"profiles/{tag}_fold_stats.csv").read())
print(open(f"profiles/{tag}_latency_stats.csv").read()[:3000])
