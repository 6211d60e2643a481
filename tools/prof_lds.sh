#!/bin/bash
# Run ON THE GPU BOX: counters of the single-sweep kernel (n = 2^LOG, 2^28 elements per launch).  Summaries -> gpurun_out/prof_lds/
set -u
LOG=${1:-13}
OUT=gpurun_out/prof_lds
mkdir -p $OUT
export PROF_LOG_N=$LOG PROF_BATCH=$((1 << (28 - LOG))) PROF_REPS=3 PROF_EXTRAS=0
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/prof_ntt.py > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 tools/prof_ntt.py > $OUT/sq.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/lds -- python3 tools/prof_ntt.py > $OUT/lds.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for sub in ("sq", "lds"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/prof_lds/{sub}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "ntt_" in k:
            print(k[:70], {c: round(sum(v) / len(v)) for c, v in cs.items()})
for f in glob.glob("gpurun_out/prof_lds/stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "ntt_" in r["Name"]:
            print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
