#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 passes over the SAME command bench.py is judged on.
#   1. --kernel-trace --stats          -> per-kernel average duration (must agree with bench.py's HIP-event number)
#   2. --pmc FETCH_SIZE / WRITE_SIZE   -> HBM traffic per launch (separate passes; TCC slots do not fit both)
#   3. --pmc SQ_* / GRBM_GUI_ACTIVE    -> VALU issue, waits, effective clock
# Summaries land in gpurun_out/profiles_<tag>/ ; tools/summarize_profiles.py turns them into the files under profiles/.
set -u
TAG=${1:-r01}
BATCH=${2:-1024}
LOGN=${3:-20}
SUFFIX=""
[ "$LOGN" != "20" ] && SUFFIX="_2p$LOGN"
OUT=gpurun_out/profiles_$TAG$SUFFIX
mkdir -p $OUT
python3 tools/csrc_hash.py > $OUT/csrc_sha256.txt   # which kernels these profiles measure (bench.py checks it)
CMD="python3 bench.py --steps 3 --warmup 1 --log-n $LOGN --batch $BATCH --no-extras --no-cpu-baseline"
# the timing pass runs bench.py's DEFAULT step counts (the command the bench line is judged on): averages over 20 steps, not over a
# handful of launches that are still warming up; the counter passes replay kernels and keep the short run
STATS_CMD="python3 bench.py --log-n $LOGN --batch $BATCH --no-extras --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $STATS_CMD > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/lds -- $CMD > $OUT/lds.log 2>&1 || exit 1
grep -h '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_under_rocprof.json
echo "profiles collected in $OUT"
