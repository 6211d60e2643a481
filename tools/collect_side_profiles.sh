#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 evidence for the kernels that are not the headline pass kernels.
#   fold     : the structured fold on a 2^27 layer (one size: both stream sizes are the same kernel symbol since round 4) (tools/foldbench.py): --kernel-trace --stats, then FETCH_SIZE / WRITE_SIZE
#   latency  : the single-transform kernels (three-step shapes), n = 2^16 .. 2^22 at batch 1: --kernel-trace --stats
# Output: gpurun_out/profiles_<tag>_side/ ; tools/summarize_side_profiles.py turns it into profiles/<tag>_fold_stats.csv etc.
set -u
TAG=${1:-r02}
OUT=gpurun_out/profiles_${TAG}_side
mkdir -p $OUT
python3 tools/csrc_hash.py > $OUT/csrc_sha256.txt   # which kernels these profiles measure (bench.py checks it)
FOLD_LOGS=27 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fold_stats -- python3 tools/foldbench.py > $OUT/fold_stats.log 2>&1 || exit 1
FOLD_LOGS=27 timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fold_fetch -- python3 tools/foldbench.py > $OUT/fold_fetch.log 2>&1 || exit 1
FOLD_LOGS=27 timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/fold_write -- python3 tools/foldbench.py > $OUT/fold_write.log 2>&1 || exit 1
# explicit-point fold on a 2^24 layer (the size bench.py quotes): VALU instructions per launch, counters in their own pass
XS_LOGS=24 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/fold_xs_valu -- python3 tools/foldxsbench.py > $OUT/fold_xs_valu.log 2>&1 || exit 1
LAT_RANGE=16:23 LAT_BATCHES=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/latency_stats -- python3 tools/latency.py > $OUT/latency_stats.log 2>&1 || exit 1
echo "side profiles collected in $OUT"
