#!/bin/bash
# Run ON THE GPU BOX: A/B of the three-pass splits of n = 2^21 .. 2^24 through the TOYNI_SPLIT3 experiment switch (ntt_plan.hpp:
# split_passes).  One tools/sweep.py process per candidate (the switch is read once per process); forward, 2^28 elements per launch
# sequence, median of five 10-call windows.  Output: gpurun_out/ab_split3.txt
set -u
OUT=${1:-gpurun_out/ab_split3.txt}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
run() {  # log_n a,b,c
    local line
    line=$(TOYNI_SPLIT3=$2 SWEEP_RANGE=$1:$(($1 + 1)) timeout -k 10 120 python3 tools/sweep.py 2>&1 | grep "n=2^$1") || line="n=2^$1 FAILED"
    echo "split=$2  $line" | tee -a "$OUT"
}
for rep in 1 2; do
    echo "# repetition $rep" | tee -a "$OUT"
    for s in 7,7,7 6,7,8 6,6,9 6,8,7 7,6,8 6,9,6 8,7,6 6,10,5 7,9,5 8,8,5; do run 21 $s; done
    for s in 7,7,8 6,8,8 6,7,9 6,6,10 7,8,7 8,8,6 7,10,5 8,9,5; do run 22 $s; done
    for s in 7,8,8 6,8,9 6,7,10 7,7,9 8,8,7 8,10,5 9,9,5 6,9,8; do run 23 $s; done
    for s in 8,8,8 7,8,9 6,8,10 7,7,10 6,9,9 9,10,5 8,9,7 7,9,8; do run 24 $s; done
done
