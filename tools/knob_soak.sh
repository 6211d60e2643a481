#!/bin/bash
# The parity / fuzz / Merkle / prover / large-fold files once under each dispatch knob (a soak, not part of the regular suite; the
# regular suite covers every kernel through tests/dispatch_matrix.py's child programs).  Four tests assert WHICH kernel the default
# launcher picks and are deselected under the knob that changes exactly that pick.
FILES="tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_merkle.py tests/test_gpu_prover.py tests/test_gpu_fold_large.py tests/test_gpu_stream3.py tests/test_gpu_multi.py"
# PART=1|2: the first / second half (one gpurun call may run 1200 s; a setting takes ~200 s); unset: everything
PART=${PART:-0}
run() { echo "== $*"; env "$@" python -m pytest $FILES -m gpu -q -k "$K" 2>&1 | tail -4; }
if [ "$PART" != "2" ] && [ "$PART" != "3" ]; then
K="" run TOYNI_NT_MIN_BYTES=0
K="" run TOYNI_P3_TILES=-1
K="" run TOYNI_P3_TILES=-1 TOYNI_WIDE_TILES=0
K="" run TOYNI_LAT_TILES=-1
K="not take_the_single_sweep_kernel and not both_executors" run TOYNI_NO_LDS_KERNEL=1
fi
if [ "$PART" != "1" ] && [ "$PART" != "3" ]; then
K="" run TOYNI_LDS_MAX_LOG=15 TOYNI_LDS_MIN_ELEMS=0
K="not stream_kernel and not xs16_vs_oracle" run TOYNI_FOLD_XS16=0 TOYNI_FOLD_NT_MIN_BYTES=0
K="" run TOYNI_MERKLE_COOP_LOG=-1 TOYNI_FENCE=always
# round 5: the three-pass plan of n = 2^21 and no streaming 2048-point shapes (the tests that assert the two-pass plan / its kernels are deselected)
K="not two_pass and not through_the_two_pass and not keeps_the_three_pass and not launches_exactly" run TOYNI_S3_TILES=99
K="" run TOYNI_NT_MIN_BYTES=0 TOYNI_WIDE_TILES=0
# round 5: EVERY transform of n = 2^11 / 2^12 (lone ones, ragged batches, coset forms) through the one- / two-waves-per-transform kernels; and never
K="not one_wave_per_transform and not two_waves_per_transform and not single_sweep and not both_executors and not launches_exactly" run TOYNI_R2048_MIN_ROWS=0 TOYNI_R4096_MIN_ROWS=0
K="not one_wave_per_transform and not two_waves_per_transform" run TOYNI_R2048_MIN_ROWS=999999999 TOYNI_R4096_MIN_ROWS=999999999
fi
if [ "$PART" = "3" ]; then   # PART=3: only the two settings above (the round's last addition)
K="not one_wave_per_transform and not two_waves_per_transform and not single_sweep and not both_executors and not launches_exactly" run TOYNI_R2048_MIN_ROWS=0 TOYNI_R4096_MIN_ROWS=0
K="not one_wave_per_transform and not two_waves_per_transform" run TOYNI_R2048_MIN_ROWS=999999999 TOYNI_R4096_MIN_ROWS=999999999
fi
