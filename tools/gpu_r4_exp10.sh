#!/bin/bash
# the batch made by the counting launch: tests, then A/B against HEAD on one box
o=gpurun_out/r4/exp10; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_e2e.py -x -q > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log; [ $rc -ne 0 ] && exit 1
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],4), d['phases_ms'])
PY
}
run() { name=$1; lib=$2; shift 2; MRI_LIB=$lib timeout -k 10 300 python bench.py --no-records-leg --no-cpu-baseline --psnr-steps 300 "$@" > $o/$name.json 2> $o/$name.err; rc=$?; [ $rc -ne 0 ] && { echo "$name rc=$rc"; tail -3 $o/$name.err; }; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1; line $o/$name.json; }
N=mri_interpolation_amd/libmri_inr.so; H=tools/libmri_head.so
run new_a $N; run head_a $H; run new_b $N; run head_b $H; run new_c $N; run head_c $H
run new_cfg5 $N --workload cfg5; run head_cfg5 $H --workload cfg5
