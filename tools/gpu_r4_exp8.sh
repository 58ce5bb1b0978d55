#!/bin/bash
# SIREN chain with the image operand software-pipelined: tests, then cfg3 A/B on one box
o=gpurun_out/r4/exp8; mkdir -p $o
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "siren or cfg3 or chain" > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log; [ $rc -ne 0 ] && exit 1
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],4), d['phases_ms'])
PY
}
run() { name=$1; lib=$2; shift 2; MRI_LIB=$lib timeout -k 10 300 python bench.py --workload cfg3 --no-cpu-baseline --psnr-steps 0 "$@" > $o/$name.json 2> $o/$name.err; rc=$?; [ $rc -ne 0 ] && { echo "$name rc=$rc"; tail -3 $o/$name.err; }; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1; line $o/$name.json; }
run new_a mri_interpolation_amd/libmri_inr.so
run head_a tools/libmri_head.so
run r3_a tools/libmri_r3siren.so
run new_b mri_interpolation_amd/libmri_inr.so
run head_b tools/libmri_head.so
run r3_b tools/libmri_r3siren.so
