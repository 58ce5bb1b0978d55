#!/bin/bash
# A/B of a library variant against the shipped one at cfg3: bash tools/gpu_r4_exp14.sh tools/libmri_nt.so
o=gpurun_out/r4/exp14; mkdir -p $o
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],4), d['phases_ms'])
PY
}
run() { name=$1; lib=$2; MRI_LIB=$lib timeout -k 10 300 python bench.py --workload cfg3 --no-cpu-baseline --psnr-steps 0 > $o/$name.json 2> $o/$name.err; rc=$?; [ $rc -ne 0 ] && { echo "$name rc=$rc"; tail -3 $o/$name.err; }; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1; line $o/$name.json; }
run base_a mri_interpolation_amd/libmri_inr.so
run var_a $1
run base_b mri_interpolation_amd/libmri_inr.so
run var_b $1
