#!/bin/bash
o=gpurun_out/r4/exp7; mkdir -p $o
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],4), d['phases_ms'])
PY
}
run() { name=$1; lib=$2; shift 2; MRI_LIB=$lib timeout -k 10 200 python bench.py --no-records-leg --no-cpu-baseline --psnr-steps 300 "$@" > $o/$name.json 2> $o/$name.err; rc=$?; [ $rc -ne 0 ] && { echo "$name rc=$rc"; tail -3 $o/$name.err; }; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1; line $o/$name.json; }
D=mri_interpolation_amd/libmri_inr.so
run base_a $D
run pk1_a tools/libmri_pk1.so
run pk4_a tools/libmri_pk4.so
run base_b $D
run pk1_b tools/libmri_pk1.so
run pk4_b tools/libmri_pk4.so
