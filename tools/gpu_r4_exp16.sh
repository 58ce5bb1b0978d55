#!/bin/bash
# A/B of several library variants at cfg3 (each twice, alternating): bash tools/gpu_r4_exp16.sh lib1 lib2 ...
o=gpurun_out/r4/exp16; mkdir -p $o
for rep in a b; do for lib in mri_interpolation_amd/libmri_inr.so "$@"; do
  n=$(basename $lib .so)_$rep
  MRI_LIB=$lib timeout -k 10 300 python bench.py --workload cfg3 --no-cpu-baseline --psnr-steps 0 > $o/$n.json 2> $o/$n.err; rc=$?; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1
  python - $o/$n.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],4), d['phases_ms'])
PY
done; done
