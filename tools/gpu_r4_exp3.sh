#!/bin/bash
# where does a sampled step stall the host?  per-step host times of the 20-step driver form
o=gpurun_out/r4/exp3; mkdir -p $o
for pe in 4 6 32; do
  MRI_STEP_TIMES=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --phase-every $pe --no-cpu-baseline --no-records-leg > $o/drv_pe$pe.json 2> $o/drv_pe$pe.err || exit 1
  python -c "import json;d=json.load(open('$o/drv_pe$pe.json'));print('pe$pe',d['ms_per_step'],d['phases_samples'])"; grep "host ms" $o/drv_pe$pe.err
done
MRI_STEP_TIMES=1 timeout -k 10 200 python bench.py --steps 40 --warmup 10 --psnr-steps 0 --no-cpu-baseline --no-records-leg > $o/cold40.json 2> $o/cold40.err || exit 1
python -c "import json;d=json.load(open('$o/cold40.json'));print('cold40',d['ms_per_step'],d['phases_samples'])"; grep "host ms" $o/cold40.err
