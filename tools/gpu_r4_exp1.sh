#!/bin/bash
# round 4, experiment batch 1: phase-sampling cost in the 20-step driver form; HIP stream priorities
o=gpurun_out/r4/exp1; mkdir -p $o
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], 'host', d.get('host_queue_ms_per_step'), d['phases_ms'], 'samples', d.get('phases_samples'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -x -q -s > $o/pytest_r4.log 2>&1; echo "pytest_r4 rc=$?"; tail -3 $o/pytest_r4.log
for pe in 4 7 10 32; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --phase-every $pe --no-records-leg --no-cpu-baseline > $o/drv_pe$pe.json 2> $o/drv_pe$pe.err || exit 1
  line $o/drv_pe$pe.json
done
run() { name=$1; shift; timeout -k 10 200 env "$@" python bench.py --no-records-leg --no-cpu-baseline > $o/$name.json 2> $o/$name.err; rc=$?; [ $rc -ne 0 ] && { echo "$name rc=$rc"; tail -3 $o/$name.err; }; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1; line $o/$name.json; }
run base_a X=1
run side_low MRI_SIDE_PRIORITY=1
run main_high MRI_MAIN_PRIORITY=-1
run base_b X=1
run side_low_b MRI_SIDE_PRIORITY=1
run main_high_b MRI_MAIN_PRIORITY=-1
grep -h "priority range" $o/*.err | head -2
