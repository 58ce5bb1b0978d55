#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py -q -m gpu -x -k "steady" > $o/tests_i.out 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $o/tests_i.out
[ $rc -ne 0 ] && exit 1
one() { timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --psnr-steps 0 --no-records-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % '$*', round(d['ms_per_step'], 4), 'host', d['host_queue_ms_per_step'], d['phases_ms'], d['final_loss'])"; }
one; one --launch eager; one --launch graph; one --phase-every 1000000; one --launch eager --phase-every 1000000; one --workload cfg2; one --workload cfg2 --launch eager; one --workload cfg5; one --workload cfg5 --launch eager
