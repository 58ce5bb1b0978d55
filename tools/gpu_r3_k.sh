#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py -q -m gpu -x -k "steady" > $o/tests_k.out 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $o/tests_k.out
[ $rc -ne 0 ] && exit 1
one() { timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --psnr-steps 0 --no-records-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % '$*', round(d['ms_per_step'], 4), 'host', d['host_queue_ms_per_step'], d['phases_ms'], d['launch'][:12])"; }
one; one --launch eager; one --workload cfg2; one --workload cfg2 --launch eager; one --workload cfg2; one --workload cfg2 --launch eager;  one --workload cfg5; one --workload cfg5 --launch eager; one; one --launch eager
for i in 1 2 3; do timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --psnr-steps 0 --no-records-leg | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver-like 20/5', round(d['ms_per_step'], 4), d['phases_ms'])"; done
