#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py -q -m gpu -x -k "grouped or checkpoint or batchnorm" > $o/tests_c.out 2>&1; echo "tests rc=$?"; tail -5 $o/tests_c.out
one() { timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --psnr-steps 0 --no-records-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s' % '$*', round(d['ms_per_step'], 4), d['phases_ms'], d['final_loss'])"; }
one --batch-group 1; one; one --batch-group 4; one --batch-group 16; one --batch-group 1; one; one --fixed-batch
