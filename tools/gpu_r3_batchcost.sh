#!/bin/bash
one() { timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --psnr-steps 0 --no-records-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s' % '$*', round(d['ms_per_step'], 4), d['phases_ms'])"; }
one; one --fixed-batch; one --no-prefetch; one; one --fixed-batch
