#!/bin/bash
# f32-record table gradient: the speed-only knobs, re-swept on one box (they were tuned under packed records)
o=gpurun_out/r3; mkdir -p $o; : > $o/sweep.log
one() { timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --psnr-steps 0 --no-records-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-60s' % '$*', round(d['ms_per_step'], 4), d['phases_ms']['hashgrid_bwd'])" >> $o/sweep.log; }
one
for v in 32 48 96 128; do one --opt bwd_blocks_per_level=$v; done
for v in 1 2 8 16; do one --opt bwd_dense_max_parts=$v; done
for v in 48 64 128 192; do one --opt bwd_dense_blocks=$v; done
one --opt bwd_fuse_dense=0
one --no-count-ahead
one
cat $o/sweep.log
