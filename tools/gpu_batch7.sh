#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 200 python tools/bench_kernels.py mlp > gpurun_out/r2/mlp_alone.out 2>&1; grep "H=128" gpurun_out/r2/mlp_alone.out
for v in "" "--no-prefetch"; do
timeout -k 10 200 python bench.py --no-cpu-baseline --psnr-steps 0 $v > gpurun_out/r2/b_x.out 2>/dev/null
sed -e 's/.*"ms_per_step": \([0-9.]*\).*"phases_ms": \({[^}]*}\).*/ms_per_step \1 phases \2/' gpurun_out/r2/b_x.out
done
