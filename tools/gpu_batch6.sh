#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "overlapped" > gpurun_out/r2/tests4.out 2>&1
rc=$?; echo tests rc=$rc; tail -5 gpurun_out/r2/tests4.out
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --psnr-steps 0 > gpurun_out/r2/b_ov.out 2> gpurun_out/r2/b_ov.err; echo rc=$?
cut -c1-900 gpurun_out/r2/b_ov.out; tail -3 gpurun_out/r2/b_ov.err
