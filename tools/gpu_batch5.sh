#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 300 python tools/siren_time.py > gpurun_out/r2/chain2.out 2> gpurun_out/r2/chain2.err
echo chain rc=$?; cat gpurun_out/r2/chain2.out; tail -3 gpurun_out/r2/chain2.err
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/r2/tests3.out 2>&1
echo tests rc=$?; tail -4 gpurun_out/r2/tests3.out
