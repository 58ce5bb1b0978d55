#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 600 python tools/cfg5_sweep.py > gpurun_out/r2/cfg5_sweep.out 2> gpurun_out/r2/cfg5_sweep.err
echo sweep rc=$?; cat gpurun_out/r2/cfg5_sweep.out; tail -3 gpurun_out/r2/cfg5_sweep.err
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r2/tests5.out 2>&1
echo tests rc=$?; tail -4 gpurun_out/r2/tests5.out
