#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r2/tests7.out 2>&1
echo tests rc=$?; tail -4 gpurun_out/r2/tests7.out
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/r2/smoke.out 2>&1; echo smoke rc=$?; tail -2 gpurun_out/r2/smoke.out
