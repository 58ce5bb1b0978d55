#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 1000 python -m pytest tests/test_00_dp_two_rank_gpu.py tests/test_gpu_round3.py -q -m gpu -x > $o/tests_l.out 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $o/tests_l.out
