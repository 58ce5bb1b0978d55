#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py -q -m gpu -x -k "graphed" > $o/tests_g.out 2>&1; rc=$?; echo "tests rc=$rc"; tail -25 $o/tests_g.out
