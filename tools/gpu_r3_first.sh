#!/bin/bash
# round 3, first contact: new record formats + DP rehearsal tests, then the bench line and a kernel trace
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_00_dp_two_rank_gpu.py tests/test_gpu_round3.py -x -q -m gpu -s > $o/tests_first.out 2>&1; rc=$?
tail -5 $o/tests_first.out; echo "tests rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
timeout -k 10 300 python bench.py > $o/bench_cfg4.json 2> $o/bench_cfg4.err; echo "bench rc=$?"; cut -c1-3000 $o/bench_cfg4.json
