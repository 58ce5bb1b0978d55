#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 1100 python -m pytest tests -q -m gpu "$@" > $o/tests_all.out 2>&1; rc=$?
tail -15 $o/tests_all.out; echo "tests rc=$rc"
