#!/bin/bash
# rows-in-registers SIREN chain: tests, kernel timings, phase counters
o=gpurun_out/r4/exp10; mkdir -p $o
timeout -k 10 600 python -m pytest tests -m gpu -q -k "siren or cfg3 or chain" > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $o/pytest.log
[ $rc -eq 124 -o $rc -eq 137 ] && exit 1
timeout -k 10 300 python tools/siren_time.py 256 > $o/time.log 2>&1; rc=$?; cat $o/time.log; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1
MRI_LIB=tools/libmri_sprof.so timeout -k 10 200 python tools/rows_phases.py > $o/phases.log 2>&1; rc=$?; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1
MRI_LIB=tools/libmri_sprof.so timeout -k 10 200 python tools/rows_phases.py train >> $o/phases.log 2>&1; cat $o/phases.log
