#!/bin/bash
# rows-in-registers SIREN chain: phase counters (and a timing-only build without the activation stores)
o=gpurun_out/r4/exp11; mkdir -p $o
timeout -k 10 300 python tools/siren_time.py 256 > $o/time.log 2>&1; rc=$?; head -3 $o/time.log; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1
for v in sprof sprof_ns; do
  for m in "" train; do
    echo "== $v $m" >> $o/phases.log
    MRI_LIB=tools/libmri_$v.so timeout -k 10 200 python tools/rows_phases.py $m >> $o/phases.log 2>&1; rc=$?; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1
  done
done
grep -v amdgpu.ids $o/phases.log
