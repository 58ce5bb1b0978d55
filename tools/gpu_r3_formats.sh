#!/bin/bash
# kernel times of the three record formats on one box
o=gpurun_out/r3; mkdir -p $o
for r in 0 1 2; do
  MRI_ROUND=r3 bash tools/gpu_prof.sh rec$r --steps 40 --warmup 10 --opt bwd_records=$r --no-records-leg > $o/prof_rec$r.log 2>&1 || exit 1
  grep -E "bin_kernel|dense_and|finalize|tiny_mlp|hashgrid_fwd|adam" $o/rec${r}_kernel_stats.csv
  grep -o '"ms_per_step": [0-9.]*' $o/prof_rec$r/bench.out | head -1
done
export MRI_LIB=$GRAFT_REPO_ROOT/tools/libmri_aosload1.so
MRI_ROUND=r3 bash tools/gpu_prof.sh rec0v1 --steps 40 --warmup 10 --opt bwd_records=0 --no-records-leg > $o/prof_rec0v1.log 2>&1 || exit 1
grep -E "bin_kernel|dense_and|finalize" $o/rec0v1_kernel_stats.csv
grep -o '"ms_per_step": [0-9.]*' $o/prof_rec0v1/bench.out | head -1
unset MRI_LIB
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -s > $o/tests_r3.out 2>&1; echo "tests rc=$?"; tail -3 $o/tests_r3.out
