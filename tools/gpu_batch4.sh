#!/bin/bash
mkdir -p gpurun_out/r2/prof_cfg3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2/prof_cfg3 -o cfg3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 10 --warmup 3 --psnr-steps 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2/prof_cfg3.out 2> $GRAFT_REPO_ROOT/gpurun_out/r2/prof_cfg3.err
echo rc=$?
cd $GRAFT_REPO_ROOT
find gpurun_out/r2/prof_cfg3 -name "*kernel_stats*" | head
f=$(find gpurun_out/r2/prof_cfg3 -name "*kernel_stats.csv" | head -1)
head -20 "$f" | cut -c1-160
cat gpurun_out/r2/prof_cfg3.out | cut -c1-600
