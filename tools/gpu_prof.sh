#!/bin/bash
# usage: tools/gpu_prof.sh NAME bench-args...   -> gpurun_out/${MRI_ROUND:-r4}/prof_NAME/, summary in gpurun_out/${MRI_ROUND:-r4}/NAME_kernel_stats.csv
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/${MRI_ROUND:-r4}/prof_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $name -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --psnr-steps 0 > $out/bench.out 2> $out/bench.err
rc=$?; echo "prof $name rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi  # a killed GPU step: no further GPU step in this call
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_prof.py stats "$f" gpurun_out/${MRI_ROUND:-r4}/${name}_kernel_stats.csv && cat gpurun_out/${MRI_ROUND:-r4}/${name}_kernel_stats.csv
find $out -name "*.csv" -size +2M -delete
