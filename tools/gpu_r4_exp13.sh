#!/bin/bash
# cfg3 bench line + kernel trace
o=gpurun_out/r4/exp13; mkdir -p $o
timeout -k 10 300 python bench.py --workload cfg3 --no-cpu-baseline --psnr-steps 0 > $o/bench.json 2> $o/bench.err; rc=$?; [ $rc -ne 0 ] && tail -3 $o/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4/exp13/bench.json')); print(d['ms_per_step'], d['phases_ms'])
PY
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$o/prof -o k --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3 --no-cpu-baseline --psnr-steps 0 --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$o/prof.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4/exp13/prof/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
