#!/bin/bash
# usage: tools/gpu_pmc.sh NAME "COUNTERS" bench-args...
name=$1; ctrs=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/${MRI_ROUND:-r4}/pmc_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc $ctrs --output-format csv -d $out -o $name -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --psnr-steps 0 > $out/bench.out 2> $out/bench.err
rc=$?; echo "pmc $name rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi  # a killed GPU step: no further GPU step in this call
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "mri::" not in k: continue
    m = re.search(r"(\w+_kernel\w*)", k)
    agg[m.group(1) if m else k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
find $out -name "*.csv" -size +3M -delete
