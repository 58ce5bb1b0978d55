#!/bin/bash
# SQ counters of the SIREN kernels at cfg3 (one counter set per pass)
o=gpurun_out/r4/exp15; mkdir -p $o; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $o/$n -o c --output-format csv -- python3 bench.py --workload cfg3 --no-cpu-baseline --psnr-steps 0 --steps 4 --warmup 2 > $o/$n.log 2>&1; echo "$n rc=$?"
done
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for f in sorted(glob.glob('gpurun_out/r4/exp15/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        k='fwd' if 'forward_rows' in k else 'bwd' if 'backward_rows' in k else 'wgrad' if 'wgrad_rows' in k else None
        if not k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])].add(r['Dispatch_Id'])
for k,v in acc.items():
    print(k, {c: round(x/len(cnt[(k,c)])/1e6,2) for c,x in v.items()}, '(millions per launch)')
PY
