#!/bin/bash
# instruction-cache counters of the rows kernel
o=gpurun_out/r4/exp12; mkdir -p $o; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/$o/counters.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_INSTS_VALU\b\|SQ_INST_CYCLES_VMEM[A-Z_]*\|SQ_WAIT_ANY\|SQ_WAIT_IFETCH" $R/$o/counters.txt | sort -u | head -40
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES -d $o/pmc1 -o ic --output-format csv -- python3 tools/rows_run.py > $o/pmc1.log 2>&1; echo "rc=$?"; tail -2 $o/pmc1.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MFMA -d $o/pmc2 -o sq --output-format csv -- python3 tools/rows_run.py > $o/pmc2.log 2>&1; echo "rc=$?"; tail -2 $o/pmc2.log
find $o -name "*counter_collection.csv" | head
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/r4/exp12/pmc*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:50]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items():
        if 'siren' in k: print(f.split('/')[-1], k, dict(v))
PY
