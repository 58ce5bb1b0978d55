#!/bin/bash
# measurement batch of a round: bench lines of every workload, kernel stats, PMC traffic
#   MRI_ROUND=r3 bash tools/gpu_final.sh [part ...]     parts: bench prof pmc (default: all)
export MRI_ROUND=${MRI_ROUND:-r4}
o=gpurun_out/$MRI_ROUND/final; mkdir -p $o
parts=${@:-bench prof pmc}   # (pmc2: only the FETCH / WRITE passes of cfg2 and cfg5)
run() { local name=$1 limit=$2; shift 2; timeout -k 10 "$limit" "$@" > "$o/$name.json" 2> "$o/$name.err"; local rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; }
if [[ $parts == *bench* ]]; then
run bench_cfg4 600 python bench.py
run bench_cfg2 300 python bench.py --workload cfg2
run bench_cfg3 300 python bench.py --workload cfg3
run bench_cfg5 300 python bench.py --workload cfg5
run predict_cfg4 200 python bench.py --mode predict --steps 64
run predict_cfg3 200 python bench.py --mode predict --workload cfg3 --steps 32
run predict_cfg5 200 python bench.py --mode predict --workload cfg5 --steps 40
fi
if [[ $parts == *prof* ]]; then
for w in cfg4 cfg3 cfg5; do bash tools/gpu_prof.sh $w --workload $w --steps 20 --warmup 5 --no-records-leg > $o/prof_$w.log 2>&1 || exit 1; cp gpurun_out/$MRI_ROUND/${w}_kernel_stats.csv $o/; done
bash tools/gpu_prof.sh cfg4_packed --steps 20 --warmup 5 --records packed --no-records-leg > $o/prof_cfg4_packed.log 2>&1 || exit 1; cp gpurun_out/$MRI_ROUND/cfg4_packed_kernel_stats.csv $o/
bash tools/gpu_prof.sh cfg4_predict --mode predict --steps 40 --warmup 5 > $o/prof_cfg4_predict.log 2>&1 || exit 1; cp gpurun_out/$MRI_ROUND/cfg4_predict_kernel_stats.csv $o/
fi
if [[ " $parts " == *" pmc "* ]]; then
for w in cfg4 cfg3 cfg2 cfg5; do
  bash tools/gpu_pmc.sh ${w}_fetch "FETCH_SIZE" --workload $w --steps 6 --warmup 2 --no-records-leg > $o/pmc_${w}_fetch.log 2>&1 || exit 1
  bash tools/gpu_pmc.sh ${w}_write "WRITE_SIZE" --workload $w --steps 6 --warmup 2 --no-records-leg > $o/pmc_${w}_write.log 2>&1 || exit 1
done
bash tools/gpu_pmc.sh cfg4_packed_fetch "FETCH_SIZE" --steps 6 --warmup 2 --records packed --no-records-leg > $o/pmc_cfg4_packed_fetch.log 2>&1 || exit 1
bash tools/gpu_pmc.sh cfg4_packed_write "WRITE_SIZE" --steps 6 --warmup 2 --records packed --no-records-leg > $o/pmc_cfg4_packed_write.log 2>&1 || exit 1
for w in cfg4 cfg3; do
  bash tools/gpu_pmc.sh ${w}_sq "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" --workload $w --steps 6 --warmup 2 --no-records-leg > $o/pmc_${w}_sq.log 2>&1 || exit 1
done
bash tools/gpu_pmc.sh cfg4_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" --steps 6 --warmup 2 --no-records-leg > $o/pmc_cfg4_lds.log 2>&1 || exit 1
grep -h "^siren\|^tiny\|^hash\|^bin\|^dense\|^adam" $o/pmc_*.log | cut -c1-300
fi
if [[ $parts == *pmc2* ]]; then
for w in cfg2 cfg5; do
  bash tools/gpu_pmc.sh ${w}_fetch "FETCH_SIZE" --workload $w --steps 6 --warmup 2 --no-records-leg > $o/pmc_${w}_fetch.log 2>&1 || exit 1
  bash tools/gpu_pmc.sh ${w}_write "WRITE_SIZE" --workload $w --steps 6 --warmup 2 --no-records-leg > $o/pmc_${w}_write.log 2>&1 || exit 1
done
fi
if [ -f $o/bench_cfg4.json ]; then cut -c1-1500 $o/bench_cfg4.json; fi
