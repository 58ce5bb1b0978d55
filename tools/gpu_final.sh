#!/bin/bash
# round-2 measurement batch: bench lines of every workload, kernel stats, PMC traffic
mkdir -p gpurun_out/r2/final
o=gpurun_out/r2/final
run() { local name=$1 limit=$2; shift 2; timeout -k 10 "$limit" "$@" > "$o/$name.json" 2> "$o/$name.err"; local rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; }
run bench_cfg4 300 python bench.py
run bench_cfg2 300 python bench.py --workload cfg2
run bench_cfg3 300 python bench.py --workload cfg3
run bench_cfg5 300 python bench.py --workload cfg5
run predict_cfg4 200 python bench.py --mode predict --steps 64
run predict_cfg3 200 python bench.py --mode predict --workload cfg3 --steps 32
run predict_cfg5 200 python bench.py --mode predict --workload cfg5 --steps 40
for w in cfg4 cfg3 cfg5; do bash tools/gpu_prof.sh $w --workload $w --steps 20 --warmup 5 > $o/prof_$w.log 2>&1; cp gpurun_out/r2/${w}_kernel_stats.csv $o/; done
bash tools/gpu_prof.sh cfg4_predict --mode predict --steps 40 --warmup 5 > $o/prof_cfg4_predict.log 2>&1; cp gpurun_out/r2/cfg4_predict_kernel_stats.csv $o/
for w in cfg4 cfg3; do
  bash tools/gpu_pmc.sh ${w}_fetch "FETCH_SIZE" --workload $w --steps 6 --warmup 2 > $o/pmc_${w}_fetch.log 2>&1
  bash tools/gpu_pmc.sh ${w}_write "WRITE_SIZE" --workload $w --steps 6 --warmup 2 > $o/pmc_${w}_write.log 2>&1
done
for w in cfg4 cfg3; do
  bash tools/gpu_pmc.sh ${w}_sq "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" --workload $w --steps 6 --warmup 2 > $o/pmc_${w}_sq.log 2>&1
done
bash tools/gpu_pmc.sh cfg4_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" --steps 6 --warmup 2 > $o/pmc_cfg4_lds.log 2>&1
grep -h "^siren\|^tiny\|^hash\|^bin\|^dense\|^adam" $o/pmc_*.log | cut -c1-300
cat $o/bench_cfg4.json | cut -c1-1500
