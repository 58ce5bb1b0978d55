#!/bin/bash
# the cfg3 part of tools/gpu_final.sh (the rows kernels changed after the round's batch)
export MRI_ROUND=r4
o=gpurun_out/$MRI_ROUND/final; mkdir -p $o
run() { local name=$1 limit=$2; shift 2; timeout -k 10 "$limit" "$@" > "$o/$name.json" 2> "$o/$name.err"; local rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; }
run bench_cfg3 300 python bench.py --workload cfg3
run predict_cfg3 200 python bench.py --mode predict --workload cfg3 --steps 32
bash tools/gpu_prof.sh cfg3 --workload cfg3 --steps 20 --warmup 5 --no-records-leg > $o/prof_cfg3.log 2>&1 || exit 1; cp gpurun_out/$MRI_ROUND/cfg3_kernel_stats.csv $o/
bash tools/gpu_pmc.sh cfg3_fetch "FETCH_SIZE" --workload cfg3 --steps 6 --warmup 2 --no-records-leg > $o/pmc_cfg3_fetch.log 2>&1 || exit 1
bash tools/gpu_pmc.sh cfg3_write "WRITE_SIZE" --workload cfg3 --steps 6 --warmup 2 --no-records-leg > $o/pmc_cfg3_write.log 2>&1 || exit 1
bash tools/gpu_pmc.sh cfg3_sq "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" --workload cfg3 --steps 6 --warmup 2 --no-records-leg > $o/pmc_cfg3_sq.log 2>&1 || exit 1
cut -c1-600 $o/bench_cfg3.json
