#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -m gpu -x -k "gradient or backward or reproducible or record or full_size or bucketed or level_mask or fused_table or counting or config5 or steady or encoder" > $o/tests_n.out 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $o/tests_n.out
[ $rc -ne 0 ] && exit 1
one() { timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --psnr-steps 0 --no-records-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-24s' % '$MRI_LIB'[-14:], '%-16s' % '$*', round(d['ms_per_step'], 4), d['phases_ms'], d['final_loss'])"; }
for rep in 1 2; do
unset MRI_LIB; one; one --workload cfg2; one --workload cfg5
export MRI_LIB=$GRAFT_REPO_ROOT/tools/libmri_old.so; one; one --workload cfg2; one --workload cfg5
done
unset MRI_LIB
MRI_ROUND=r3 bash tools/gpu_prof.sh pairvals --steps 20 --warmup 5 --no-records-leg > $o/prof_pairvals.log 2>&1; grep -E "bin_kernel|dense_and" $o/pairvals_kernel_stats.csv
