#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 900 python -m pytest tests -q -m gpu -x --deselect tests/test_00_dp_two_rank_gpu.py > $o/tests_e.out 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $o/tests_e.out
[ $rc -ne 0 ] && exit 1
one() { timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --psnr-steps 0 --no-records-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-30s' % '$MRI_LIB'[-20:], '%-16s' % '$*', round(d['ms_per_step'], 4), d['phases_ms'], d['final_loss'])"; }
unset MRI_LIB; one; one --workload cfg2
export MRI_LIB=$GRAFT_REPO_ROOT/tools/libmri_old.so; one; one --workload cfg2
unset MRI_LIB; one; one --workload cfg2
