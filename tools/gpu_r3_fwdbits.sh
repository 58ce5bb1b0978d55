#!/bin/bash
# lookup with cache-policy bits on the table gathers: same-box A/B + the rest of the GPU tests
o=gpurun_out/r3; mkdir -p $o
for lib in - tools/libmri_fwd_sc0.so tools/libmri_fwd_sc1.so tools/libmri_fwd_sc0sc1.so -; do
  if [ "$lib" = "-" ]; then unset MRI_LIB; else export MRI_LIB=$GRAFT_REPO_ROOT/$lib; fi
  timeout -k 10 120 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --psnr-steps 0 --no-records-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % '$lib', round(d['ms_per_step'], 4), d.get('phases_ms'))"
done
unset MRI_LIB
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --deselect tests/test_00_dp_two_rank_gpu.py > $o/tests_all2.out 2>&1; echo "tests rc=$?"; tail -5 $o/tests_all2.out
