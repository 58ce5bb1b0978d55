#!/bin/bash
# two-barrier decoder kernel: parity tests, then A/B against the three-barrier kernel on one box
o=gpurun_out/r4/exp6; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q -k "tiny_mlp or decoder or fused or full_size_cfg4 or steady or extreme or smoke or e2e_hash or relu_kink" > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $o/pytest.log; [ $rc -ne 0 ] && exit 1
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],4), d['phases_ms'], 'psnr', d.get('psnr'))
PY
}
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-records-leg --no-cpu-baseline --psnr-steps 500 "$@" > $o/$name.json 2> $o/$name.err; rc=$?; [ $rc -ne 0 ] && { echo "$name rc=$rc"; tail -3 $o/$name.err; }; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1; line $o/$name.json; }
run new_a
run old_a --opt mlp_x3=3
run new_b
run old_b --opt mlp_x3=3
run new_cfg5 --workload cfg5
run old_cfg5 --workload cfg5 --opt mlp_x3=3
