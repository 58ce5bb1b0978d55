#!/bin/bash
# per-config sweep of the table gradient's dense / binned split (cfg5: D = 4; cfg2)
o=gpurun_out/r4/exp9; mkdir -p $o
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],4), d['phases_ms'])
PY
}
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-records-leg --no-cpu-baseline --psnr-steps 300 "$@" > $o/$name.json 2> $o/$name.err; rc=$?; [ $rc -ne 0 ] && { echo "$name rc=$rc"; tail -3 $o/$name.err; }; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1; line $o/$name.json; }
for w in cfg5 cfg2; do
run ${w}_base --workload $w
run ${w}_d8 --workload $w --opt bwd_dense_max_parts=8
run ${w}_d2 --workload $w --opt bwd_dense_max_parts=2
run ${w}_d8_b192 --workload $w --opt bwd_dense_max_parts=8 --opt bwd_dense_blocks=192
run ${w}_bpl96 --workload $w --opt bwd_blocks_per_level=96
run ${w}_base2 --workload $w
done
