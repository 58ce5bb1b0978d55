#!/bin/bash
# round 4, batch 2: pair-routed records -- tests, then the bench in driver form and default form
o=gpurun_out/r4/exp2; mkdir -p $o
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],4), 'host', d.get('host_queue_ms_per_step'), d['phases_ms'], 'samples', d.get('phases_samples'), 'packed', (d.get('packed_records') or {}).get('ms_per_step'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $o/pytest.log; [ $rc -ne 0 ] && exit 1
for pe in 4 0; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --phase-every $pe --no-cpu-baseline > $o/drv_pe$pe.json 2> $o/drv_pe$pe.err || exit 1
  line $o/drv_pe$pe.json
done
for w in cfg4 cfg2 cfg5; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $o/bench_$w.json 2> $o/bench_$w.err || exit 1
  line $o/bench_$w.json
done
