#!/bin/bash
one() { timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --psnr-steps 0 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % '$*', round(d['ms_per_step'], 4), 'host', d['host_queue_ms_per_step'], d['phases_ms'], d['final_loss'], d.get('packed_records', {}).get('ms_per_step'))"; }
one; one --no-graph; one --workload cfg2; one --workload cfg2 --no-graph; one --workload cfg5; one --workload cfg5 --no-graph; one --phase-every 1000000; one --workload cfg3 --steps 20 --warmup 5
timeout -k 10 300 python bench.py | cut -c1-2500
