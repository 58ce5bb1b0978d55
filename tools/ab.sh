#!/bin/bash
# Same-box A/B of library builds: tools/ab.sh OUTDIR REPEATS lib1 lib2 ...   ("-" = the in-tree build)
# (bench.py default workload; one line per run: ms/step and the phase times)
out=gpurun_out/$1; reps=$2; shift 2
mkdir -p $out
for i in $(seq $reps); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset MRI_LIB; else export MRI_LIB=$lib; fi
    python bench.py --steps 300 --warmup 30 $AB_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % '$lib', round(d['ms_per_step'], 4), d.get('phases_ms'))" >> $out/ab.log
  done
done
