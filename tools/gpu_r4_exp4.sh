#!/bin/bash
# how often does a 40-step cold leg hit the ~50 ms host stall, with and without priming the timing events?
o=gpurun_out/r4/exp4; mkdir -p $o
for i in 1 2 3 4 5 6; do
 for mode in a b; do
  true
  MRI_STEP_TIMES=1 timeout -k 10 120 python bench.py --steps 40 --warmup 10 --psnr-steps 0 --no-cpu-baseline --no-records-leg > $o/$mode$i.json 2> $o/$mode$i.err || exit 1
  python - $o/$mode$i <<'PY'
import json,sys,re
d=json.load(open(sys.argv[1]+'.json'))
t=[l for l in open(sys.argv[1]+'.err') if l.startswith('host ms')]
v=eval(t[-1].split(':',1)[1]) if t else []
big=[(i,x) for i,x in enumerate(v) if x>3]
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],4), 'samples', d['phases_samples'], 'stalls', big)
PY
 done
done
