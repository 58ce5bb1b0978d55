#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
run() {
  local name=$1 limit=$2; shift 2
  timeout -k 10 "$limit" "$@" > "gpurun_out/r2/$name.out" 2> "gpurun_out/r2/$name.err"
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out: stopping"; exit 1; fi
  return 0
}
run tests2 1000 python -m pytest tests -x -q -m gpu
tail -5 gpurun_out/r2/tests2.out
run fuzz2 600 python tools/fuzz.py --encoders 100 --layers 40 --steps 600 --sirens 20
grep -n "kink\|FAIL\|UNEXPL\|finished" gpurun_out/r2/fuzz2.out | head -40
