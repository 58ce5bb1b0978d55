#!/bin/bash
# round-2 batch 1: new tests, bench lines (cfg4 / cfg5 / predict), fuzz sweep for kink seeds
set -o pipefail
mkdir -p gpurun_out/r2
run() {  # name, timeout, command...: stop the whole batch if a step hangs or is killed
  local name=$1 limit=$2; shift 2
  timeout -k 10 "$limit" "$@" > "gpurun_out/r2/$name.out" 2> "gpurun_out/r2/$name.err"
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out: stopping"; exit 1; fi
  return 0
}
run tests 1000 python -m pytest tests -x -q -m gpu
tail -5 gpurun_out/r2/tests.out
run bench_cfg4 300 python bench.py
run bench_cfg5 300 python bench.py --workload cfg5 --no-cpu-baseline
run predict_cfg4 200 python bench.py --mode predict --steps 64
run predict_cfg3 200 python bench.py --mode predict --workload cfg3 --steps 32
run fuzz 500 python tools/fuzz.py --encoders 60 --layers 40 --steps 400 --sirens 20
tail -3 gpurun_out/r2/fuzz.out
cat gpurun_out/r2/bench_cfg4.out gpurun_out/r2/bench_cfg5.out gpurun_out/r2/predict_cfg4.out gpurun_out/r2/predict_cfg3.out
