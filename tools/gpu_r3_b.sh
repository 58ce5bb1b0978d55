#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -m gpu -k "e2e_hash_adam_golden or e2e_siren_adam_golden or runs_the_reference_goldens or gelu_notebook" > $o/tests_b.out 2>&1; echo "tests rc=$?"; tail -3 $o/tests_b.out
MRI_ROUND=r3 bash tools/gpu_final.sh bench
