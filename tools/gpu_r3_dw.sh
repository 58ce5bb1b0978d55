#!/bin/bash
timeout -k 10 280 python tools/dw_terms_probe.py
MRI_LIB=$GRAFT_REPO_ROOT/tools/libmri_dw3.so timeout -k 10 280 python tools/dw_terms_probe.py
one() { timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --psnr-steps 0 --no-records-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-24s' % '$MRI_LIB'[-14:], round(d['ms_per_step'], 4), d['phases_ms'])"; }
unset MRI_LIB; one; export MRI_LIB=$GRAFT_REPO_ROOT/tools/libmri_dw3.so; one; unset MRI_LIB; one; export MRI_LIB=$GRAFT_REPO_ROOT/tools/libmri_dw3.so; one
