#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 300 python tools/siren_chain_check.py > gpurun_out/r2/chain1.out 2> gpurun_out/r2/chain1.err
echo rc=$?; tail -30 gpurun_out/r2/chain1.out; tail -5 gpurun_out/r2/chain1.err
