#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 280 python -X faulthandler -c "
import faulthandler, sys, runpy
faulthandler.dump_traceback_later(100, repeat=True, file=open('$o/hang_trace.txt', 'w'))
sys.argv = ['bench.py']
runpy.run_path('bench.py', run_name='__main__')
" > $o/hang_bench.json 2> $o/hang_bench.err; echo "rc=$?"; tail -c 600 $o/hang_bench.json; tail -40 $o/hang_trace.txt
