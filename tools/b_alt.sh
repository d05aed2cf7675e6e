#!/bin/bash
# usage: tools/b_alt.sh path/to/alt.so [ENV=VAL ...] -- the C4 bench (no CPU leg, one chain) on another build of the library
so=$1; shift
env "$@" timeout -k 10 300 python -c "
import sys, runpy
from bwgr_amd import build as B
B.LIB = '$so'
B.needs_build = lambda: False
sys.argv = ['bench.py', '--steps', '10', '--warmup', '3', '--no-cpu', '--chains', '1']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$so $*', 'iter/s %.2f' % d['value'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], d['chain_check'])"
