#!/bin/bash
# usage: tools/bw.sh WORKLOAD [ENV=VAL ...] -- prints iter/s and sweep kernel ms of a bench workload (no CPU leg, one chain)
wl=$1; shift
env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu --no-extra --shards 0 --chains 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl $*', 'iter/s %.2f' % d['value'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], d['chain_check'])"
