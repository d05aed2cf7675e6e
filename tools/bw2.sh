#!/bin/bash
# usage: tools/bw2.sh WORKLOAD [ENV=VAL ...] -- iter/s and sweep kernel ms of one bench workload (no CPU leg, one chain)
WL=$1; shift
env "$@" timeout -k 10 300 python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu --no-extra --shards 0 --chains 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$WL $*', 'iter/s %.2f' % d['value'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], d['chain_check'])"
