#!/bin/bash
# usage: tools/b.sh [ENV=VAL ...] -- prints iter/s and sweep kernel ms of the C4 bench (no CPU leg, one chain)
env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-extra --shards 0 --chains 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', 'iter/s %.2f' % d['value'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], d['chain_check'])"
