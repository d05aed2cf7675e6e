#!/bin/bash
# usage: tools/cpi_alt.sh path/to/alt.so [n p [model]] [ENV=VAL ...] (CPI_PI=x: the model's pi, default 0.99) -- dense-inclusion sweeps (default BayesCpi at 10000 x 200000) on another build of the library
so=$1; shift
n=10000; p=200000; model=BayesCpi
if [[ "$1" =~ ^[0-9]+$ ]]; then n=$1; p=$2; shift 2; fi
if [[ "$1" =~ ^Bayes ]]; then model=$1; shift; fi
env "$@" timeout -k 10 300 python -c "
import sys, time, torch
sys.path.insert(0, '.')
from bwgr_amd import build as B
B.LIB = '$so'
B.needs_build = lambda: False
import bwgr_amd
from bwgr_amd import synth
n, p = $n, $p
X = synth.genotypes(n, p, device=0); y = synth.scale_phenotype(synth.phenotype(X, n))
P = bwgr_amd.Panel(X, n=n, device=0); del X
import os
ch = bwgr_amd.Chain(P, '$model', y, it=100, bi=0, pi=float(os.environ.get('CPI_PI', '0.99')), df=5, R2=0.5, seed=7)
ch.run(5); ch.sync()
torch.cuda.synchronize(); t0 = time.perf_counter(); ch.run(20); ch.sync(); t1 = time.perf_counter()
st = ch.state(); nb = (p + 127) // 128
print('$so $*', '$model', 'iter/s %.2f' % (20 / (t1 - t0)), 'us/block %.2f' % ((t1 - t0) / 20 / nb * 1e6), 'mean d %.4f' % float(st['d'].mean()), 've %.6f' % float(st['ve']))
" 2>&1 | grep -v amdgpu.ids
