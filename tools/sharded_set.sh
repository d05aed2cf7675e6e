#!/bin/bash
# The sharded leg's single-GPU evidence (run on the GPU box from the repo root): one rank's eighth of C4 with its eight exchange rounds
# (implicitly centred int8 and, for comparison, uncentred), the whole C4 panel as one implicitly centred shard, and two ranks on the one GPU.
#   bash tools/sharded_set.sh r04
TAG=${1:-r04}
OUT=gpurun_out
mkdir -p $OUT
python tools/shard_probe.py 8 implicit > $OUT/${TAG}_shard_probe_g8_implicit.json 2> $OUT/sp.err; cat $OUT/${TAG}_shard_probe_g8_implicit.json
python tools/shard_probe.py 8 none > $OUT/${TAG}_shard_probe_g8_none.json 2>> $OUT/sp.err; cat $OUT/${TAG}_shard_probe_g8_none.json
BWGR_FORCE_DIST=1 BWGR_FORCE_CENTRE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 python bench.py --sharded --steps 10 --warmup 2 > $OUT/${TAG}_sharded_rehearsal_c4_implicit_1rank.json 2> $OUT/r1.err
tail -c 1200 $OUT/${TAG}_sharded_rehearsal_c4_implicit_1rank.json; echo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --sharded --steps 10 --warmup 2 > $OUT/${TAG}_sharded_rehearsal_c4_implicit_2ranks_one_gpu.json 2> $OUT/r2.err
tail -c 1200 $OUT/${TAG}_sharded_rehearsal_c4_implicit_2ranks_one_gpu.json; echo; tail -3 $OUT/r2.err
# config 5's share: an eighth of the 50k x 1M panel under BayesCpi (dense inclusion: k_sweep2, 32-bit Gram entries), implicitly centred int8
SP_N=50000 SP_MODEL=BayesCpi SP_PI=0.5 SP_K=10 python tools/shard_probe.py 8 implicit > $OUT/${TAG}_shard_probe_c5_g8_implicit.json 2>> $OUT/sp.err; cat $OUT/${TAG}_shard_probe_c5_g8_implicit.json
