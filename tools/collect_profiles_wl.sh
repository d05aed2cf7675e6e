#!/bin/bash
# Profile set for one bench workload (run on the GPU box from the repo root):
#   BWGR_COMMIT=<git head> bash tools/collect_profiles_wl.sh r03 c5 50000 1000000
# writes gpurun_out/<tag>_bench_<wl>.json, <tag>_bench_<wl>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_pmc_<wl>.json
# (separate --pmc FETCH_SIZE / WRITE_SIZE passes, summarised by tools/pmc_summary.py); copy what is to be judged into profiles/.
set -e
TAG=$1; WL=$2; N=$3; P=$4
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$PMC_ONLY" ]; then
python3 bench.py --workload $WL --chains 1 --no-extra --shards 0 ${BENCH_EXTRA} > $OUT/${TAG}_bench_${WL}.json 2> $OUT/${TAG}_bench_${WL}.err
echo "$WL bench done"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_${WL} -o ${TAG}${WL} --output-format csv -- python3 $REPO/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu --no-extra --shards 0 --chains 1 > $OUT/${TAG}_bench_${WL}_under_rocprof.json 2> $OUT/${TAG}_rocprof_${WL}.err
find $OUT/prof_stats_${WL} -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bench_${WL}_kernel_stats.csv \;
rm -rf $OUT/prof_stats_${WL}
echo "$WL stats done"
fi
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch_${WL} -o ${TAG} --output-format csv -- python3 $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu --no-extra --shards 0 --chains 1 > $OUT/${TAG}_pmc_fetch_${WL}.log 2>&1
echo "$WL fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write_${WL} -o ${TAG} --output-format csv -- python3 $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu --no-extra --shards 0 --chains 1 > $OUT/${TAG}_pmc_write_${WL}.log 2>&1
echo "$WL write done"
cd $REPO
python3 tools/pmc_summary.py $OUT/prof_fetch_${WL} $OUT/prof_write_${WL} $OUT/${TAG}_pmc_${WL}.json $WL $N $P > /dev/null
rm -rf $OUT/prof_fetch_${WL} $OUT/prof_write_${WL}
echo "$WL pmc summary done"
