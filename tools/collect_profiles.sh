#!/bin/bash
# Round profile set for the headline workload (run on the GPU box from the repo root):
#   bash tools/collect_profiles.sh r02
# writes gpurun_out/<tag>_* ; copy what is to be judged into profiles/ afterwards.
set -e
TAG=${1:-r02}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/${TAG}_bench_c4.json 2> $OUT/${TAG}_bench_c4.err
echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o ${TAG} --output-format csv -- python3 $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu --no-extra --shards 0 --chains 1 > $OUT/${TAG}_bench_c4_under_rocprof.json 2> $OUT/${TAG}_rocprof.err
echo "stats done"
# the concurrent-chains leg on its own (kernel durations with five sweeps side by side)
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_cc -o ${TAG}cc --output-format csv -- python3 $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu --no-extra --shards 0 > $OUT/${TAG}_bench_c4_concurrent_under_rocprof.json 2>> $OUT/${TAG}_rocprof.err
find $OUT/prof_stats_cc -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bench_c4_concurrent_kernel_stats.csv \;
find $OUT/prof_stats_cc -name "*kernel_trace.csv" -delete
echo "concurrent stats done"
if [ -n "$WITH_EM" ]; then
# the EM family at C2 size: kernel time split between the column gather, the Gram rebuild and the sweep
PYTHONPATH=$OLDPWD rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_em -o ${TAG}em --output-format csv -- python3 $OLDPWD/tools/em_probe.py 5000 50000 5 1 > $OUT/${TAG}_em_c2_under_rocprof.json 2>> $OUT/${TAG}_rocprof.err
find $OUT/prof_stats_em -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_em_c2_kernel_stats.csv \;
find $OUT/prof_stats_em -name "*kernel_trace.csv" -delete
echo "em stats done"
fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -o ${TAG} --output-format csv -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu --no-extra --shards 0 --chains 1 > $OUT/${TAG}_pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -o ${TAG} --output-format csv -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu --no-extra --shards 0 --chains 1 > $OUT/${TAG}_pmc_write.log 2>&1
echo "write done"
cd $OLDPWD
python3 tools/pmc_summary.py $OUT/prof_fetch $OUT/prof_write $OUT/${TAG}_pmc_c4.json c4 10000 1000000 > /dev/null
find $OUT/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bench_c4_kernel_stats.csv \;
# the raw per-dispatch counter files are large: keep the summaries only
rm -rf $OUT/prof_fetch $OUT/prof_write
find $OUT/prof_stats -name "*kernel_trace.csv" -delete
ls -la $OUT | head -30
