#!/bin/bash
# usage: tools/gpu_bench_prof.sh TAG   -- bench + rocprofv3 kernel stats on the GPU box, results under gpurun_out/TAG
TAG=${1:-run}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --steps 500 --warmup 50 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/prof -o k -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err; echo "prof rc=$?"
python tools/rocpd_summary.py $(find $OUT/prof -name "*.db" | head -1) > $OUT/kernel_stats.csv
cat $OUT/kernel_stats.csv | cut -c1-200
rm -rf $OUT/prof
