#!/bin/bash
# first GPU pass of the session: parity tests, bench, rocprof stats
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
python bench.py --steps 500 --warmup 50 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01 -o r01 -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-graph > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; echo "prof rc=$?"
cat gpurun_out/prof_bench.json
find gpurun_out/prof_r01 -name "*kernel_stats*" | head
