#!/bin/bash
# rocprofv3 kernel stats of the LWL solve (tools/bench_lwl.py); prints the per-kernel averages
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/lwlprof -o k -- python tools/bench_lwl.py --n ${1:-32} --iters 3 --reps 10 > /dev/null 2>&1
python tools/rocpd_summary.py $(find gpurun_out/lwlprof -name "*.db" | head -1) | cut -c1-140 | head -7
rm -rf gpurun_out/lwlprof
