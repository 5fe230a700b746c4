#!/bin/bash
# rocprofv3 kernel stats of the ToMP frame (tools/bench_tomp.py); prints the per-kernel averages
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/tompprof -o k -- python tools/bench_tomp.py --reps 10 > /dev/null 2>&1
python tools/rocpd_summary.py $(find gpurun_out/tompprof -name "*.db" | head -1) | cut -c1-150 | head -${1:-24} | tee gpurun_out/tomp_kernels.csv
rm -rf gpurun_out/tompprof
