#!/bin/bash
# rocprofv3 kernel stats of the IoU refinement (tools/bench_iou.py); prints the per-kernel averages
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/iouprof -o k -- python tools/bench_iou.py > /dev/null 2>&1
python tools/rocpd_summary.py $(find gpurun_out/iouprof -name "*.db" | head -1) | cut -c1-150 | head -${1:-24} | tee gpurun_out/iou_kernels.csv
rm -rf gpurun_out/iouprof
