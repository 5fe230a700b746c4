#!/bin/bash
# rocprofv3 kernel stats of the IoU refinement (tools/bench_iou.py); prints the per-kernel averages
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/iouprof -o k -- python tools/bench_iou.py > /dev/null 2>&1
python tools/rocpd_summary.py $(find gpurun_out/iouprof -name "*.db" | head -1) > gpurun_out/iou_kernels.csv
NROWS=${1:-24} python - <<'PY'
import csv
for r in list(csv.reader(open('gpurun_out/iou_kernels.csv')))[:int(__import__("os").environ.get("NROWS", "24"))]:
    if len(r) > 5:
        print(f"{r[0].replace('(anonymous namespace)::', '')[:44]:46s} calls {r[1]:>6s}  avg {r[3]:>9s}  min {r[4]:>8s}  max {r[5]:>8s}")
PY
rm -rf gpurun_out/iouprof
