#!/bin/bash
# rocprofv3 kernel stats of the ATOM optimisers (tools/bench_atom.py); prints the per-kernel averages
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/atomprof -o k -- python tools/bench_atom.py > /dev/null 2>&1
python tools/rocpd_summary.py $(find gpurun_out/atomprof -name "*.db" | head -1) | cut -c1-150 | head -${1:-16} | tee gpurun_out/atom_kernels.csv
rm -rf gpurun_out/atomprof
