#!/bin/bash
# round 6, GPU call 19: ToMP encoder norm1 folded into the FFN's first product (k_gemm_ps<.., LNA>): parity + A/B
export TMPDIR=/tmp
OUT=gpurun_out/r06r
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tomp" > $OUT/pytest_tomp.log 2>&1; echo "tomp tests rc=$?" | tee -a $OUT/pytest_tomp.log; tail -4 $OUT/pytest_tomp.log
for rep in 1 2 3; do
  echo "ln1 folded  : $(python tools/bench_tomp.py --graph --reps 200 2>/dev/null | tail -1)" | tee -a $OUT/tomp_ab.txt
  echo "ln1 launch  : $(PT_TOMP_LN1_FOLD=0 python tools/bench_tomp.py --graph --reps 200 2>/dev/null | tail -1)" | tee -a $OUT/tomp_ab.txt
done
rocprofv3 --kernel-trace --stats -d $OUT/prof -o k -- python tools/bench_tomp.py --reps 10 > /dev/null 2>&1
python tools/rocpd_summary.py $(find $OUT/prof -name "*.db" | head -1) > $OUT/tomp_kernel_stats.csv
python tools/short_stats.py 14 < $OUT/tomp_kernel_stats.csv
rm -rf $OUT/prof
