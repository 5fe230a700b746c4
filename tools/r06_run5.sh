#!/bin/bash
# round 6, GPU call 5: transposed-accumulator 16-byte epilogue of k_gemm_ps (A/B), ToMP parity, kernel stats
export TMPDIR=/tmp
OUT=gpurun_out/r06e
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "tomp or smoke or apply_filter or feat_transpose or install or tracker" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_subset.log; tail -3 $OUT/pytest_subset.log
for rep in 1 2 3; do
  echo "tomp wide store : $(python tools/bench_tomp.py --graph 2>/dev/null | tail -1 | cut -c120-260)" | tee -a $OUT/tomp_store_ab.txt
  echo "tomp 4-byte     : $(PT_GEMM_WIDE_STORE=0 python tools/bench_tomp.py --graph 2>/dev/null | tail -1 | cut -c120-260)" | tee -a $OUT/tomp_store_ab.txt
done
rocprofv3 --kernel-trace --stats -d $OUT/p_tomp -o k -- python tools/bench_tomp.py --reps 10 > /dev/null 2>&1
python tools/rocpd_summary.py $(find $OUT/p_tomp -name "*.db" | head -1) | cut -c1-170 | head -14 | tee $OUT/tomp_kernel_stats.csv
rm -rf $OUT/p_tomp
