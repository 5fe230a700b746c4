#!/bin/bash
# round 6, GPU call 3: full parity suite; sgq2 single-barrier A/B; ToMP FFN on 128x64 tiles (k_gemm_ps) A/B + kernel stats
export TMPDIR=/tmp
OUT=gpurun_out/r06c
mkdir -p $OUT
V=$PWD/pytracking_amd/variants
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
B="python bench.py --no-other --no-cpu-baseline --no-gpu-baseline --no-roofline"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "new reduction  driver-style: $($B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/sgq_ab.txt
  echo "old reduction  driver-style: $(PT_HOT_LIB=$V/libpt_hot_sgq8.so $B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/sgq_ab.txt
  echo "new reduction  500 steps   : $($B 2>/dev/null | val)" | tee -a $OUT/sgq_ab.txt
  echo "old reduction  500 steps   : $(PT_HOT_LIB=$V/libpt_hot_sgq8.so $B 2>/dev/null | val)" | tee -a $OUT/sgq_ab.txt
done
for rep in 1 2 3; do
  echo "tomp ps+split4 : $(python tools/bench_tomp.py --graph 2>/dev/null | tail -1)" | tee -a $OUT/tomp_ab.txt
  echo "tomp ps, split2: $(PT_TOMP_FFN2_SPLIT=2 python tools/bench_tomp.py --graph 2>/dev/null | tail -1)" | tee -a $OUT/tomp_ab.txt
  echo "tomp r5 routes : $(PT_GEMM_PS=0 PT_TOMP_FFN2_SPLIT=2 python tools/bench_tomp.py --graph 2>/dev/null | tail -1)" | tee -a $OUT/tomp_ab.txt
done
rocprofv3 --kernel-trace --stats -d $OUT/p_tomp -o k -- python tools/bench_tomp.py --reps 10 > /dev/null 2>&1
python tools/rocpd_summary.py $(find $OUT/p_tomp -name "*.db" | head -1) | cut -c1-170 | head -24 | tee $OUT/tomp_kernel_stats.csv
python tools/rocpd_by_grid.py $(find $OUT/p_tomp -name "*.db" | head -1) 2>/dev/null | head -40 > $OUT/tomp_kernels_by_grid.txt
rm -rf $OUT/p_tomp
