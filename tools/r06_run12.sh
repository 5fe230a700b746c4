#!/bin/bash
# round 6, GPU call 12: k_corr2's shift-and-add with uniform tap steps + row / column masks: parity subset, A/B against the per-tap index form
export TMPDIR=/tmp
OUT=gpurun_out/r06m
mkdir -p $OUT
V=$PWD/pytracking_amd/variants
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "apply_filter or feat_transpose or chain or closed_loop or sd_ or prdimp or frame or loss or atom" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_subset.log; tail -3 $OUT/pytest_subset.log
B="python bench.py --no-other --no-cpu-baseline --no-gpu-baseline --no-roofline"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "lds barrier  500 : $($B 2>/dev/null | val)" | tee -a $OUT/ldsbar_ab.txt
  echo "syncthreads  500 : $(PT_HOT_LIB=$V/libpt_hot_ldsbar0.so $B 2>/dev/null | val)" | tee -a $OUT/ldsbar_ab.txt
  echo "lds barrier  drv : $($B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ldsbar_ab.txt
  echo "syncthreads  drv : $(PT_HOT_LIB=$V/libpt_hot_ldsbar0.so $B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ldsbar_ab.txt
done
