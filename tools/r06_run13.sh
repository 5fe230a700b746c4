#!/bin/bash
# round 6, GPU call 13: k_fast_init2 requests the three look-up tables with its first loads (one contiguous array): parity subset + A/B
export TMPDIR=/tmp
OUT=gpurun_out/r06n
mkdir -p $OUT
V=$PWD/pytracking_amd/variants
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_frame_full.py tests/test_install_device_dispatch.py -x -q -k "chain or closed_loop or sd_ or dimp or frame or loss or tracker or module" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_subset.log; tail -3 $OUT/pytest_subset.log
B="python bench.py --no-other --no-cpu-baseline --no-gpu-baseline --no-roofline"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "lut hot   500 : $($B 2>/dev/null | val)" | tee -a $OUT/lut_ab.txt
  echo "lut late  500 : $(PT_HOT_LIB=$V/libpt_hot_lutlate.so $B 2>/dev/null | val)" | tee -a $OUT/lut_ab.txt
  echo "lut hot   drv : $($B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/lut_ab.txt
  echo "lut late  drv : $(PT_HOT_LIB=$V/libpt_hot_lutlate.so $B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/lut_ab.txt
done
