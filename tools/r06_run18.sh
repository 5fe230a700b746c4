#!/bin/bash
# round 6, GPU call 18: branch-free look-up-table interpolation (init stage), fused init stage again (opt-in), parity subset
export TMPDIR=/tmp
OUT=gpurun_out/r06q
mkdir -p $OUT
V=$PWD/pytracking_amd/variants
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_frame_full.py tests/test_install_device_dispatch.py -x -q -k "fused_init or chain or closed_loop or sd_ or dimp or frame or loss or tracker or module" > $OUT/pytest_subset.log 2>&1; echo "subset rc=$?" | tee -a $OUT/pytest_subset.log; tail -3 $OUT/pytest_subset.log
B="python bench.py --no-other --no-cpu-baseline --no-gpu-baseline --no-roofline"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "branch-free 500 : $($B 2>/dev/null | val)" | tee -a $OUT/lut_branch_ab.txt
  echo "branchy     500 : $(PT_HOT_LIB=$V/libpt_hot_lutbranchy.so $B 2>/dev/null | val)" | tee -a $OUT/lut_branch_ab.txt
  echo "fused init  500 : $(PT_SD_FUSE_INIT=1 $B 2>/dev/null | val)" | tee -a $OUT/lut_branch_ab.txt
done
