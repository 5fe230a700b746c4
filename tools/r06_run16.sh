#!/bin/bash
# round 6, GPU call 16: init stage folded into the first adjoint pass (k_adj2<.., INIT>): bit-identity tests, parity subset, A/B
export TMPDIR=/tmp
OUT=gpurun_out/r06p
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_init" > $OUT/pytest_fused.log 2>&1; echo "fused tests rc=$?" | tee -a $OUT/pytest_fused.log; tail -15 $OUT/pytest_fused.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_frame_full.py tests/test_install_device_dispatch.py -x -q -k "chain or closed_loop or sd_ or dimp or frame or loss or tracker or module" > $OUT/pytest_subset.log 2>&1; echo "subset rc=$?" | tee -a $OUT/pytest_subset.log; tail -3 $OUT/pytest_subset.log
B="python bench.py --no-other --no-cpu-baseline --no-gpu-baseline --no-roofline"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "fused    500 : $($B 2>/dev/null | val)" | tee -a $OUT/fuse_ab.txt
  echo "unfused  500 : $(PT_SD_FUSE_INIT=0 $B 2>/dev/null | val)" | tee -a $OUT/fuse_ab.txt
  echo "fused    drv : $($B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/fuse_ab.txt
  echo "unfused  drv : $(PT_SD_FUSE_INIT=0 $B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/fuse_ab.txt
done
