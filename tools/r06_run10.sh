#!/bin/bash
# round 6, GPU call 10: re-sweep of the pass knobs (tuned in round 3, before sample pairs / frame chains) on the final kernels
export TMPDIR=/tmp
OUT=gpurun_out/r06k
mkdir -p $OUT
V=$PWD/pytracking_amd/variants
B="python bench.py --no-other --no-cpu-baseline --no-gpu-baseline --no-roofline"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "default : $($B 2>/dev/null | val)" | tee -a $OUT/knob_sweep.txt
  for v in cd1 cd3 cd4 pd2 pd4 pd6 minw5 minw7 minw8 early0 c2bar adjbar; do
    echo "$v : $(PT_HOT_LIB=$V/libpt_hot_k_$v.so $B 2>/dev/null | val)" | tee -a $OUT/knob_sweep.txt
  done
done
