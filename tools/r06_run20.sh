#!/bin/bash
# round 6, GPU call 20: runtime knobs of the HIP runtime that touch dependent-launch cost (kernel-argument placement, graph packet capture)
export TMPDIR=/tmp
OUT=gpurun_out/r06s
mkdir -p $OUT
B="timeout 120 python bench.py --no-other --no-cpu-baseline --no-gpu-baseline --no-roofline"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run() { echo "$1 : $(env $2 $B 2>/dev/null | val)   drv $(env $2 $B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/runtime_knobs.txt; }
for rep in 1 2; do
  run "default                         " "X=1"
  run "HIP_FORCE_DEV_KERNARG=1         " "HIP_FORCE_DEV_KERNARG=1"
  run "HIP_FORCE_DEV_KERNARG=0         " "HIP_FORCE_DEV_KERNARG=0"
  run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
  run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
  # (ROC_SYSTEM_SCOPE_SIGNAL=0 hangs the process: removed from the sweep)
  run "DEBUG_HIP_KERNARG_COPY_OPT=0    " "DEBUG_HIP_KERNARG_COPY_OPT=0"
  run "ROC_USE_FGS_KERNARG=0           " "ROC_USE_FGS_KERNARG=0"
  run "GPU_FLUSH_ON_EXECUTION=1        " "GPU_FLUSH_ON_EXECUTION=1"
  run "DEBUG_HIP_GRAPH_BATCH_SIZE=1000 " "DEBUG_HIP_GRAPH_BATCH_SIZE=1000"
done
