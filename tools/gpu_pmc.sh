#!/bin/bash
# usage: tools/gpu_pmc.sh TAG [bench args]  -- rocprofv3 PMC passes over a short bench.py run (one counter group per pass,
# kernel-trace only, as MI355X_MICROARCH.md prescribes); per-kernel averages -> gpurun_out/TAG/pmc_*.txt
# --no-other: the side workloads poll pinned host memory (localisation / IoU results); under counter collection that run did not finish
# within 25 minutes (round 3) -- keep them out of PMC passes.
TAG=${1:-pmc}; shift
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-graph --no-other "$@" > $OUT/p$i.json 2> $OUT/p$i.err
  python tools/pmc_summary.py $OUT/p$i > $OUT/pmc_$i.txt
  rm -rf $OUT/p$i
done
cat $OUT/pmc_*.txt | grep -v -E "rocclr|at::native" 
