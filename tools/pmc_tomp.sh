#!/bin/bash
# rocprofv3 PMC passes over the ToMP frame (tools/bench_tomp.py), one counter group per pass, kernel-trace only
# (MI355X_MICROARCH.md: never combine --pmc with the tracing domains); per-kernel averages -> gpurun_out/TAG/pmc_*.txt
TAG=${1:-tomp_pmc}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python tools/bench_tomp.py --reps 3 > $OUT/p$i.json 2> $OUT/p$i.err
  python tools/pmc_summary.py $OUT/p$i > $OUT/pmc_$i.txt
  rm -rf $OUT/p$i
done
cat $OUT/pmc_*.txt | grep -A8 -E "k_gemm|k_attn" | grep -v -E "rocclr|at::native"
