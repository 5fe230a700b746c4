#!/bin/bash
# rocprofv3 PMC passes over the LWL solve (tools/bench_lwl.py), one counter group per pass, kernel-trace only
# (MI355X_MICROARCH.md: never combine --pmc with the tracing domains); per-kernel averages -> gpurun_out/TAG/pmc_*.txt
TAG=${1:-lwl_pmc}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python tools/bench_lwl.py --n 32 --iters 3 --reps 3 > $OUT/p$i.json 2> $OUT/p$i.err
  python tools/pmc_summary.py $OUT/p$i > $OUT/pmc_$i.txt
  rm -rf $OUT/p$i
done
cat $OUT/pmc_*.txt | grep -A8 -E "k_mf_corr|k_mf_adj" | grep -v -E "rocclr|at::native"
