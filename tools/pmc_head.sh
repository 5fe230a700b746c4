#!/bin/bash
# rocprofv3 PMC passes over the classification-feature head (tools/bench_head.py), one counter group per pass, kernel-trace only;
# per-kernel averages of the head's correlation kernel -> gpurun_out/TAG/pmc_head.txt
TAG=${1:-head_pmc}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python tools/bench_head.py > $OUT/p$i.json 2> $OUT/p$i.err
  python tools/pmc_summary.py $OUT/p$i > $OUT/pmc_$i.txt
  rm -rf $OUT/p$i
done
cat $OUT/pmc_*.txt | grep -A8 -E "k_mf_corr<9, 2, true" | grep -v -E "rocclr|at::native" > $OUT/pmc_head.txt
cat $OUT/pmc_head.txt
