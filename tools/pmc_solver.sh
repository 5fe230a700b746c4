#!/bin/bash
# rocprofv3 PMC passes over a short bench.py run: instruction mix and wave occupancy of the solver kernels (one counter
# group per pass, kernel-trace only) -> gpurun_out/TAG/pmc_*.txt
TAG=${1:-solver_pmc}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-graph ${PT_PMC_ARGS} > $OUT/p$i.json 2> $OUT/p$i.err
  python tools/pmc_summary.py $OUT/p$i > $OUT/pmc_$i.txt
  rm -rf $OUT/p$i
done
cat $OUT/pmc_*.txt | grep -A7 -E "k_adj2|k_corr2|k_fast_sgq" | grep -v -E "rocclr|at::native"
