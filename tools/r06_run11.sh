#!/bin/bash
# round 6, GPU call 11: ablations of the two passes (wrong results, timing only): k_adj2's four scalar LDS gathers per group as one aligned 16-byte
# read; k_corr2 without its shift-and-add epilogue -- upper bounds of what reworking either could return
export TMPDIR=/tmp
OUT=gpurun_out/r06l
mkdir -p $OUT
V=$PWD/pytracking_amd/variants
B="python bench.py --no-other --no-cpu-baseline --no-gpu-baseline --no-roofline"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "default : $($B 2>/dev/null | val)" | tee -a $OUT/pass_ablation.txt
  echo "adjexp1 : $(PT_HOT_LIB=$V/libpt_hot_adjexp1.so $B 2>/dev/null | val)" | tee -a $OUT/pass_ablation.txt
  echo "c2exp1 : $(PT_HOT_LIB=$V/libpt_hot_c2exp1.so $B 2>/dev/null | val)" | tee -a $OUT/pass_ablation.txt
done
