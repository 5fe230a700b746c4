#!/bin/bash
# round 6, GPU call 21: attention ablations (PT_ATTN_EXP builds, wrong results, timing only): where is the non-MFMA half of k_attn?
export TMPDIR=/tmp
OUT=gpurun_out/r06t
mkdir -p $OUT
V=$PWD/pytracking_amd/variants
ms() { python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms'])"; }
for rep in 1 2; do
  echo "product                          : $(timeout 120 python tools/bench_tomp.py --graph --reps 200 2>/dev/null | ms)" | tee -a $OUT/attn_ablation.txt
  for v in 1 2 4 8 15; do
    echo "PT_ATTN_EXP=$v                    : $(PT_HOT_LIB=$V/libpt_hot_attn$v.so timeout 120 python tools/bench_tomp.py --graph --reps 200 2>/dev/null | ms)" | tee -a $OUT/attn_ablation.txt
  done
done
PT_HOT_LIB=$V/libpt_hot_attn15.so timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o k -- python tools/bench_tomp.py --reps 10 > /dev/null 2>&1
python tools/rocpd_summary.py $(find $OUT/prof -name "*.db" | head -1) | python tools/short_stats.py 4 | tee -a $OUT/attn_ablation.txt
rm -rf $OUT/prof
