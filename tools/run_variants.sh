#!/bin/bash
# usage: tools/run_variants.sh OUTDIR "n list" name...   -- exp_pass_vs_n.py for the default library and each named variant
OUT=$1; NS=$2; shift 2
mkdir -p $OUT
echo "== default"; python tools/exp_pass_vs_n.py $NS 2>/dev/null | tee $OUT/default.jsonl
for v in "$@"; do
  echo "== $v"; PT_HOT_LIB=$PWD/pytracking_amd/variants/libpt_hot_$v.so python tools/exp_pass_vs_n.py $NS 2>/dev/null | tee $OUT/$v.jsonl
done
