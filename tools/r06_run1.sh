#!/bin/bash
# round 6, GPU call 1: parity suite with the conflict-free LDS layouts of the banded correlation / adjoint, A/B against the
# round-5 object (variants/libpt_hot_r5mf.so), head counters, baseline bench of this box.  Results -> gpurun_out/r06a
export TMPDIR=/tmp
OUT=gpurun_out/r06a
mkdir -p $OUT
OLD=$PWD/pytracking_amd/variants/libpt_hot_r5mf.so
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
for rep in 1 2; do
  echo "== head new";  python tools/bench_head.py 2>/dev/null | tee -a $OUT/head_new.jsonl
  echo "== head old";  PT_HOT_LIB=$OLD python tools/bench_head.py 2>/dev/null | tee -a $OUT/head_old.jsonl
done
for n in 1 4 8 32; do
  echo "== lwl n=$n new"; python tools/bench_lwl.py --n $n --iters 3 --reps 30 2>/dev/null | tee -a $OUT/lwl_new.jsonl
  echo "== lwl n=$n old"; PT_HOT_LIB=$OLD python tools/bench_lwl.py --n $n --iters 3 --reps 30 2>/dev/null | tee -a $OUT/lwl_old.jsonl
done
# per-kernel times of the head and of the LWL solve, new library
rocprofv3 --kernel-trace --stats -d $OUT/p_head -o k -- python tools/bench_head.py > /dev/null 2>&1
python tools/rocpd_summary.py $(find $OUT/p_head -name "*.db" | head -1) | cut -c1-160 | head -8 | tee $OUT/head_kernel_stats.csv
rm -rf $OUT/p_head
for n in 8 32; do
  rocprofv3 --kernel-trace --stats -d $OUT/p_lwl -o k -- python tools/bench_lwl.py --n $n --iters 3 --reps 10 > /dev/null 2>&1
  python tools/rocpd_summary.py $(find $OUT/p_lwl -name "*.db" | head -1) | cut -c1-160 | head -9 | tee $OUT/lwl_n${n}_kernel_stats.csv
  rm -rf $OUT/p_lwl
done
bash tools/pmc_head.sh r06a/head_pmc > /dev/null 2>&1; cat $OUT/head_pmc/pmc_head.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench.json
