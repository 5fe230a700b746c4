#!/bin/bash
# round 6, GPU call 2: frame chain (deferred last update) parity + A/B, k_fast_sgq2 ablations, position-complete correlation probe
export TMPDIR=/tmp
OUT=gpurun_out/r06b
mkdir -p $OUT
V=$PWD/pytracking_amd/variants
timeout 900 python -m pytest tests/test_gpu_frame_full.py tests/test_gpu_parity.py -x -q -k "chain or oracle or refuses or multi_sequence or closed_loop" > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_new.log; tail -4 $OUT/pytest_new.log
B="python bench.py --no-other --no-cpu-baseline --no-gpu-baseline --no-roofline"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "chain  driver-style: $($B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/chain_ab.txt
  echo "plain  driver-style: $(PT_BENCH_NO_CHAIN=1 $B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/chain_ab.txt
  echo "chain  500 steps   : $($B 2>/dev/null | val)" | tee -a $OUT/chain_ab.txt
  echo "plain  500 steps   : $(PT_BENCH_NO_CHAIN=1 $B 2>/dev/null | val)" | tee -a $OUT/chain_ab.txt
done
for v in 1 2 3 4; do
  for rep in 1 2; do
    echo "sgq_exp=$v plain 500 steps: $(PT_BENCH_NO_CHAIN=1 PT_HOT_LIB=$V/libpt_hot_sgq$v.so $B 2>/dev/null | val)" | tee -a $OUT/sgq_ablation.txt
  done
done
./experiments/corr_position_tile_probe 50 | tee $OUT/probe.jsonl
rocprofv3 --kernel-trace --stats -d $OUT/p_probe -o k -- ./experiments/corr_position_tile_probe 50 > /dev/null 2>&1
python tools/rocpd_summary.py $(find $OUT/p_probe -name "*.db" | head -1) | cut -c1-150 | tee $OUT/probe_kernel_stats.csv
rm -rf $OUT/p_probe
