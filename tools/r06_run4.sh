#!/bin/bash
# round 6, GPU call 4: qkv projection on 64x64 pinned-schedule tiles (A/B), LDS zero fill behind the first loads (A/B), installed track()
# breakdown, the configs[2] per-GPU workload under the launcher with nccl (1 rank)
export TMPDIR=/tmp
OUT=gpurun_out/r06d
mkdir -p $OUT
V=$PWD/pytracking_amd/variants
timeout 900 python -m pytest tests -m gpu -x -q -k "tomp or head or lwl or mf or clf or smoke or multi_filter" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_subset.log; tail -3 $OUT/pytest_subset.log
for rep in 1 2 3; do
  echo "tomp qkv 64x64 ps : $(python tools/bench_tomp.py --graph 2>/dev/null | tail -1 | cut -c120-260)" | tee -a $OUT/tomp_qkv_ab.txt
  echo "tomp qkv 32x32    : $(PT_GEMM_PS_QKV=0 python tools/bench_tomp.py --graph 2>/dev/null | tail -1 | cut -c120-260)" | tee -a $OUT/tomp_qkv_ab.txt
done
for rep in 1 2; do
  echo "head new fill: $(python tools/bench_head.py 2>/dev/null | head -1 | cut -c1-60)" | tee -a $OUT/zerofill_ab.txt
  echo "head old fill: $(PT_HOT_LIB=$V/libpt_hot_mfzf0.so python tools/bench_head.py 2>/dev/null | head -1 | cut -c1-60)" | tee -a $OUT/zerofill_ab.txt
  for n in 1 8; do
    echo "lwl n=$n new fill: $(python tools/bench_lwl.py --n $n --iters 3 --reps 40 2>/dev/null | cut -c85-130)" | tee -a $OUT/zerofill_ab.txt
    echo "lwl n=$n old fill: $(PT_HOT_LIB=$V/libpt_hot_mfzf0.so python tools/bench_lwl.py --n $n --iters 3 --reps 40 2>/dev/null | cut -c85-130)" | tee -a $OUT/zerofill_ab.txt
  done
done
rocprofv3 --kernel-trace --stats -d $OUT/p_tomp -o k -- python tools/bench_tomp.py --reps 10 > /dev/null 2>&1
python tools/rocpd_summary.py $(find $OUT/p_tomp -name "*.db" | head -1) | cut -c1-170 | head -12 | tee $OUT/tomp_kernel_stats.csv
rm -rf $OUT/p_tomp
# installed track(): device time per call from a kernel trace of the plain loop (2 passes x 40 frames + 2 initialize), then the breakdown
rocprofv3 --kernel-trace --stats -d $OUT/p_trk -o k -- python -B tests/profile_installed_track.py --frames 40 --plain > $OUT/track_plain.log 2>&1
DB=$(find $OUT/p_trk -name "*.db" | head -1)
python tools/rocpd_summary.py $DB | cut -c1-150 | head -30 > $OUT/track_kernel_stats.csv
python - "$DB" > $OUT/track_device_time.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tot, n = c.execute("select sum(duration), count(*) from kernels").fetchone()
print(f"all kernels of the run (2 x initialize + 2 x 40 track): {tot/1e3:.1f} us in {n} launches")
PY
cat $OUT/track_device_time.txt
rm -rf $OUT/p_trk
python -B tests/profile_installed_track.py --frames 40 --out $OUT/installed_track_breakdown.txt > /dev/null 2> $OUT/track_profile.err; tail -3 $OUT/track_profile.err; head -12 $OUT/installed_track_breakdown.txt
# configs[2]'s per-GPU workload as the headline under the launcher (1 rank, nccl)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload prdimp50 --steps 20 --warmup 5 --no-other > $OUT/torchrun_prdimp50.json 2> $OUT/torchrun_prdimp50.err; echo "torchrun rc=$?"; tail -1 $OUT/torchrun_prdimp50.json | cut -c1-400
