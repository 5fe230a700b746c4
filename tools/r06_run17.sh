#!/bin/bash
# round 6, GPU call 17: where does the fused init stage lose?  kernel durations of the eager chain, fused vs unfused
export TMPDIR=/tmp
OUT=gpurun_out/r06p
mkdir -p $OUT
for mode in fused unfused; do
  [ $mode = unfused ] && export PT_SD_FUSE_INIT=0
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$mode -o k -- python bench.py --profile-child --profile-mode eager --steps 60 --warmup 10 > /dev/null 2> $OUT/prof_$mode.err
  python tools/rocpd_summary.py $(find $OUT/prof_$mode -name "*.db" | head -1) | python tools/short_stats.py 9 > $OUT/kernels_$mode.txt
  echo "== $mode"; cat $OUT/kernels_$mode.txt
  rm -rf $OUT/prof_$mode
done
