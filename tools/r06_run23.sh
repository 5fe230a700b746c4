#!/bin/bash
# round 6, GPU call 23+: box-to-box spread of the headline with the final library (each gpurun call is a fresh box)
export TMPDIR=/tmp
OUT=gpurun_out/r06v
mkdir -p $OUT
B="timeout 200 python bench.py --no-other --no-cpu-baseline --no-gpu-baseline --no-roofline"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
TAG=$(cat /proc/sys/kernel/random/boot_id | cut -c1-8)
echo "box $TAG  500 steps: $($B 2>/dev/null | val) | $($B 2>/dev/null | val)   driver style: $($B --steps 20 --warmup 5 2>/dev/null | val) | $($B --steps 20 --warmup 5 2>/dev/null | val) | $($B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/box_spread_$TAG.txt
