#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r06o
mkdir -p $OUT
python tools/exp_first_launch.py 2>&1 | grep -v amdgpu.ids | tee $OUT/first_launch.txt
