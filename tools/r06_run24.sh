#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r06w
mkdir -p $OUT
timeout 300 python tools/exp_tomp_two_streams.py 2>&1 | grep -v amdgpu.ids | tee $OUT/tomp_two_streams.txt
