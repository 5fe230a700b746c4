#!/bin/bash
# round 6, GPU call 22: k_attn exponentials issued one key tile ahead, between the PV MFMAs: parity + A/B
export TMPDIR=/tmp
OUT=gpurun_out/r06u
mkdir -p $OUT
V=$PWD/pytracking_amd/variants
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tomp" > $OUT/pytest_tomp.log 2>&1; echo "tomp tests rc=$?" | tee -a $OUT/pytest_tomp.log; tail -3 $OUT/pytest_tomp.log
ms() { python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms'])"; }
for rep in 1 2 3 4; do
  echo "exp one tile ahead : $(timeout 120 python tools/bench_tomp.py --graph --reps 200 2>/dev/null | ms)" | tee -a $OUT/attn_pipe_ab.txt
  echo "exp in front       : $(PT_HOT_LIB=$V/libpt_hot_attnpipe0.so timeout 120 python tools/bench_tomp.py --graph --reps 200 2>/dev/null | ms)" | tee -a $OUT/attn_pipe_ab.txt
done
