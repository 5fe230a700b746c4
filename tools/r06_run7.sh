#!/bin/bash
# round 6, GPU call 7: graph-replayed one-call frame (pt_frame_full.dyn): parity vs the eager call, frame timings
export TMPDIR=/tmp
OUT=gpurun_out/r06g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_frame_full.py -x -q -k "graph" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_subset.log; tail -12 $OUT/pytest_subset.log | cut -c1-300
for rep in 1 2 3; do if [ $rep = 3 ]; then export PT_FRAME_DYN_COPY=1; echo "--- with the copy node"; fi
python tools/bench_dimp_frame_extended.py > $OUT/frame_extended_$rep.json 2> $OUT/frame_extended_$rep.err; python -c "
import json; d=json.loads(open('$OUT/frame_extended_$rep.json').read().strip().splitlines()[-1])
print({k:(v.get('us_per_frame', v.get('error')) if isinstance(v,dict) else v) for k,v in d.items() if k!='workload'})"; tail -2 $OUT/frame_extended_$rep.err
done
