#!/bin/bash
# round 6, GPU call 6: the round's evidence run with the final library -- counters, full suite, bench (default + driver style x3), graph-mode
# kernel stats, the trackers on the device, the frame after the backbone, per-workload kernel stats, smoke
export TMPDIR=/tmp
OUT=gpurun_out/r06f
mkdir -p $OUT
bash tools/gpu_pmc4.sh r06f/pmc > $OUT/pmc_passes.log 2>&1; tail -3 $OUT/pmc_passes.log
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
PT_BENCH_KEEP_STATS=$OUT/kernel_stats.csv python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
for rep in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-other > $OUT/bench_steps20_$rep.json 2>> $OUT/bench.err
  python -c "import json,sys; d=json.loads(open('$OUT/bench_steps20_$rep.json').read().strip().splitlines()[-1]); print('driver-style', d['value'], d['ms_per_step'], d.get('repeats_us_per_frame'), d['roofline']['frac'], d['roofline'].get('frac_of_floor'))" | tee -a $OUT/driver_style_three_runs.txt
done
bash tools/prof_graph.sh $OUT dimp50 > /dev/null 2>&1; bash tools/prof_graph.sh $OUT prdimp50 > /dev/null 2>&1; head -12 $OUT/graph_kernel_stats_dimp50.csv
python -B tests/trackers_on_device.py --out $OUT/trackers_on_device.txt > /dev/null 2>&1; echo "trackers rc=$?"; grep -E "^== " $OUT/trackers_on_device.txt | cut -c1-120
python tools/bench_dimp_frame_extended.py > $OUT/frame_extended.json 2>> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/frame_extended.json').read().strip().splitlines()[-1])
print({k:(v.get('us_per_frame') if isinstance(v,dict) else v) for k,v in d.items() if k!='workload'})"
bash tools/prof_tomp.sh > $OUT/tomp_kernels.txt 2>&1; bash tools/prof_atom.sh 20 > $OUT/atom_kernels.txt 2>&1; bash tools/prof_iou.sh 24 > $OUT/iou_kernels.txt 2>&1
cp gpurun_out/atom_kernels.csv gpurun_out/iou_kernels.csv $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
