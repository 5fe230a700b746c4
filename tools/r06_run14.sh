#!/bin/bash
# round 6, GPU call 14: does the closing synchronize of the 2 ms driver-style region cost a blocking-wait wake-up?  (PT_BENCH_SPIN)
export TMPDIR=/tmp
OUT=gpurun_out/r06o
mkdir -p $OUT
B="python bench.py --no-other --no-cpu-baseline --no-gpu-baseline --no-roofline"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3 4; do
  echo "block drv : $($B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/spin_ab.txt
  echo "spin  drv : $(PT_BENCH_SPIN=1 $B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/spin_ab.txt
done
for rep in 1 2; do
  echo "block 500 : $($B 2>/dev/null | val)" | tee -a $OUT/spin_ab.txt
  echo "spin  500 : $(PT_BENCH_SPIN=1 $B 2>/dev/null | val)" | tee -a $OUT/spin_ab.txt
done
# where the 20-frame region's time goes on the host side: graph launch call, completion
python - <<'PY' 2>&1 | tee $OUT/region_anatomy.txt
import time, torch, sys
sys.path.insert(0, '.')
import bench
from pytracking_amd import bench_frame, synth
dev = torch.device('cuda', 0)
cfg = synth.DIMP50
st = bench_frame.TrackState(cfg, 50, seed=1234, device=dev, kind='dimp')
pool = bench.make_pool(cfg, 4321, dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    bench.run_frames(st, pool, 0, 2); stream.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        bench.run_frames(st, pool, 5, 20)
    for _ in range(30): g.replay()
    stream.synchronize()
    for mode in ('block', 'spin', 'block', 'spin'):
        rows = []
        for rep in range(12):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            g.replay()
            t1 = time.perf_counter()
            e1.record(stream)
            if mode == 'spin':
                while not e1.query():
                    pass
            stream.synchronize(); torch.cuda.synchronize()
            t2 = time.perf_counter()
            rows.append((1e6 * (t1 - t0), 1e6 * (t2 - t0), 1e3 * e0.elapsed_time(e1)))
        rows = rows[2:]
        import statistics as S
        print(mode, 'launch call us %.1f  region wall us %.1f  device (event pair) us %.1f  -> wall - device %.1f' % (
            S.median(r[0] for r in rows), S.median(r[1] for r in rows), S.median(r[2] for r in rows), S.median(r[1] - r[2] for r in rows)))
PY
