"""Experiment (round 6, N): what does the FIRST launch of a captured 20-frame graph executable cost against the later ones?
bench.py's driver-style region (--steps 20 --warmup 5) launches its timed graph for the first time inside the timed bracket."""
import statistics as S
import sys
import time

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from pytracking_amd import bench_frame, synth  # noqa: E402

dev = torch.device('cuda', 0)
cfg = synth.DIMP50
st = bench_frame.TrackState(cfg, 50, seed=1234, device=dev, kind='dimp')
pool = bench.make_pool(cfg, 4321, dev)
stream = torch.cuda.Stream(device=dev)


def capture(first, count):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        bench.run_frames(st, pool, first, count)
    return g


with torch.cuda.stream(stream):
    bench.run_frames(st, pool, 0, 2)
    stream.synchronize()
    warm = capture(0, 5)
    for _ in range(3):
        warm.replay()
    stream.synchronize()
    rows = {k: [] for k in range(4)}
    for trial in range(8):
        g = capture(5, 20)
        for _ in range(12):                                    # the clock warm-up of bench.py (idempotent passes), shortened
            bench.event_period_us(st, stream, 0)
        for k in range(4):
            warm.replay()
            stream.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            stream.synchronize()
            torch.cuda.synchronize()
            rows[k].append(1e6 * (time.perf_counter() - t0) / 20)
        del g
    for k in range(4):
        print('launch #%d of a fresh 20-frame executable: median %.2f us per frame (min %.2f, max %.2f)' % (
            k + 1, S.median(rows[k]), min(rows[k]), max(rows[k])))
