"""Per-kernel durations of the benchmark frame as the hipGraph replays it (bench.py's own profile leg launches eagerly):
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o g -- python tools/prof_graph_frames.py [workload]
20 frames per graph, 10 replays."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from pytracking_amd import bench_frame, synth  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "dimp50"
    cfg = synth.DIMP50 if kind == "dimp50" else synth.PRDIMP50
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream()
    pool = bench.make_pool(cfg, 99, dev)
    with torch.cuda.stream(stream):
        st = bench_frame.TrackState(cfg, 50, seed=1234, device=dev, kind="dimp" if kind == "dimp50" else "prdimp")
        bench.run_frames(st, pool, 0, 5)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            bench.run_frames(st, pool, 5, 20)
        for _ in range(10):
            g.replay()
        stream.synchronize()


if __name__ == "__main__":
    main()
