"""Experiment: period of the two solver passes and of the whole frame against the number of memory samples n.
n = 32 puts one k_corr2 workgroup on every CU, n = 64 two; n = 50 (the benchmark) leaves 144 CUs with two and 112 with
one -- how much of the pass time is that imbalance?   python tools/exp_pass_vs_n.py [n ...]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from pytracking_amd import bench_frame, synth  # noqa: E402


def main():
    ns = [int(a) for a in sys.argv[1:]] or [24, 32, 40, 48, 50, 56, 64]
    kind = os.environ.get("PT_EXP_KIND", "dimp")                 # "prdimp": BASELINE configs[2]'s per-GPU workload
    cfg = synth.PRDIMP50 if kind == "prdimp" else synth.DIMP50
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream()
    pool = bench.make_pool(cfg, 99, dev)
    out = []
    with torch.cuda.stream(stream):
        for n in ns:
            st = bench_frame.TrackState(cfg, n, seed=1234, device=dev, kind=kind)
            bench.run_frames(st, pool, 0, 10)
            stream.synchronize()
            corr = min(bench.event_period_us(st, stream, 0) for _ in range(3))
            adj = min(bench.event_period_us(st, stream, 1) for _ in range(3))
            # whole frames, one graph of 20
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                bench.run_frames(st, pool, 10, 20)
            g.replay()
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(10):
                g.replay()
            e1.record(stream)
            e1.synchronize()
            frame = e0.elapsed_time(e1) * 1e3 / 200
            rec = {"kind": kind, "n": n, "corr_period_us": round(corr, 2), "adj_period_us": round(adj, 2), "frame_us": round(frame, 2),
                   "MB_per_pass": round(4e-6 * n * cfg["C"] * cfg["H"] * cfg["W"], 2)}
            print(json.dumps(rec), flush=True)
            out.append(rec)


if __name__ == "__main__":
    main()
