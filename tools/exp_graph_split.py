"""Experiment (round 6): is the ~40 us a single 20-frame graph replay costs over 20 frames of a long run (107.9 vs 105.9 us per frame) the
graph's submission latency?  Replay the same 20 frames as ONE graph and as a short head graph + the rest, region timed like bench.py
(synchronise, t0, replay(s), synchronise).    python tools/exp_graph_split.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pytracking_amd import bench_frame, synth  # noqa: E402


def main():
    cfg, dev = synth.DIMP50, torch.device("cuda:0")
    stream = torch.cuda.Stream()
    pool = bench.make_pool(cfg, 99, dev)
    out = {}
    with torch.cuda.stream(stream):
        st = bench_frame.TrackState(cfg, 50, seed=1234, device=dev)
        bench.run_frames(st, pool, 0, 5)
        stream.synchronize()

        def cap(first, count, flush=True):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for f in range(first, first + count):
                    st.step(pool[f % bench.POOL], slot=f % st.n, num_iter=5, defer=True)
                if flush:
                    st.flush()
            return g
        plans = {"one_graph_20": [cap(5, 20)]}
        for head in (1, 2, 4):
            plans[f"head_{head}_plus_{20 - head}"] = [cap(5, head, flush=False), cap(5 + head, 20 - head)]
        plans["four_graphs_5"] = [cap(5, 5, False), cap(10, 5, False), cap(15, 5, False), cap(20, 5)]
        for name, gs in plans.items():
            ts = []
            for rep in range(12):
                stream.synchronize()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for g in gs:
                    g.replay()
                stream.synchronize()
                ts.append(1e6 * (time.perf_counter() - t0) / 20)
            ts = sorted(ts[2:])
            out[name] = {"us_per_frame_median": round(ts[len(ts) // 2], 2), "min": round(ts[0], 2), "max": round(ts[-1], 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
