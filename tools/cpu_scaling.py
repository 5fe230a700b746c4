"""Thread scaling of the CPU baseline (bench.py `cpu_baseline`: the reference's CPU execution path, oracle/frame_port.
TorchCpuTracker) on this host: frames/s of the headline workload at 1 / 8 / 32 / 64 / 128 / all threads.  Justifies (or
refutes) the 32-thread pin of bench.py.   python tools/cpu_scaling.py > profiles/rNN_cpu_thread_scaling.json"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.frame_port import TorchCpuTracker  # noqa: E402  (baseline leg: the checker timed as the CPU path)
from pytracking_amd import synth  # noqa: E402

if __name__ == "__main__":
    cfg, n = synth.DIMP50, 50
    host = os.cpu_count() or 1
    pool = torch.from_numpy(synth.clf_features(np.random.default_rng(99), 8, cfg["C"], cfg["H"], cfg["W"], cfg["K"]))
    rows = []
    for th in [t for t in (1, 8, 16, 32, 64, 128, 256) if t <= host]:
        tr = TorchCpuTracker(cfg, n, seed=1234, threads=th)
        tr.step(pool[0], 0, 5)
        t0, frames = time.perf_counter(), 0
        while time.perf_counter() - t0 < (4.0 if th > 1 else 6.0) or frames < 3:
            tr.step(pool[frames % 8], frames % n, 5)
            frames += 1
        dt = time.perf_counter() - t0
        rows.append({"threads": th, "frames_per_s": round(frames / dt, 3), "frames": frames})
        print(rows[-1], file=sys.stderr)
    print(json.dumps({"host_cores": host, "workload": "DiMP-50 frame, n=50x512x18x18, 5 SD iterations, torch-CPU port of the "
                      "reference path (fp32)", "scaling": rows}))
