"""LWL few-shot learner timing at BASELINE configs[4] geometry: 16 filters 3x3, 512 channels, 30x52 maps, n samples,
num_iter GN steepest-descent iterations (reference uses 3 per frame, 20 at init; BASELINE.json quotes 4).
    python tools/bench_lwl.py [--n 32] [--iters 3] [--reps 50]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import _lib  # noqa: E402
from tools import workloads  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    if _lib.needs_build():
        _lib.build_library()
    print(json.dumps(workloads.lwl(torch.device("cuda", 0), a.n, a.iters, a.reps)))
