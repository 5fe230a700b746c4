"""ToMP per-frame model prediction timing at BASELINE configs[3] geometry (tomp.py:282-303): 2 memory frames + the test
frame of 256 x 18 x 18 head features -> 972 tokens x 2 batch rows, 6 encoder + 6 decoder layers (8 heads, FFN 2048),
classifier + dense box regressor.   python tools/bench_tomp.py [--reps 50] [--graph]
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
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--graph", action="store_true", help="replay the frame from a captured HIP graph")
    a = ap.parse_args()
    if _lib.needs_build():
        _lib.build_library()
    print(json.dumps(workloads.tomp(torch.device("cuda", 0), a.reps, a.graph)))
