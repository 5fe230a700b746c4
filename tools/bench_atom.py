"""ATOM timing: the online filter update (BASELINE configs[0] shape: ConjugateGradient on ConvProblem, n=250 samples of
64x18x18, 4x4 filter, 5 CG iterations, pytracking/parameter/atom/default.py) and the first-frame joint optimisation.
    python tools/bench_atom.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import _lib  # noqa: E402
from tools import workloads  # noqa: E402

if __name__ == "__main__":
    if _lib.needs_build():
        _lib.build_library()
    dev = torch.device("cuda", 0)
    print(json.dumps(workloads.atom_cg(dev, reps=100)))
    print(json.dumps(workloads.atom_first_frame(dev, reps=10)))
