"""LWL few-shot learner timing at BASELINE configs[4] geometry: 16 filters 3x3, 512 channels, 30x52 maps, n samples,
num_iter GN steepest-descent iterations (reference uses 3 per frame, 20 at init; BASELINE.json quotes 4).
    python tools/bench_lwl.py [--n 32] [--iters 3] [--reps 50]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import _lib, synth  # noqa: E402
from pytracking_amd.steepestdescent import GNSteepestDescent, LWTLResidual  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    if _lib.needs_build():
        _lib.build_library()
    dev = torch.device("cuda", 0)
    F, C, H, W, K = 16, 512, 30, 52, 3
    rng = np.random.default_rng(3)
    feat = torch.from_numpy(synth.clf_features(rng, a.n, C, H, W, K)).to(dev)[:, None]
    label = torch.from_numpy(rng.uniform(0, 1, (a.n, 1, F, H, W)).astype(np.float32)).to(dev)
    sw = torch.from_numpy(rng.uniform(0.2, 1, (a.n, 1, F, H, W)).astype(np.float32)).to(dev)
    w0 = torch.zeros(1, F, C, K, K, device=dev)
    opt = GNSteepestDescent(LWTLResidual(0.05).to(dev), num_iter=a.iters, compute_losses=False, residual_batch_dim=1)
    with torch.no_grad():
        for _ in range(3):
            opt(w0, feat=feat, label=label, sample_weight=sw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            opt(w0, feat=feat, label=label, sample_weight=sw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
    passes = 2 * a.iters + 1
    flops = passes * 2.0 * a.n * F * C * K * K * H * W
    byts = passes * 4.0 * a.n * C * H * W
    print(json.dumps({"workload": f"LWL GN-SD n={a.n} F=16 C=512 30x52 K=3, {a.iters} iterations", "ms_per_solve": round(dt * 1e3, 4),
                      "solves_per_s": round(1 / dt, 1), "passes": passes, "TFLOPs": round(flops / dt / 1e12, 2),
                      "frac_of_157_TFLOPs_f32_mfma": round(flops / dt / 157.3e12, 3), "feature_GBs": round(byts / dt / 1e9, 1)}))


if __name__ == "__main__":
    main()
