"""ATOM online filter update timing (BASELINE configs[0] shape): ConjugateGradient on ConvProblem, n=250 samples of
64x18x18, 4x4 filter, 5 CG iterations (pytracking/parameter/atom/default.py).   python tools/bench_atom.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import _lib, synth  # noqa: E402
from pytracking_amd.optimization import ConjugateGradient, ConvProblem, FactorizedConvProblem, GaussNewtonCG, MLU  # noqa: E402


def main():
    if _lib.needs_build():
        _lib.build_library()
    dev = "cuda:0"
    c = synth.ATOM18
    n = c["memory"]
    x0, samples, y, sw = synth.atom_problem(1, n)
    T = lambda a: torch.from_numpy(a).to(dev)
    x = [T(x0.copy())[None].clone()]
    prob = ConvProblem([T(samples)], [T(y)[:, None]], [c["filter_reg"]], [T(sw)], MLU(c["act_min_val"]))
    opt = ConjugateGradient(prob, x, fletcher_reeves=False, direction_forget_factor=0)
    for _ in range(3):
        opt.run(c["cg_iter"])
    torch.cuda.synchronize()
    reps = 100
    t0 = time.perf_counter()
    for _ in range(reps):
        opt.run(c["cg_iter"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    passes = 2 * c["cg_iter"] + 2
    byts = passes * 4.0 * n * c["C"] * c["H"] * c["W"]
    print(json.dumps({"workload": f"ATOM ConvProblem CG n={n} C=64 18x18 K=4, {c['cg_iter']} iterations", "us_per_update": round(dt * 1e6, 1),
                      "updates_per_s": round(1 / dt, 1), "passes": passes, "feature_GBs": round(byts / dt / 1e9, 1)}))

    # first-frame joint optimisation (atom.py:156-176): 30 augmented samples x 256 x 18 x 18, 64 compressed channels,
    # init_CG_iter 60 / init_GN_iter 6
    import numpy as np
    rng = np.random.default_rng(5)
    na, M, Kc, K = 30, 256, 64, 4
    raw = T(rng.standard_normal((na, M, 18, 18), dtype=np.float32) * np.float32(0.1))
    _, _, y2, sw2 = synth.atom_problem(5, na)
    P0 = T(rng.standard_normal((Kc, M, 1, 1), dtype=np.float32) * np.float32(1.0 / np.sqrt(M)))
    jp = FactorizedConvProblem([raw], [T(y2)[:, None]], [c["filter_reg"]], [1e-4], None, [T(sw2)], None,
                               MLU(c["act_min_val"]))

    def joint():
        var = [torch.zeros(1, Kc, K, K, device=dev), P0.clone()]
        GaussNewtonCG(jp, var).run(10, 6)
        return var

    joint()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        joint()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(json.dumps({"workload": "ATOM FactorizedConvProblem GaussNewtonCG n=30 M=256 Kc=64 18x18 K=4, 6 x 10 CG",
                      "ms_per_first_frame_solve": round(dt * 1e3, 3)}))


if __name__ == "__main__":
    main()
