"""Randomised-shape parity (`-m gpu`): the single-filter passes, the multi-filter passes and the DiMP steepest-descent
solver against the float64 oracle over seeded random sample counts / channel counts / map sizes / kernel sizes --
about 20 shapes each, sized for the oracle to finish in seconds.  `python tests/test_gpu_stress.py [N]` runs N times as
many shapes by hand (what tests/stress_*.py did in round 2)."""
import sys

import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from pytracking_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
f64 = lambda a: a.astype(np.float64)


def _rel(a, b):
    return float(np.abs(a.cpu().numpy() - b).max()) / max(1.0, float(np.abs(b).max()))


def single_filter_shapes(count, seed=4321):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        C = int(rng.choice([128, 256, 512, 1024, 64, 20, 33]))
        K = int(rng.choice([1, 2, 3, 4, 4, 4]))
        n = int(rng.integers(1, 60 if C <= 256 else 20))
        out.append((n, C, int(rng.integers(4, 26)), int(rng.integers(4, 26)), K, int(rng.integers(1 << 30))))
    return out


def multi_filter_shapes(count, seed=1234):
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(1, 9)), int(rng.integers(1, 17)), int(rng.integers(1, 70)), int(rng.integers(1, 40)),
             int(rng.integers(1, 80)), int(rng.choice([1, 3, 3, 3])), int(rng.integers(1 << 30))) for _ in range(count)]


def solver_shapes(count, seed=99):
    rng = np.random.default_rng(seed)
    out = []
    for it in range(count):
        n, C = int(rng.integers(1, 60)), int(rng.choice([128, 256, 512]))
        H = int(rng.choice([10, 12, 14, 16, 18, 20, 22]))
        if n * C * H * H > 9e6:
            n = max(1, int(9e6 / (C * H * H)))
        out.append((n, C, H, 1000 + it))
    return out


def check_single(n, C, H, W, K, seed):
    from pytracking_amd import filter as F
    rng = np.random.default_rng(seed)
    feat = rng.standard_normal((n, C, H, W), dtype=np.float32)
    filt = rng.standard_normal((C, K, K), dtype=np.float32) * np.float32(0.05)
    s = F.apply_filter(T(feat), T(filt[None]))[:, 0]
    ref = O.apply_filter(f64(feat), f64(filt))
    inp = rng.standard_normal(ref.shape).astype(np.float32)
    adj = F.apply_feat_transpose(T(feat), T(inp)[:, None], K, training=False)[0]
    return _rel(s, ref), _rel(adj, O.apply_feat_transpose(f64(feat), f64(inp), K))


def check_multi(n, Fn, C, H, W, K, seed):
    from pytracking_amd import filter as FL
    rng = np.random.default_rng(seed)
    feat = rng.standard_normal((n, C, H, W), dtype=np.float32)
    filt = rng.standard_normal((Fn, C, K, K), dtype=np.float32) * np.float32(0.1)
    s = FL.apply_filter(T(feat)[:, None], T(filt)[None])[:, 0]
    ref = O.apply_filter(f64(feat), f64(filt))
    inp = rng.standard_normal(ref.shape).astype(np.float32)
    adj = FL.apply_feat_transpose(T(feat)[:, None], T(inp)[:, None], K)[0]
    return _rel(s, ref), _rel(adj, O.apply_feat_transpose(f64(feat), f64(inp), K))


def check_solver(n, C, H, seed):
    import test_gpu_parity as TG
    from oracle.frame_port import _dimp_kwargs
    w0, feat, bb, sw = synth.dimp_problem(seed, n, small=dict(C=C, H=H, W=H))
    its, losses = TG._run(TG._dimp_module(), w0, feat, bb, sw, 3)
    ref_its, ref_l = O.dimp_sd(f64(w0), f64(feat), f64(bb), f64(sw), num_iter=3, **_dimp_kwargs(synth.DIMP50))
    e1 = float(np.abs(its.cpu().numpy() - ref_its).max())
    e2 = float(np.abs(losses.cpu().numpy() - np.array(ref_l)).max()) / max(1.0, float(np.abs(ref_l).max()))
    return e1, e2


def _covered(fn, shape):
    """A shape the kernels declare as not covered (the ABI's UNSUPPORTED error) is skipped, any other error fails."""
    try:
        return fn(*shape)
    except RuntimeError as exc:
        if "not covered" in str(exc) or "UNSUPPORTED" in str(exc).upper():
            pytest.skip(str(exc)[:80])
        raise


@pytest.mark.parametrize("shape", single_filter_shapes(20), ids=lambda s: "x".join(map(str, s[:5])))
def test_single_filter_passes_random_shapes(shape):
    """k_corr2 / k_adj2 (XCD-aligned) and the generic k_corr / k_adj, K in 1..4, ragged maps."""
    err, erra = _covered(check_single, shape)
    assert err <= 2e-5 and erra <= 5e-5, (shape, err, erra)


@pytest.mark.parametrize("shape", multi_filter_shapes(20), ids=lambda s: "x".join(map(str, s[:6])))
def test_multi_filter_passes_random_shapes(shape):
    """k_mf_corr / k_mf_adj / k_mf_corr1: 1..16 filters, 1..69 channels, maps down to 1 x 1."""
    err, erra = _covered(check_multi, shape)
    assert err <= 2e-5 and erra <= 2e-5, (shape, err, erra)


@pytest.mark.parametrize("shape", solver_shapes(20), ids=lambda s: "x".join(map(str, s[:3])))
def test_dimp_solver_random_shapes(shape):
    """3 iterations of DiMPSteepestDescentGN (fast path incl. the quad table): iterates and losses within 1e-4."""
    e1, e2 = check_solver(*shape)
    assert e1 <= 1e-4 and e2 <= 1e-4, (shape, e1, e2)


if __name__ == "__main__":
    sys.path.insert(0, "."); sys.path.insert(0, "tests")
    mult = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    for name, shapes, fn, tol in (("single", single_filter_shapes(20 * mult), check_single, (2e-5, 5e-5)),
                                  ("multi", multi_filter_shapes(20 * mult), check_multi, (2e-5, 2e-5)),
                                  ("solver", solver_shapes(7 * mult), check_solver, (1e-4, 1e-4))):
        bad = skipped = 0
        for sh in shapes:
            try:
                e = fn(*sh)
            except RuntimeError:
                skipped += 1
                continue
            if not (e[0] <= tol[0] and e[1] <= tol[1]):
                bad += 1
                print("MISMATCH", name, sh, e)
        print(name, "ran", len(shapes) - skipped, "skipped", skipped, "bad", bad)
