"""One-off randomized check of the multi-filter passes (k_mf_corr / k_mf_adj / k_mf_corr1) against the float64 oracle on 160
random shapes.  python tests/stress_mf.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from pytracking_amd import filter as FL
from oracle import np_oracle as O
rng = np.random.default_rng(1234)
dev = torch.device('cuda', 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = 0; ran = 0; skipped = 0
for it in range(160):
    K = int(rng.choice([1, 3, 3, 3]))
    n = int(rng.integers(1, 9)); F = int(rng.integers(1, 17)); C = int(rng.integers(1, 70))
    H = int(rng.integers(1, 40)); W = int(rng.integers(1, 80))
    if K == 3 and (H < 1 or W < 1): continue
    feat = rng.standard_normal((n, C, H, W), dtype=np.float32)
    filt = (rng.standard_normal((F, C, K, K), dtype=np.float32) * 0.1)
    try:
        s = FL.apply_filter(T(feat)[:, None], T(filt)[None])[:, 0]
    except RuntimeError as e:
        skipped += 1; continue
    ref = O.apply_filter(feat.astype(np.float64), filt.astype(np.float64))
    err = float(np.abs(s.cpu().numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
    inp = rng.standard_normal(ref.shape).astype(np.float32)
    adj = FL.apply_feat_transpose(T(feat)[:, None], T(inp)[:, None], K)[0]
    refa = O.apply_feat_transpose(feat.astype(np.float64), inp.astype(np.float64), K)
    erra = float(np.abs(adj.cpu().numpy() - refa).max()) / max(1.0, float(np.abs(refa).max()))
    ran += 1
    if err > 2e-5 or erra > 2e-5 or not np.isfinite(err + erra):
        bad += 1; print('MISMATCH', (n, F, C, H, W, K), err, erra)
print('ran', ran, 'skipped', skipped, 'bad', bad)
