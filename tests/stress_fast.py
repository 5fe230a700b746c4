"""One-off randomized check of the single-filter passes (k_corr2 / k_adj2 and the generic k_corr / k_adj) against the float64
oracle on many shapes.  python tests/stress_fast.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from pytracking_amd import filter as F
from oracle import np_oracle as O
rng = np.random.default_rng(4321)
dev = torch.device('cuda', 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = ran = skipped = 0
for it in range(120):
    C = int(rng.choice([128, 256, 512, 1024, 64, 20, 33]))
    K = int(rng.choice([1, 2, 3, 4, 4, 4]))
    n = int(rng.integers(1, 60 if C <= 256 else 20)); H = int(rng.integers(4, 26)); W = int(rng.integers(4, 26))
    feat = rng.standard_normal((n, C, H, W), dtype=np.float32)
    filt = (rng.standard_normal((C, K, K), dtype=np.float32) * 0.05)
    try:
        s = F.apply_filter(T(feat), T(filt[None]))[:, 0]
    except RuntimeError:
        skipped += 1; continue
    ref = O.apply_filter(feat.astype(np.float64), filt.astype(np.float64))
    err = float(np.abs(s.cpu().numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
    inp = rng.standard_normal(ref.shape).astype(np.float32)
    adj = F.apply_feat_transpose(T(feat), T(inp)[:, None], K, training=False)[0]
    refa = O.apply_feat_transpose(feat.astype(np.float64), inp.astype(np.float64), K)
    erra = float(np.abs(adj.cpu().numpy() - refa).max()) / max(1.0, float(np.abs(refa).max()))
    ran += 1
    if err > 2e-5 or erra > 5e-5 or not np.isfinite(err + erra):
        bad += 1; print('MISMATCH', (n, C, H, W, K), err, erra)
print('ran', ran, 'skipped', skipped, 'bad', bad)
