"""One-off randomized check of the DiMP steepest-descent solver (fast path incl. the k_adj2 quad table) against the float64
oracle over random sample counts / map sizes / channel counts.  python tests/stress_solver.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_parity as TG
from pytracking_amd import synth
from oracle import np_oracle as O
rng = np.random.default_rng(99)
f64 = lambda a: a.astype(np.float64)
bad = ran = 0
for it in range(40):
    n = int(rng.integers(1, 60)); C = int(rng.choice([128, 256, 512])); H = int(rng.choice([10, 12, 14, 16, 18, 20, 22]))
    if n * C * H * H > 9e6: n = max(1, int(9e6 / (C * H * H)))
    w0, feat, bb, sw = synth.dimp_problem(1000 + it, n, small=dict(C=C, H=H, W=H))
    mod = TG._dimp_module()
    its, losses = TG._run(mod, w0, feat, bb, sw, 3)
    c = synth.DIMP50
    ref_its, ref_l = O.dimp_sd(f64(w0), f64(feat), f64(bb), f64(sw), num_iter=3, step_length=c["init_step_length"],
                               filter_reg=c["init_filter_reg"], min_filter_reg=c["min_filter_reg"], feat_stride=c["feat_stride"],
                               label_w=synth.gauss_lut(c["num_dist_bins"], c["bin_displacement"], c["init_gauss_sigma"]),
                               mask_w=synth.mask_lut(c["num_dist_bins"], c["bin_displacement"], c["mask_init_factor"]),
                               spatial_w=np.ones(c["num_dist_bins"], np.float32), bin_displacement=c["bin_displacement"])
    e1 = float(np.abs(its.cpu().numpy() - ref_its).max()); e2 = float(np.abs(losses.cpu().numpy() - np.array(ref_l)).max())
    ran += 1
    if e1 > 1e-4 or e2 > 1e-4 * max(1.0, float(np.abs(ref_l).max())) or not np.isfinite(e1 + e2):
        bad += 1; print('MISMATCH', (n, C, H), e1, e2)
print('ran', ran, 'bad', bad)
