"""TEST INFRASTRUCTURE ONLY -- autograd restatement of Precise RoI Pooling (CPU, torch).

PARITY UNPINNED: the reference vendors PrRoIPool as the git submodule
`ltr/external/PreciseRoIPooling` (`/root/reference/.gitmodules:1-3`, github.com/vacancy/PreciseRoIPooling,
commit pin not recoverable) and the directory is EMPTY in the snapshot; the reference has no test or
golden vector for it.  This file restates the published definition (Jiang et al.,
"Acquisition of Localization Confidence for Accurate Object Detection", ECCV 2018, eq. 4-6;
SURVEY.md Appendix A): the exact integral of the bilinear interpolant (zero outside the map) over
each bin divided by the bin area.  It is self-pinned by tests/test_oracle_prroi.py (brute-force
quadrature, analytic cases, fp64 gradcheck).  Call sites it must serve:
`ltr/models/target_classifier/initializer.py:18,45`, `ltr/models/bbreg/atom_iou_net.py:31-32,41-42,126-127,157,160`.

It is written with differentiable torch ops so `torch.autograd` supplies the feature and the RoI
coordinate gradients that the analytic numpy oracle (oracle/np_oracle.py) and the HIP kernels are
checked against.
"""
import torch
import torch.nn as nn


def _hat_cdf(u):
    """G(u) = integral_{-inf}^{u} max(0, 1-|t|) dt  (piecewise quadratic, C1)."""
    t = u.clamp(-1.0, 1.0)
    return torch.where(t <= 0, 0.5 * (t + 1.0) ** 2, 1.0 - 0.5 * (1.0 - t) ** 2)


def prroi_pool2d(features, rois, pooled_height, pooled_width, spatial_scale):
    """features (N,C,H,W); rois (R,5) = [batch_idx, x0, y0, x1, y1] in image coords -> (R,C,PH,PW)."""
    N, C, H, W = features.shape
    R = rois.shape[0]
    PH, PW = int(pooled_height), int(pooled_width)
    dt = features.dtype
    b = rois[:, 0].long()
    x0 = rois[:, 1] * spatial_scale
    y0 = rois[:, 2] * spatial_scale
    x1 = rois[:, 3] * spatial_scale
    y1 = rois[:, 4] * spatial_scale
    bw = (x1 - x0).clamp(min=0) / PW
    bh = (y1 - y0).clamp(min=0) / PH
    q = torch.arange(PW, dtype=dt, device=features.device)
    p = torch.arange(PH, dtype=dt, device=features.device)
    xs = x0[:, None] + q[None, :] * bw[:, None]            # (R,PW)
    xe = xs + bw[:, None]
    ys = y0[:, None] + p[None, :] * bh[:, None]            # (R,PH)
    ye = ys + bh[:, None]
    ii = torch.arange(W, dtype=dt, device=features.device)
    jj = torch.arange(H, dtype=dt, device=features.device)
    wx = _hat_cdf(xe[:, :, None] - ii) - _hat_cdf(xs[:, :, None] - ii)   # (R,PW,W)
    wy = _hat_cdf(ye[:, :, None] - jj) - _hat_cdf(ys[:, :, None] - jj)   # (R,PH,H)
    area = bw * bh                                                       # (R,)
    feat_r = features[b]                                                 # (R,C,H,W)
    integ = torch.einsum("rcji,rpj,rqi->rcpq", feat_r, wy, wx)
    safe = torch.where(area > 0, area, torch.ones_like(area))
    out = integ / safe[:, None, None, None]
    return torch.where((area > 0)[:, None, None, None], out, torch.zeros_like(out))


class PrRoIPool2D(nn.Module):
    """Same constructor/forward signature as the upstream module the reference imports."""

    def __init__(self, pooled_height, pooled_width, spatial_scale):
        super().__init__()
        self.pooled_height = int(pooled_height)
        self.pooled_width = int(pooled_width)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois):
        return prroi_pool2d(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale)
