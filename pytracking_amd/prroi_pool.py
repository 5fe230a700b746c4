"""`PrRoIPool2D` -- drop-in for `ltr.external.PreciseRoIPooling.pytorch.prroi_pool.PrRoIPool2D`, the CUDA-only
extension the reference JIT-builds from an (empty here) git submodule.  Import sites:
`ltr/models/target_classifier/initializer.py:4,18,45`, `ltr/models/bbreg/atom_iou_net.py:4,31-32,41-42,126-127,157,160`.

forward(features (N,C,H,W) fp32, rois (R,5) = [batch_idx, x0, y0, x1, y1]) -> (R,C,PH,PW); differentiable
w.r.t. the features and w.r.t. the RoI coordinates (the trackers read `bb.grad`, pytracking/tracker/dimp/dimp.py:737-745).
"""
import torch
import torch.nn as nn

from . import _lib
from .filter import _ptr, _require_device, _stream, device_guarded


class _PrRoIPool2DFunction(torch.autograd.Function):
    @staticmethod
    @device_guarded
    def forward(ctx, features, rois, pooled_height, pooled_width, spatial_scale):
        _require_device(features, rois)
        features = features.contiguous()
        rois = rois.contiguous()
        N, C, H, W = features.shape
        R = rois.shape[0]
        assert rois.dim() == 2 and rois.shape[1] == 5
        out = torch.empty((R, C, pooled_height, pooled_width), dtype=torch.float32, device=features.device)
        rc = _lib.lib().pt_prroi_fwd_f32(_ptr(features), _ptr(rois), _ptr(out), N, C, H, W, R, pooled_height,
                                         pooled_width, float(spatial_scale), _stream())
        _lib.check(rc, "pt_prroi_fwd_f32")
        ctx.params = (pooled_height, pooled_width, float(spatial_scale))
        ctx.save_for_backward(features, rois)
        return out

    @staticmethod
    @device_guarded
    def backward(ctx, grad_out):
        features, rois = ctx.saved_tensors
        PH, PW, scale = ctx.params
        N, C, H, W = features.shape
        R = rois.shape[0]
        grad_out = grad_out.contiguous()
        L = _lib.lib()
        g_feat = g_rois = None
        if ctx.needs_input_grad[0]:
            g_feat = torch.zeros_like(features)
            rc = L.pt_prroi_bwd_feat_f32(_ptr(grad_out), _ptr(rois), _ptr(g_feat), N, C, H, W, R, PH, PW, scale, _stream())
            _lib.check(rc, "pt_prroi_bwd_feat_f32")
        if ctx.needs_input_grad[1]:
            g_rois = torch.empty_like(rois)
            rc = L.pt_prroi_bwd_coor_f32(_ptr(grad_out), _ptr(features), _ptr(rois), _ptr(g_rois), N, C, H, W, R, PH,
                                         PW, scale, _stream())
            _lib.check(rc, "pt_prroi_bwd_coor_f32")
        return g_feat, g_rois, None, None, None


def prroi_pool2d(features, rois, pooled_height, pooled_width, spatial_scale):
    return _PrRoIPool2DFunction.apply(features, rois, int(pooled_height), int(pooled_width), float(spatial_scale))


class PrRoIPool2D(nn.Module):
    def __init__(self, pooled_height, pooled_width, spatial_scale):
        super().__init__()
        self.pooled_height = int(pooled_height)
        self.pooled_width = int(pooled_width)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois):
        return prroi_pool2d(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale)

    def extra_repr(self):
        return 'kernel_size=({pooled_height}, {pooled_width}), spatial_scale={spatial_scale}'.format(**self.__dict__)
