"""Host-side mirror of the classification-feature head: `ltr/models/target_classifier/features.py:49-73`
(`residual_bottleneck`) with `ltr/models/layers/normalization.py:7-20` (`InstanceL2Norm`), for the configuration every
tracker of the reference instantiates -- no residual blocks, a final 3x3 convolution without bias, instance L2
normalisation (dimpnet.py:169-172, tompnet.py:99-102): `nn.Sequential(Conv2d, InstanceL2Norm)`, state_dict key `0.weight`.

The returned module is an `nn.Sequential` with those two children (so checkpoints load unchanged) whose forward runs both
as one fused call of the C ABI (`pt_clf_head_f32`).  Inference only; other configurations raise NotImplementedError.
"""
import torch
import torch.nn as nn

from . import _lib
from .filter import _ptr, _require_device, _stream, device_guarded, workspace


class InstanceL2Norm(nn.Module):
    """Parameter-free holder with the reference's constructor; executed inside `ClfHead`."""

    def __init__(self, size_average=True, eps=1e-5, scale=1.0):
        super().__init__()
        self.size_average = size_average
        self.eps = eps
        self.scale = scale

    def forward(self, input):
        raise NotImplementedError("InstanceL2Norm runs fused with the preceding convolution (ClfHead)")


class ClfHead(nn.Sequential):
    """Conv2d(Cin, Cout, 3, padding=1, bias=False) -> InstanceL2Norm(size_average=True) as one device call."""

    def __init__(self, conv, norm):
        super().__init__(conv, norm)
        self._key, self._wt = None, None

    def _weight_tap_major(self):
        w = self[0].weight
        key = (w.data_ptr(), w._version)
        if key != self._key:
            with torch.no_grad():
                self._wt = w.detach().permute(0, 2, 3, 1).contiguous()      # (Cout, ky, kx, Cin)
            self._key = key
        return self._wt

    @device_guarded
    def forward(self, x):
        if self.training or (torch.is_grad_enabled() and (x.requires_grad or self[0].weight.requires_grad)):
            raise NotImplementedError("ClfHead: the gfx950 path is inference only (eval() under torch.no_grad())")
        _require_device(x)
        lead = x.shape[:-3]
        x4 = x.reshape(-1, *x.shape[-3:]).contiguous()
        n, Cin, H, W = x4.shape
        conv, norm = self[0], self[1]
        Cout = conv.out_channels
        L = _lib.lib()
        nb = L.pt_clf_head_ws_bytes(n, Cin, Cout, H, W)
        if nb == 0:
            raise NotImplementedError("ClfHead: configuration not covered by the gfx950 kernels")
        ws = workspace(nb, x4.device)
        out = torch.empty(n, Cout, H, W, dtype=torch.float32, device=x4.device)
        rc = L.pt_clf_head_f32(_ptr(x4), _ptr(self._weight_tap_major()), _ptr(out), n, Cin, Cout, H, W, float(norm.scale),
                               float(norm.eps), _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "pt_clf_head_f32")
        return out.reshape(*lead, Cout, H, W)


def covered(feature_dim=256, num_blocks=1, l2norm=True, final_conv=False, norm_scale=1.0, out_dim=None, interp_cat=False,
            final_relu=False, final_pool=False, input_dim=None, final_stride=1):
    """True for the argument sets `residual_bottleneck` turns into Conv2d(3x3, no bias) + InstanceL2Norm."""
    cin = 4 * feature_dim if input_dim is None else input_dim
    cout = feature_dim if out_dim is None else out_dim
    return (num_blocks == 0 and final_conv and l2norm and not interp_cat and not final_relu and not final_pool
            and final_stride == 1 and cin % 64 == 0 and cout % 4 == 0)


def residual_bottleneck(feature_dim=256, num_blocks=1, l2norm=True, final_conv=False, norm_scale=1.0, out_dim=None,
                        interp_cat=False, final_relu=False, final_pool=False, input_dim=None, final_stride=1):
    if not covered(feature_dim, num_blocks, l2norm, final_conv, norm_scale, out_dim, interp_cat, final_relu, final_pool,
                   input_dim, final_stride):
        raise NotImplementedError("residual_bottleneck: only num_blocks=0 + final 3x3 conv + InstanceL2Norm is on the hot path")
    cin = 4 * feature_dim if input_dim is None else input_dim
    cout = feature_dim if out_dim is None else out_dim
    return ClfHead(nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=False, stride=1), InstanceL2Norm(scale=norm_scale))
