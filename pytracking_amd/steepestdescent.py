"""Host-side mirror of the LWL few-shot learner: `ltr/models/meta/steepestdescent.py` (`GNSteepestDescent`) on
`ltr/models/lwl/loss_residual_modules.py` (`LWTLResidual`).

Same class names, constructor arguments, parameter names (`LWTLResidual.filter_reg`) and
`forward(meta_parameter, num_iter=None, feat=, label=, sample_weight=)` -> (meta_parameter, iterates, losses) contract:
`meta_parameter` may be a tensor (sequences, filters, C, K, K) or a one-element list (the reference's TensorList);
iterates and the result keep that form.  The generic autograd formulation of the reference is replaced by the explicit
Gauss-Newton recurrences of the fused gfx950 solver (pt_lwl_gn_solve_f32), so only an `LWTLResidual` residual module is
accepted; anything else raises (no autograd fallback on the product path).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .filter import _ptr, _require_device, _stream, apply_filter, device_guarded, workspace
from .optimizer import _ScalarCacheMixin, _host_scalar


class LWTLResidual(_ScalarCacheMixin, nn.Module):
    """reference: loss_residual_modules.py:8-41 (residuals W*(T(x) - E(y)) and lambda*tau of the few-shot loss)."""

    def __init__(self, init_filter_reg=1e-2, filter_dilation_factors=None):
        super().__init__()
        self.filter_reg = nn.Parameter(init_filter_reg * torch.ones(1))
        self.filter_dilation_factors = filter_dilation_factors

    def forward(self, meta_parameter, feat, label, sample_weight=None):
        """The residual vectors themselves (reference semantics), computed with the gfx950 filter layer."""
        filt = meta_parameter[0] if isinstance(meta_parameter, (list, tuple)) else meta_parameter
        num_images = feat.shape[0]
        num_sequences = feat.shape[1] if feat.dim() == 5 else 1
        scores = apply_filter(feat, filt, dilation_factors=self.filter_dilation_factors)
        if sample_weight is None:
            sample_weight = (1.0 / num_images) ** 0.5
        elif isinstance(sample_weight, torch.Tensor):
            if sample_weight.numel() == scores.numel():
                sample_weight = sample_weight.view(scores.shape)
            elif sample_weight.dim() == 1:
                sample_weight = sample_weight.view(-1, 1, 1, 1, 1)
        data_residual = sample_weight * (scores - label.view(scores.shape))
        reg_residual = self.filter_reg * filt.reshape(1, num_sequences, -1)
        return [data_residual, reg_residual]


class GNSteepestDescent(nn.Module):
    """reference: steepestdescent.py:8-105."""

    def __init__(self, residual_module, num_iter=1, compute_losses=False, detach_length=float('Inf'),
                 parameter_batch_dim=0, residual_batch_dim=0, steplength_reg=0.0):
        super().__init__()
        self.residual_module = residual_module
        self.num_iter = num_iter
        self.compute_losses = compute_losses
        self.detach_length = detach_length
        self.steplength_reg = steplength_reg
        self._parameter_batch_dim = parameter_batch_dim
        self._residual_batch_dim = residual_batch_dim

    @device_guarded
    def forward(self, meta_parameter, num_iter=None, *args, **kwargs):
        res = self.residual_module
        if not isinstance(res, LWTLResidual) or res.filter_dilation_factors is not None:
            raise NotImplementedError("fused GNSteepestDescent covers LWTLResidual without dilation factors")
        if self._parameter_batch_dim != 0 or self._residual_batch_dim != 1:
            # lwl_net.py:192-194 builds it with residual_batch_dim=1: one step length per sequence
            raise NotImplementedError("fused GNSteepestDescent: parameter_batch_dim=0, residual_batch_dim=1 only")
        input_is_list = isinstance(meta_parameter, (list, tuple))
        weights = meta_parameter[0] if input_is_list else meta_parameter
        if torch.is_grad_enabled() and weights.requires_grad:
            raise NotImplementedError("back-propagation through the unrolled optimiser (offline training) is out of scope")
        feat, label = kwargs["feat"], kwargs["label"]
        sample_weight = kwargs.get("sample_weight")
        num_iter = self.num_iter if num_iter is None else num_iter
        _require_device(weights, feat, label)
        f5 = feat if feat.dim() == 5 else feat.unsqueeze(1)
        n, S, C, H, W = f5.shape
        assert weights.dim() == 5 and weights.shape[0] == S
        Fn, K = weights.shape[1], weights.shape[-1]
        if f5.stride()[2:] != (H * W, W, 1):
            f5 = f5.contiguous()
        lab = label.reshape(n, S, Fn, H, W).to(torch.float32)
        sw_mode, sw = 0, None
        if isinstance(sample_weight, torch.Tensor):
            if sample_weight.numel() == lab.numel():
                sw_mode, sw = 2, sample_weight.reshape(n, S, Fn, H, W).to(torch.float32)
            elif sample_weight.dim() == 1:
                sw_mode, sw = 1, sample_weight.reshape(n).to(torch.float32).contiguous()
            else:
                raise NotImplementedError("sample_weight must be per element or per image")
        elif sample_weight is not None:
            raise NotImplementedError("scalar sample_weight")
        L = _lib.lib()
        nb = L.pt_lwl_ws_bytes(n, Fn, C, H, W, K)
        if nb == 0:
            raise RuntimeError("GNSteepestDescent: configuration not covered by the gfx950 kernels "
                               "(needs <= 16 filters, K in {1,3}, W <= 256)")
        ws = workspace(nb, feat.device)
        w_in = weights.detach().contiguous()
        iters = torch.empty((S, num_iter + 1, Fn, C, K, K), dtype=torch.float32, device=feat.device)
        losses = torch.zeros((S, num_iter + 1), dtype=torch.float32, device=feat.device) if self.compute_losses else None
        lam = _host_scalar(res, "filter_reg")
        keep = []
        for s in range(S):
            fs = f5[:, s]
            ls = lab[:, s].contiguous()
            ss = sw if sw_mode == 1 else (sw[:, s].contiguous() if sw_mode == 2 else None)
            keep.extend((ls, ss))
            rc = L.pt_lwl_gn_solve_f32(_ptr(w_in[s]), _ptr(fs), fs.stride(0), _ptr(ls), _ptr(ss) if ss is not None else None,
                                       sw_mode, lam, float(self.steplength_reg), n, Fn, C, H, W, K, int(num_iter),
                                       _ptr(iters[s]), _ptr(losses[s]) if losses is not None else None, _ptr(ws),
                                       ws.numel(), _stream())
            _lib.check(rc, "pt_lwl_gn_solve_f32")
        its = [weights] + [iters[:, t] for t in range(1, num_iter + 1)]
        loss_list = []
        if self.compute_losses:
            # the reference's loss is sum(r^2)/numel(r) over ALL sequences (steepestdescent.py:28-29); the C ABI
            # returns the per-sequence value, and with equal-sized sequences the overall value is their mean
            tot = losses.mean(dim=0)
            loss_list = [tot[t] for t in range(num_iter + 1)]
        if input_is_list:
            return [its[-1]], [[w] for w in its], loss_list
        return its[-1], its, loss_list
