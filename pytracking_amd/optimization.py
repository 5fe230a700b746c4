"""Host-side mirror of the ATOM online optimiser: `pytracking/libs/optimization.py:227-289`
(`ConjugateGradient`) for `pytracking/tracker/atom/optim.py:71-99` (`ConvProblem`).

`ConjugateGradient(problem, variable, ...)` keeps the reference's constructor and `.run(num_cg_iter)`;
`variable` (the tracker's filter tensors, aliased at pytracking/tracker/atom/atom.py:141,200,302) is updated
IN PLACE like `self.x += delta_x` (optimization.py:259-260).  The fast path is keyed on the problem being a
ConvProblem with an MLU response activation; anything else raises (no autograd fallback on the product path).
"""
import ctypes
import math

import torch

from . import _lib
from .filter import _ptr, _require_device, _stream, on_device, workspace


class MLU:
    """Marker for the MLU response activation (ltr/models/layers/activation.py:20-29)."""

    def __init__(self, min_val, inplace=False):
        self.min_val = min_val


def activation_kind(fn):
    """Recognise the two activations the fused solvers implement, whatever object carries them.

    The reference tracker passes plain lambdas (atom.py:444-466): `lambda x: x` for 'none' and
    `lambda x: F.elu(F.leaky_relu(x, 1/a), a)` for ('mlu', a); its networks pass `activation.MLU` modules.  Returns
    ("identity", None), ("mlu", a) or (None, None).  Lambdas are recognised by evaluating them on nine probe values
    on the host -- nothing is sent to the device and nothing is assumed about an unrecognised callable."""
    if fn is None or getattr(fn, "is_identity", False):
        return "identity", None
    if hasattr(fn, "min_val"):
        return "mlu", float(fn.min_val)
    probe = torch.tensor([-1e4, -3.0, -0.7, -0.05, 0.0, 0.04, 0.9, 2.5, 40.0], dtype=torch.float64)
    try:
        out = fn(probe.clone())
    except Exception:
        return None, None
    if not isinstance(out, torch.Tensor) or out.shape != probe.shape:
        return None, None
    if torch.equal(out, probe):
        return "identity", None
    a = -float(out[0])
    if a > 0:
        mlu = torch.where(probe >= 0, probe, a * torch.expm1(probe / a))
        if torch.allclose(out, mlu, rtol=1e-9, atol=1e-12):
            return "mlu", a
    return None, None


class ConvProblem:
    """Data holder with the reference's constructor signature (atom/optim.py:71-77).  TensorList arguments of
    the reference are accepted as plain lists with one entry per feature block."""

    def __init__(self, training_samples, y, filter_reg, sample_weights, response_activation):
        self.training_samples = training_samples
        self.y = y
        self.filter_reg = filter_reg
        self.sample_weights = sample_weights
        self.response_activation = response_activation


def _first(x):
    return x[0] if isinstance(x, (list, tuple)) or hasattr(x, "__getitem__") and not isinstance(x, torch.Tensor) else x


class FactorizedConvProblem:
    """Data holder with the reference's constructor signature (atom/optim.py:6-17)."""

    def __init__(self, training_samples, y, filter_reg, projection_reg, params, sample_weights, projection_activation,
                 response_activation):
        self.training_samples = training_samples
        self.y = y
        self.filter_reg = filter_reg
        self.projection_reg = projection_reg
        self.params = params
        self.sample_weights = sample_weights
        self.projection_activation = projection_activation
        self.response_activation = response_activation


class GaussNewtonCG:
    """`GaussNewtonCG` (optimization.py:293-421) for `FactorizedConvProblem`: `variable` = [filter (1,Kc,K,K),
    projection matrix (Kc,M,1,1)] is updated IN PLACE; `.run(num_cg_iter, num_gn_iter)` as in the reference."""

    def __init__(self, problem, variable, cg_eps=0.0, fletcher_reeves=True, standard_alpha=True,
                 direction_forget_factor=0, debug=False, analyze=False, plotting=False, visdom=None):
        kind, self.act_min_val = activation_kind(problem.response_activation)
        if not hasattr(problem, "projection_reg") or kind != "mlu":
            raise NotImplementedError("fast GaussNewtonCG covers FactorizedConvProblem with an MLU response activation")
        if not standard_alpha or cg_eps != 0.0 or debug or analyze or plotting or direction_forget_factor != 0:
            raise NotImplementedError("non-standard alpha / cg_eps / forgetting / debug modes are not on the hot path")
        if activation_kind(problem.projection_activation)[0] != "identity":
            raise NotImplementedError("only the identity projection activation (ATOM default 'none') is covered")
        self.problem = problem
        self.x = variable
        self.fletcher_reeves = fletcher_reeves
        self.residuals = torch.zeros(0)
        self.losses = torch.zeros(0)

    def run_GN(self, *args, **kwargs):
        return self.run(*args, **kwargs)

    def run(self, num_cg_iter, num_gn_iter=None):
        if isinstance(num_cg_iter, int):
            if num_gn_iter is None:
                raise ValueError('Must specify number of GN iter if CG iter is constant')
            num_cg_iter = [num_cg_iter] * num_gn_iter
        if len(num_cg_iter) == 0:
            return
        samples = _first(self.problem.training_samples)
        y = _first(self.problem.y)
        sw = _first(self.problem.sample_weights)
        lf = float(_first(self.problem.filter_reg))
        lP = float(_first(self.problem.projection_reg))
        filt, proj = self.x[0], self.x[1]                   # (1,Kc,K,K), (Kc,M,1,1): both updated in place
        _require_device(samples, y, sw, filt, proj)
        n, M, H, W = samples.shape
        Kc, K = filt.shape[1], filt.shape[-1]
        assert filt.is_contiguous() and proj.is_contiguous() and samples.stride()[1:] == (H * W, W, 1)
        assert proj.shape[0] == Kc and proj.shape[1] == M
        if filt.shape[-2] != K or K * K > 16:
            raise RuntimeError("GaussNewtonCG: square filters with at most 16 taps are covered by the gfx950 kernels")
        L = _lib.lib()
        with on_device(filt):
            y = y.reshape(n, H, W).contiguous()
            # the tracker's first frame passes ONE weight, 1 / n, broadcast over the samples (atom.py:578 init_sample_weights)
            sw = (sw.reshape(-1).expand(n) if sw.numel() == 1 else sw.reshape(n)).contiguous()
            nb = L.pt_atom_gn_ws_bytes(n, M, Kc, H, W, K)
            if nb == 0:
                raise RuntimeError("GaussNewtonCG: configuration not covered by the gfx950 kernels")
            ws = workspace(nb, filt.device)
            iters = (ctypes.c_int * len(num_cg_iter))(*[int(v) for v in num_cg_iter])
            rc = L.pt_atom_gn_f32(_ptr(filt), _ptr(proj), _ptr(samples), samples.stride(0), _ptr(y), _ptr(sw), lf, lP,
                                  self.act_min_val, n, M, Kc, H, W, K, iters,
                                  len(num_cg_iter), int(bool(self.fletcher_reeves)), _ptr(ws), ws.numel(), _stream())
            _lib.check(rc, "pt_atom_gn_f32")
        return self.losses, self.residuals


class ConjugateGradient:
    def __init__(self, problem, variable, cg_eps=0.0, fletcher_reeves=True, standard_alpha=True,
                 direction_forget_factor=0, debug=False, plotting=False, visdom=None):
        kind, self.act_min_val = activation_kind(problem.response_activation)
        if not hasattr(problem, "training_samples") or kind != "mlu":
            raise NotImplementedError("fast ConjugateGradient covers ConvProblem with an MLU response activation")
        if not standard_alpha or cg_eps != 0.0 or debug or plotting:
            raise NotImplementedError("non-standard alpha / cg_eps / debug modes are not on the hot path")
        self.problem = problem
        self.x = variable
        self.fletcher_reeves = fletcher_reeves
        self.direction_forget_factor = direction_forget_factor
        self.residuals = torch.zeros(0)
        self.losses = torch.zeros(0)
        self._state = None

    def reset_state(self):
        self._state = None

    def run(self, num_cg_iter):
        if num_cg_iter == 0:
            return
        samples = _first(self.problem.training_samples)
        y = _first(self.problem.y)
        sw = _first(self.problem.sample_weights)
        lam = float(_first(self.problem.filter_reg))
        x = _first(self.x)                                  # (1, C, K, K), updated in place
        _require_device(samples, y, sw, x)
        n, C, H, W = samples.shape
        K = x.shape[-1]
        assert x.is_contiguous() and samples.stride()[1:] == (H * W, W, 1)
        if x.shape[-2] != K or K * K > 16:
            raise RuntimeError("ConjugateGradient: square filters with at most 16 taps are covered by the gfx950 kernels")
        L = _lib.lib()
        with on_device(x):
            y = y.reshape(n, H, W).contiguous()
            sw = sw.reshape(n).contiguous()
            if self._state is None or self._state.numel() != 2 * C * K * K + 4 or self._state.device != x.device:
                self._state = torch.zeros(2 * C * K * K + 4, dtype=torch.float32, device=x.device)
            ws = workspace(L.pt_atom_cg_ws_bytes(n, C, H, W, K), x.device)
            rc = L.pt_atom_cg_f32(_ptr(x), _ptr(samples), samples.stride(0), _ptr(y), _ptr(sw), lam,
                                  self.act_min_val, n, C, H, W, K, int(num_cg_iter),
                                  int(bool(self.fletcher_reeves)), float(self.direction_forget_factor), _ptr(self._state),
                                  _ptr(ws), ws.numel(), _stream())
            _lib.check(rc, "pt_atom_cg_f32")
