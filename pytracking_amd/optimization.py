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
from .filter import _ptr, _require_device, _stream, workspace


class MLU:
    """Marker for the MLU response activation (ltr/models/layers/activation.py:20-29)."""

    def __init__(self, min_val, inplace=False):
        self.min_val = min_val


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


class ConjugateGradient:
    def __init__(self, problem, variable, cg_eps=0.0, fletcher_reeves=True, standard_alpha=True,
                 direction_forget_factor=0, debug=False, plotting=False, visdom=None):
        if not hasattr(problem, "training_samples") or not hasattr(problem.response_activation, "min_val"):
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
        y = y.reshape(n, H, W).contiguous()
        sw = sw.reshape(n).contiguous()
        if self._state is None or self._state.numel() != 2 * C * K * K + 4:
            self._state = torch.zeros(2 * C * K * K + 4, dtype=torch.float32, device=x.device)
        L = _lib.lib()
        ws = workspace(L.pt_atom_cg_ws_bytes(n, C, H, W, K), x.device)
        rc = L.pt_atom_cg_f32(_ptr(x), _ptr(samples), samples.stride(0), _ptr(y), _ptr(sw), lam,
                              float(self.problem.response_activation.min_val), n, C, H, W, K, int(num_cg_iter),
                              int(bool(self.fletcher_reeves)), float(self.direction_forget_factor), _ptr(self._state),
                              _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "pt_atom_cg_f32")
