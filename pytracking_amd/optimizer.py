"""Host-side mirror of `ltr/models/target_classifier/optimizer.py`: the three unrolled steepest-descent
filter optimisers as nn.Modules whose `forward` runs the fused gfx950 solver (pt_sd_solve_f32).

Drop-in contract:
  * same class names, constructor arguments and `forward(weights, feat, bb, sample_weight=None,
    num_iter=None, compute_losses=True)` -> (weights, weight_iterates, losses);
  * same parameter / sub-module names so reference checkpoints load: `log_step_length`, `filter_reg`,
    `label_map_predictor.weight`, `target_mask_predictor.0.weight`, `spatial_weight_predictor.weight`;
  * hyper-parameters are read from the module attributes at call time -- the trackers mutate
    `filter_reg[0]`, `min_filter_reg`, `alpha_eps`, `label_threshold`, `label_shrink`, `softmax_reg`
    after construction (pytracking/tracker/dimp/dimp.py:589-602);
  * `weights` is not modified; every iterate is a fresh tensor.
Only the inference path (torch.no_grad, as in dimp.py:632-639) is implemented; back-propagating through the
unrolled iterations is offline training and raises.
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import _lib
from .filter import _ptr, _require_device, _stream, device_guarded, workspace


_SIDE = {}


def _side_streams(device, count):
    """Side streams of the multi-sequence solve, per (device, calling stream): created once, reused every call."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    pool = _SIDE.setdefault(key, [])
    while len(pool) < count:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:count]


# Experiment / test knob: spread the sequences of an S > 1 solve over side streams in eager mode as well (default: only inside a graph
# capture, see _solve).  tests/test_gpu_parity.py sets it to check the fork / join path of pt_sd_solve_batch_f32 result by result.
EAGER_SIDE_STREAMS = False


@device_guarded
def _solve(params: _lib.SdParams, weights, feat, bb, sample_weight, num_iter, compute_losses, keep):
    """Shared driver.  One sequence -> pt_sd_solve_f32; S > 1 (optimizer.py:101-104) -> ONE call of pt_sd_solve_batch_f32."""
    if torch.is_grad_enabled() and (weights.requires_grad or feat.requires_grad):
        raise NotImplementedError("back-propagation through the unrolled optimiser (offline training) is out of scope")
    _require_device(weights, feat, bb)
    L = _lib.lib()
    f5 = feat if feat.dim() == 5 else feat.unsqueeze(1)
    n, S, C, H, W = f5.shape
    K = weights.shape[-1]
    assert weights.shape[0] == S and weights.shape[-2] == K
    if f5.stride()[2:] != (H * W, W, 1):
        f5 = f5.contiguous()
    bb3 = bb.reshape(n, S, 4).to(torch.float32)
    sw2 = None
    if isinstance(sample_weight, torch.Tensor):
        sw2 = sample_weight.reshape(n, S).to(torch.float32)
    elif sample_weight is not None:
        raise NotImplementedError("scalar sample_weight")
    w_in = weights.detach().contiguous()
    iters = torch.empty((S, num_iter + 1, C, K, K), dtype=torch.float32, device=feat.device)
    losses = torch.zeros((S, num_iter + 1), dtype=torch.float32, device=feat.device) if compute_losses else None
    nb = L.pt_sd_ws_bytes(n, C, H, W, K)
    if S == 1:
        ws = workspace(nb, feat.device)
        fs = f5[:, 0]
        bs = bb3[:, 0].contiguous()
        ss = sw2[:, 0].contiguous() if sw2 is not None else None
        keep.extend((bs, ss))
        rc = L.pt_sd_solve_f32(ctypes.byref(params), _ptr(w_in[0]), _ptr(fs), fs.stride(0), _ptr(bs),
                               _ptr(ss) if ss is not None else None, n, C, H, W, K, num_iter, _ptr(iters[0]),
                               _ptr(losses[0]) if losses is not None else None, _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "pt_sd_solve_f32")
    else:
        # all sequences in one call (pt_sd_solve_batch_f32).  Spreading them over side streams pays only when the launches are not
        # the bottleneck: issued eagerly, two chains need two launches per ~6 us of device time, which is the host's launch rate, and
        # the second chain starts a whole chain's launch time late (profiles/r05i_multi_seq_solve.json: 0.80x / 0.97x / 1.07x for
        # S = 2 / 4 / 8).  Inside a graph capture the fork / join events become parallel branches of the graph and the replay runs
        # the chains concurrently (bench.py `multi_sequence`: 1.3x / 1.5x for 2 / 4 sequences), so side streams are used there only.
        nb_al = (nb + 255) // 256 * 256
        ws = workspace(nb_al * S, feat.device)
        bsT = bb3.permute(1, 0, 2).contiguous()                   # (S, n, 4): one dense box array per sequence
        ssT = sw2.t().contiguous() if sw2 is not None else None
        keep.extend((bsT, ssT))
        arr = ctypes.c_void_p * S
        p_w = arr(*[w_in[s].data_ptr() for s in range(S)])
        p_f = arr(*[f5[:, s].data_ptr() for s in range(S)])
        p_b = arr(*[bsT[s].data_ptr() for s in range(S)])
        p_s = arr(*[ssT[s].data_ptr() for s in range(S)]) if ssT is not None else None
        p_i = arr(*[iters[s].data_ptr() for s in range(S)])
        p_l = arr(*[losses[s].data_ptr() for s in range(S)]) if losses is not None else None
        p_ws = arr(*[ws.data_ptr() + s * nb_al for s in range(S)])
        aux = _side_streams(feat.device, min(S - 1, 3)) if (EAGER_SIDE_STREAMS or torch.cuda.is_current_stream_capturing()) else []
        p_aux = (ctypes.c_void_p * max(len(aux), 1))(*[a.cuda_stream for a in aux])
        rc = L.pt_sd_solve_batch_f32(ctypes.byref(params), S, p_w, p_f, f5.stride(0), p_b, p_s, n, C, H, W, K, num_iter, p_i, p_l,
                                     p_ws, nb_al, _stream(), p_aux, len(aux))
        _lib.check(rc, "pt_sd_solve_batch_f32")
    weight_iterates = [weights] + [iters[:, t] for t in range(1, num_iter + 1)]
    loss_list = []
    if compute_losses:
        tot = losses.sum(dim=0) / S                       # optimizer.py:143: (...)/num_sequences
        # shape (1,) per iterate like the reference's `(... + reg_weight * ...)/num_sequences` with reg_weight (1,)
        # (optimizer.py:143,168): the trackers `torch.cat(losses)` in debug mode (dimp.py:583,641)
        loss_list = [tot[t:t + 1] for t in range(num_iter + 1)]
    return weight_iterates[-1], weight_iterates, loss_list


def _host_scalar(mod, name):
    """Host value of a one-element parameter, re-read from the device only when the tensor changed (the trackers
    overwrite `filter_reg[0]` in place, dimp.py:598-600 -- that bumps `_version`): no device->host synchronisation on
    the per-frame path.  The reference re-reads the parameter on every call; what the version counter cannot see is
    covered explicitly: `load_state_dict` and `.to()/.cuda()/.float()` drop the cache (`_ScalarCacheMixin`),
    tensors without a version counter (built under `torch.inference_mode`) are read directly every time, and a write
    through `.data` (which does not bump the version) needs `refresh_host_scalars(module)`."""
    t = getattr(mod, name)
    try:
        key = (t.data_ptr(), t._version)
    except RuntimeError:                                   # inference tensors do not track versions
        return float(t.detach().reshape(-1)[0])
    cache = mod.__dict__.setdefault("_host_scalars", {})
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        hit = (key, float(t.detach().reshape(-1)[0]))
        cache[name] = hit
    return hit[1]


def refresh_host_scalars(mod):
    """Drop the cached host copies of the one-element parameters of `mod` and its sub-modules (call after writing a
    parameter through `.data`, e.g. `opt.filter_reg.data.fill_(x)`)."""
    for m in mod.modules():
        m.__dict__.pop("_host_scalars", None)
        m.__dict__.pop("_lut3_cache", None)


class _ScalarCacheMixin:
    """Invalidate the host-scalar cache wherever parameters are replaced wholesale."""

    def _load_from_state_dict(self, *args, **kw):
        self.__dict__.pop("_host_scalars", None)
        self.__dict__.pop("_lut3_cache", None)
        return super()._load_from_state_dict(*args, **kw)

    def _apply(self, fn, *args, **kw):
        self.__dict__.pop("_host_scalars", None)
        self.__dict__.pop("_lut3_cache", None)
        return super()._apply(fn, *args, **kw)


def _contiguous_luts(mod, weights):
    """The look-up tables of the DiMP optimiser (three 1x1-conv weights, optimizer.py:45-72) as views of ONE contiguous float32 array
    (label | mask | spatial): the solver's init stage then requests them with its first loads (k_fast_init2, `lut3`) instead of behind its
    late argument block.  Concatenated once and reused while the weights are unchanged -- same invalidation rules as `_host_scalar`
    (version counters; `refresh_host_scalars` after a write through `.data`); tables of different lengths stay separate tensors."""
    ws = [w.detach() for w in weights]
    if len({w.numel() for w in ws}) != 1:
        return [w.to(torch.float32).contiguous().reshape(-1) for w in ws]
    try:
        key = tuple((w.data_ptr(), w._version, str(w.device), w.dtype) for w in weights)
    except RuntimeError:                                   # inference tensors do not track versions
        key = None
    hit = mod.__dict__.get("_lut3_cache") if key is not None else None
    if hit is None or hit[0] != key:
        buf = torch.cat([w.to(torch.float32).reshape(-1) for w in ws]).contiguous()
        hit = (key, buf)
        if key is not None:
            mod.__dict__["_lut3_cache"] = hit
    n = ws[0].numel()
    return [hit[1][k * n:(k + 1) * n] for k in range(3)]


def _reg_value(mod):
    fr = _host_scalar(mod, "filter_reg")
    return max(fr * fr, float(mod.min_filter_reg) ** 2)           # optimizer.py:109


class DiMPSteepestDescentGN(_ScalarCacheMixin, nn.Module):
    """reference: optimizer.py:11-170."""

    def __init__(self, num_iter=1, feat_stride=16, init_step_length=1.0, init_filter_reg=1e-2, init_gauss_sigma=1.0,
                 num_dist_bins=5, bin_displacement=1.0, mask_init_factor=4.0, score_act='relu', act_param=None,
                 min_filter_reg=1e-3, mask_act='sigmoid', detach_length=float('Inf'), alpha_eps=0):
        super().__init__()
        if score_act not in ('relu', 'bentpar'):
            raise ValueError('Unknown score activation')
        if mask_act not in ('sigmoid', 'linear'):
            raise ValueError('Unknown activation')
        self.num_iter, self.feat_stride = num_iter, feat_stride
        self.min_filter_reg, self.detach_length, self.alpha_eps = min_filter_reg, detach_length, alpha_eps
        self.num_dist_bins, self.bin_displacement = num_dist_bins, bin_displacement
        self.score_act, self.act_param, self.mask_act = score_act, act_param, mask_act
        self.log_step_length = nn.Parameter(torch.full((1,), math.log(init_step_length)))
        self.filter_reg = nn.Parameter(torch.full((1,), float(init_filter_reg)))
        # radial look-up tables (1x1 convs over the distance bins in the reference, optimizer.py:45-72)
        dist = torch.arange(num_dist_bins, dtype=torch.float32).view(1, -1, 1, 1) * bin_displacement
        if init_gauss_sigma == 0:
            gauss = torch.zeros_like(dist)
            gauss[0, 0, 0, 0] = 1
        else:
            gauss = torch.exp(-0.5 * (dist / init_gauss_sigma) ** 2)
        self.label_map_predictor = nn.Conv2d(num_dist_bins, 1, kernel_size=1, bias=False)
        self.label_map_predictor.weight.data = gauss - gauss.min()
        self.target_mask_predictor = nn.Sequential(nn.Conv2d(num_dist_bins, 1, kernel_size=1, bias=False),
                                                   *([nn.Sigmoid()] if mask_act == 'sigmoid' else []))
        self.target_mask_predictor[0].weight.data = mask_init_factor * torch.tanh(2.0 - dist) + \
            (0.0 if mask_act == 'sigmoid' else 0.5)
        self.spatial_weight_predictor = nn.Conv2d(num_dist_bins, 1, kernel_size=1, bias=False)
        self.spatial_weight_predictor.weight.data.fill_(1.0)

    def forward(self, weights, feat, bb, sample_weight=None, num_iter=None, compute_losses=True):
        num_iter = self.num_iter if num_iter is None else num_iter
        luts = _contiguous_luts(self, (self.label_map_predictor.weight, self.target_mask_predictor[0].weight,
                                       self.spatial_weight_predictor.weight))
        p = _lib.SdParams()
        p.kind = _lib.PT_SD_DIMP
        p.step_length = math.exp(_host_scalar(self, "log_step_length"))
        p.reg = _reg_value(self)
        p.alpha_eps = float(self.alpha_eps)
        p.feat_stride = float(self.feat_stride)
        p.num_bins = int(luts[0].numel())
        p.bin_displacement = float(self.bin_displacement)
        p.label_lut, p.mask_lut, p.spatial_lut = (t.data_ptr() for t in luts)
        p.mask_act = _lib.PT_MASK_SIGMOID if self.mask_act == 'sigmoid' else _lib.PT_MASK_LINEAR
        p.score_act = _lib.PT_ACT_RELU if self.score_act == 'relu' else _lib.PT_ACT_BENTPAR
        p.act_param = float(self.act_param) if self.act_param is not None else 1.0
        return _solve(p, weights, feat, bb, sample_weight, num_iter, compute_losses, keep=luts)


class DiMPL2SteepestDescentGN(_ScalarCacheMixin, nn.Module):
    """reference: optimizer.py:174-291."""

    def __init__(self, num_iter=1, feat_stride=16, init_step_length=1.0, gauss_sigma=1.0, hinge_threshold=-999,
                 init_filter_reg=1e-2, min_filter_reg=1e-3, detach_length=float('Inf'), alpha_eps=0.0):
        super().__init__()
        self.num_iter, self.feat_stride = num_iter, feat_stride
        self.log_step_length = nn.Parameter(torch.full((1,), math.log(init_step_length)))
        self.filter_reg = nn.Parameter(torch.full((1,), float(init_filter_reg)))
        self.min_filter_reg, self.detach_length = min_filter_reg, detach_length
        self.hinge_threshold, self.gauss_sigma, self.alpha_eps = hinge_threshold, gauss_sigma, alpha_eps

    def forward(self, weights, feat, bb, sample_weight=None, num_iter=None, compute_losses=True):
        num_iter = self.num_iter if num_iter is None else num_iter
        p = _lib.SdParams()
        p.kind = _lib.PT_SD_DIMP_L2
        p.step_length = math.exp(_host_scalar(self, "log_step_length"))
        p.reg = _reg_value(self)
        p.alpha_eps = float(self.alpha_eps)
        p.feat_stride = float(self.feat_stride)
        p.gauss_sigma = float(self.gauss_sigma)
        p.hinge_threshold = float(self.hinge_threshold)
        return _solve(p, weights, feat, bb, sample_weight, num_iter, compute_losses, keep=[])


class PrDiMPSteepestDescentNewton(_ScalarCacheMixin, nn.Module):
    """reference: optimizer.py:294-439."""

    def __init__(self, num_iter=1, feat_stride=16, init_step_length=1.0, init_filter_reg=1e-2, gauss_sigma=1.0,
                 min_filter_reg=1e-3, detach_length=float('Inf'), alpha_eps=0.0, init_uni_weight=None,
                 normalize_label=False, label_shrink=0, softmax_reg=None, label_threshold=0.0):
        super().__init__()
        self.num_iter, self.feat_stride = num_iter, feat_stride
        self.log_step_length = nn.Parameter(torch.full((1,), math.log(init_step_length)))
        self.filter_reg = nn.Parameter(torch.full((1,), float(init_filter_reg)))
        self.gauss_sigma, self.min_filter_reg, self.detach_length = gauss_sigma, min_filter_reg, detach_length
        self.alpha_eps = alpha_eps
        self.uni_weight = 0 if init_uni_weight is None else init_uni_weight
        self.normalize_label, self.label_shrink = normalize_label, label_shrink
        self.softmax_reg, self.label_threshold = softmax_reg, label_threshold

    def forward(self, weights, feat, bb, sample_weight=None, num_iter=None, compute_losses=True):
        num_iter = self.num_iter if num_iter is None else num_iter
        p = _lib.SdParams()
        p.kind = _lib.PT_SD_PRDIMP
        p.step_length = math.exp(_host_scalar(self, "log_step_length"))
        p.reg = _reg_value(self)
        p.alpha_eps = float(self.alpha_eps)
        p.feat_stride = float(self.feat_stride)
        p.gauss_sigma = float(self.gauss_sigma)
        p.uni_weight = float(self.uni_weight)
        p.normalize_label = int(bool(self.normalize_label))
        p.label_shrink = float(self.label_shrink)
        p.has_softmax_reg = int(self.softmax_reg is not None)
        p.softmax_reg = float(self.softmax_reg) if self.softmax_reg is not None else 0.0
        p.label_threshold = float(self.label_threshold)
        return _solve(p, weights, feat, bb, sample_weight, num_iter, compute_losses, keep=[])
