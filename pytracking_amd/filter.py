"""Host-side mirror of `ltr/models/layers/filter.py` on top of libpt_hot.so.

Same names, argument meaning and shape conventions as the reference:
    apply_filter(feat, filter, dilation_factors=None)            filter.py:5-57
    apply_feat_transpose(feat, input, filter_ksz, training, groups)   filter.py:91-107
    filter_gradient(feat, filter, label=None, training=True)     filter.py:203-216
feat is (images, [sequences], C, H, W); filter (sequences, C, KH, KW); scores (images, sequences, OH, OW).

Device tensors only -- there is no CPU fallback on the product path (tests compare against oracle/).
`apply_filter` and `apply_feat_transpose` are each other's backward w.r.t. filter / input, so the pair
stays usable under autograd for filter-space optimisers; gradients w.r.t. the features are a training-only
path that is out of scope and raise NotImplementedError.
"""
import ctypes
import functools

import torch

from . import _lib

_WS = {}


def _first_device_tensor(objs):
    for o in objs:
        if isinstance(o, torch.Tensor):
            if o.is_cuda:
                return o
        elif isinstance(o, (list, tuple)):
            t = _first_device_tensor(o)
            if t is not None:
                return t
    return None


def device_guarded(fn):
    """The C ABI launches on `torch.cuda.current_stream()` of the CURRENT device.  Stock PyTorch ops guard on the device
    of their operands; this decorator does the same for the ctypes calls: when the first device tensor among the
    arguments lives on another GPU than the current one (single-process multi-GPU, `params.device='cuda:1'`), the body
    runs under `torch.cuda.device(that GPU)` so that stream, workspace and pointers agree."""
    @functools.wraps(fn)
    def wrapper(*args, **kw):
        t = _first_device_tensor(args)
        if t is None:
            t = _first_device_tensor(list(kw.values()))
        if t is None or t.device.index == torch.cuda.current_device():
            return fn(*args, **kw)
        with torch.cuda.device(t.device):
            return fn(*args, **kw)
    return wrapper


class on_device:
    """`with on_device(t):` -- the same guard for call sites whose tensors are not arguments."""

    def __init__(self, t):
        self._ctx = None if t.device.index == torch.cuda.current_device() else torch.cuda.device(t.device)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *exc):
        if self._ctx is not None:
            return self._ctx.__exit__(*exc)


def workspace(nbytes: int, device) -> torch.Tensor:
    """Caller-owned scratch for the C ABI (the library never allocates).  One growing buffer per (device, stream):
    calls on one stream are ordered, so they may share scratch; two streams of one device (a tracker per stream) get
    separate buffers and cannot alias.  A buffer that is outgrown is not handed back to the allocator while kernels of
    earlier calls may still use it: `record_stream` defers its reuse until the work queued on this stream has run."""
    stream = torch.cuda.current_stream(device)
    key = (device.type, device.index, stream.cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            buf.record_stream(stream)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _require_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("pytracking_amd ops run on the MI355X only (got a CPU tensor); "
                               "the CPU restatement lives in oracle/ and is test infrastructure")
        if t.dtype != torch.float32:
            raise RuntimeError(f"pytracking_amd ops are fp32 (got {t.dtype})")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"pytracking_amd ops need all operands on one GPU (got {dev} and {t.device})")


def _out_size(H, W, KH, KW):
    return H + 2 * (KH // 2) - KH + 1, W + 2 * (KW // 2) - KW + 1


@device_guarded
def corr_raw(feat4, filt3, out_hw=None):
    """feat4 (n,C,H,W) [sample-strided view allowed], filt3 (C,KH,KW) -> (n,OH,OW)."""
    n, C, H, W = feat4.shape
    KH, KW = filt3.shape[-2:]
    OH, OW = out_hw if out_hw is not None else _out_size(H, W, KH, KW)
    if feat4.stride()[1:] != (H * W, W, 1):
        feat4 = feat4.contiguous()
    filt3 = filt3.contiguous()
    L = _lib.lib()
    out = torch.empty((n, OH, OW), dtype=torch.float32, device=feat4.device)
    nb = L.pt_apply_filter_ws_bytes(n, C, H, W, KH, KW, OH, OW)
    ws = workspace(nb, feat4.device)
    rc = L.pt_apply_filter_f32(_ptr(feat4), feat4.stride(0), _ptr(filt3), _ptr(out), n, C, H, W, KH, KW, OH, OW,
                               _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "pt_apply_filter_f32")
    return out


@device_guarded
def adj_raw(feat4, inp3, ksz):
    """feat4 (n,C,H,W), inp3 (n,OH,OW) -> (C,KH,KW)."""
    n, C, H, W = feat4.shape
    KH, KW = ksz
    OH, OW = inp3.shape[-2:]
    if feat4.stride()[1:] != (H * W, W, 1):
        feat4 = feat4.contiguous()
    inp3 = inp3.contiguous()
    L = _lib.lib()
    out = torch.empty((C, KH, KW), dtype=torch.float32, device=feat4.device)
    nb = L.pt_feat_transpose_ws_bytes(n, C, H, W, KH, KW, OH, OW)
    ws = workspace(nb, feat4.device)
    rc = L.pt_feat_transpose_f32(_ptr(feat4), feat4.stride(0), _ptr(inp3), _ptr(out), n, C, H, W, KH, KW, OH, OW,
                                 _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "pt_feat_transpose_f32")
    return out


@device_guarded
def corr_mf_raw(feat4, filt4):
    """feat4 (n,C,H,W), filt4 (F,C,K,K) -> (n,F,H,W)   (multi-filter branch of apply_filter, filter.py:29-34)."""
    n, C, H, W = feat4.shape
    Fn, C2, KH, KW = filt4.shape
    assert C == C2 and KH == KW
    if feat4.stride()[1:] != (H * W, W, 1):
        feat4 = feat4.contiguous()
    filt4 = filt4.contiguous()
    out = torch.empty((n, Fn, H, W), dtype=torch.float32, device=feat4.device)
    L = _lib.lib()
    nb = L.pt_apply_filter_mf_ws_bytes(n, Fn, C, H, W, KH)
    if nb == 0:
        raise RuntimeError("multi-filter apply_filter: configuration not covered by the gfx950 kernels")
    ws = workspace(nb, feat4.device)
    rc = L.pt_apply_filter_mf_f32(_ptr(feat4), feat4.stride(0), _ptr(filt4), _ptr(out), n, Fn, C, H, W, KH, _ptr(ws),
                                  ws.numel(), _stream())
    _lib.check(rc, "pt_apply_filter_mf_f32")
    return out


@device_guarded
def adj_mf_raw(feat4, inp4, ksz):
    """feat4 (n,C,H,W), inp4 (n,F,H,W) -> (F,C,K,K)   (5-D input branch of apply_feat_transpose, filter.py:158-176)."""
    n, C, H, W = feat4.shape
    Fn = inp4.shape[1]
    KH, KW = ksz
    assert KH == KW and inp4.shape[-2:] == (H, W)
    if feat4.stride()[1:] != (H * W, W, 1):
        feat4 = feat4.contiguous()
    inp4 = inp4.contiguous()
    L = _lib.lib()
    out = torch.empty((Fn, C, KH, KW), dtype=torch.float32, device=feat4.device)
    nb = L.pt_feat_transpose_mf_ws_bytes(n, Fn, C, H, W, KH)
    if nb == 0:
        raise RuntimeError("multi-filter apply_feat_transpose: configuration not covered by the gfx950 kernels")
    ws = workspace(nb, feat4.device)
    rc = L.pt_feat_transpose_mf_f32(_ptr(feat4), feat4.stride(0), _ptr(inp4), _ptr(out), n, Fn, C, H, W, KH, _ptr(ws),
                                    ws.numel(), _stream())
    _lib.check(rc, "pt_feat_transpose_mf_f32")
    return out


class _ApplyFilterMF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, filt):
        f5 = _as5d(feat)
        S = f5.shape[1]
        scores = torch.stack([corr_mf_raw(f5[:, s], filt[s]) for s in range(S)], dim=1)
        ctx.save_for_backward(feat, filt)
        return scores

    @staticmethod
    def backward(ctx, grad):
        feat, filt = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("gradient of apply_filter w.r.t. the features (training path) is out of scope")
        return None, _ApplyFeatTransposeMF.apply(feat, grad, tuple(filt.shape[-2:]))


class _ApplyFeatTransposeMF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, inp, ksz):
        f5 = _as5d(feat)
        S = f5.shape[1]
        out = torch.stack([adj_mf_raw(f5[:, s], inp[:, s], ksz) for s in range(S)], dim=0)
        ctx.save_for_backward(feat)
        return out

    @staticmethod
    def backward(ctx, grad):
        (feat,) = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("gradient of apply_feat_transpose w.r.t. the features is out of scope")
        return None, _ApplyFilterMF.apply(feat, grad), None


def _as5d(feat):
    return feat if feat.dim() == 5 else feat.unsqueeze(1)


class _ApplyFilter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, filt):
        f5 = _as5d(feat)
        S = f5.shape[1]
        scores = torch.stack([corr_raw(f5[:, s], filt[s]) for s in range(S)], dim=1)
        ctx.save_for_backward(feat, filt)
        return scores

    @staticmethod
    def backward(ctx, grad):
        feat, filt = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("gradient of apply_filter w.r.t. the features (training path) is out of scope")
        return None, _ApplyFeatTranspose.apply(feat, grad, tuple(filt.shape[-2:]))


class _ApplyFeatTranspose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, inp, ksz):
        f5 = _as5d(feat)
        S = f5.shape[1]
        out = torch.stack([adj_raw(f5[:, s], inp[:, s], ksz) for s in range(S)], dim=0)
        ctx.save_for_backward(feat)
        ctx.ksz = ksz
        ctx.inp_hw = tuple(inp.shape[-2:])
        return out

    @staticmethod
    def backward(ctx, grad):
        (feat,) = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("gradient of apply_feat_transpose w.r.t. the features is out of scope")
        return None, _ApplyFilter.apply(feat, grad), None


def apply_filter(feat, filter, dilation_factors=None):
    """Cross-correlate each sequence's filter with its features (reference: filter.py:5-57)."""
    _require_device(feat, filter)
    if dilation_factors is not None:
        raise NotImplementedError("dilated apply_filter (filter.py:35-52) is not on the hot path")
    num_sequences = feat.shape[1] if feat.dim() == 5 else 1
    if filter.dim() == 5:                                   # (sequences, filters, feat_dim, fH, fW), filter.py:29-34
        assert filter.shape[0] == num_sequences and feat.shape[-3] == filter.shape[-3], "groups != 1 is not covered"
        return _ApplyFilterMF.apply(feat, filter)
    assert filter.shape[0] == num_sequences and feat.shape[-3] == filter.shape[-3]
    if num_sequences == 1 and not (torch.is_grad_enabled() and (feat.requires_grad or filter.requires_grad)):
        # tracking time (one sequence, nothing to differentiate): straight to the kernel -- no autograd.Function frame, no torch.stack
        # (profiles/r06_installed_track_breakdown.txt: this wrapper was the largest host item of ours in an installed track())
        return corr_raw(_as5d(feat)[:, 0], filter[0]).unsqueeze(1)
    return _ApplyFilter.apply(feat, filter)


def apply_feat_transpose(feat, input, filter_ksz, training=True, groups=1):
    """Adjoint of apply_filter w.r.t. the filter (reference: filter.py:91-107; `training` only selected
    between two equivalent conv formulations there)."""
    if groups != 1:
        raise NotImplementedError('Not implemented other values of group.')
    _require_device(feat, input)
    if isinstance(filter_ksz, int):
        filter_ksz = (filter_ksz, filter_ksz)
    if input.dim() == 5:                                    # (images, sequences, filters, H, W), filter.py:158-176
        return _ApplyFeatTransposeMF.apply(feat, input, tuple(filter_ksz))
    if (feat.dim() == 4 or feat.shape[1] == 1) and not (torch.is_grad_enabled() and (feat.requires_grad or input.requires_grad)):
        return adj_raw(_as5d(feat)[:, 0], input[:, 0], tuple(filter_ksz)).unsqueeze(0)
    return _ApplyFeatTranspose.apply(feat, input, tuple(filter_ksz))


def filter_gradient(feat, filter, label=None, training=True):
    """reference: filter.py:203-216."""
    residuals = apply_filter(feat, filter)
    if label is not None:
        residuals = residuals - label
    return apply_feat_transpose(feat, residuals, (filter.shape[-2], filter.shape[-1]), training=training)
