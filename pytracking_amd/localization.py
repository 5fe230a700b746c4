"""Host-side mirror of the score-map localisation: `dcf.max2d` (pytracking/libs/dcf.py:156-164) and the body of
`DiMP.localize_advanced` / `ToMP.localize_advanced` (pytracking/tracker/dimp/dimp.py:238-303, tracker/tomp/tomp.py).

`localize_advanced(self, scores, sample_pos, sample_scales)` is written to be bound as the tracker's method: it reads
the same attributes (`params`, `kernel_size`, `img_support_sz`, `target_sz`, `pos`, `output_window`) and returns the same
tuple.  The whole routine is ONE launch (`pt_localize_decide_f32`, csrc/localize.hip): both peaks, the outcome
(normal / hard_negative / uncertain / not_found), the displacement of the peak that outcome selects and its translation
vector are decided on the device and written into a pinned host buffer; the host forms the per-frame constants the
kernel needs (float32 scalar arithmetic on the tracker's CPU state -- numpy float32 rounds like torch's CPU kernels)
and waits for the stream once.  The reference issues two max2d's, a clone, a masked fill and about ten `.item()` /
`.cpu()` synchronisations for the same result.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from .filter import _ptr, _require_device, _stream, device_guarded


@device_guarded
def max2d(a: torch.Tensor):
    """Maximum and [row, col] arg-max over the last two dimensions (dcf.py:156-164); stays on the device."""
    _require_device(a)
    H, W = a.shape[-2:]
    lead = a.shape[:-2]
    n = max(int(torch.Size(lead).numel()), 1)
    a3 = a.reshape(n, H, W).contiguous()
    mv = torch.empty(n, dtype=torch.float32, device=a.device)
    am = torch.empty(n, 2, dtype=torch.int64, device=a.device)
    _lib.check(_lib.lib().pt_max2d_f32(_ptr(a3), _ptr(mv), _ptr(am), n, H, W, _stream()), "pt_max2d_f32")
    return mv.reshape(lead), am.reshape(*lead, 2)


@device_guarded
def two_peaks(scores, scores_hn, neigh):
    """scores (S,H,W); neigh: S pairs (rows, cols) -> CPU float tensor [max1,row1,col1,scale,max2,row2,col2,0]."""
    _require_device(scores)
    S, H, W = scores.shape
    scores = scores.contiguous()
    hn = None if scores_hn is None or scores_hn is scores else scores_hn.contiguous()
    nr = (ctypes.c_float * S)(*[float(v[0]) for v in neigh])
    nc = (ctypes.c_float * S)(*[float(v[1]) for v in neigh])
    out = torch.empty(8, dtype=torch.float32, device=scores.device)
    rc = _lib.lib().pt_localize_f32(_ptr(scores), None if hn is None else _ptr(hn), nr, nc, _ptr(out), S, H, W, _stream())
    _lib.check(rc, "pt_localize_f32")
    return out.cpu()                                         # the one synchronisation of the localisation step


_F = np.float32
_HOST_OUT = {}


def _host_out(device):
    """16 floats of pinned host memory per device that the localisation kernel writes directly (no copy launch)."""
    buf = _HOST_OUT.get(device.index)
    if buf is None:
        t = torch.zeros(16, dtype=torch.float32).pin_memory()
        buf = (t, t.numpy(), ctypes.c_void_p(t.data_ptr()))
        _HOST_OUT[device.index] = buf
    return buf


def _f32_pair(t):
    """A two-element CPU tensor (or sequence) as two numpy float32 scalars."""
    a, b = t.tolist() if isinstance(t, torch.Tensor) else t
    return _F(a), _F(b)


def _frame_constants(self, shape, sample_pos, sample_scales):
    """`pt_localize_params` for this frame.  Every product / quotient below is a float32 operation in the reference
    (float32 CPU tensors; Python scalars are rounded to float32 by torch's binary ops): numpy float32 scalars round
    identically.  Optional thresholds the parameter file does not define are -inf (the test can never fire)."""
    S, H, W = shape
    prm = self.params
    q = _lib.LocalizeParams()
    q.target_not_found_threshold = float(prm.target_not_found_threshold)
    q.uncertain_threshold = float(prm.get('uncertain_threshold', -float('inf')))
    q.hard_sample_threshold = float(prm.get('hard_sample_threshold', -float('inf')))
    q.distractor_threshold = float(prm.distractor_threshold)
    q.hard_negative_threshold = float(prm.hard_negative_threshold)
    q.target_not_found_f32 = float(prm.target_not_found_threshold)
    q.disp_threshold = prm.dispalcement_scale * math.sqrt(H * W) / 2
    kr, kc = _f32_pair(self.kernel_size)
    out_r, out_c = _F(H) - (kr + _F(1)) % _F(2), _F(W) - (kc + _F(1)) % _F(2)      # output_sz
    q.center_r, q.center_c = (_F(H) - _F(1)) / _F(2), (_F(W) - _F(1)) / _F(2)
    sup_r, sup_c = _f32_pair(self.img_support_sz)
    ratio_r, ratio_c = sup_r / out_r, sup_c / out_c
    q.ratio_r, q.ratio_c = ratio_r, ratio_c
    tns = _F(prm.target_neighborhood_scale)
    tgt_r, tgt_c = _f32_pair(self.target_sz)
    pos_r, pos_c = _f32_pair(self.pos)
    scales = sample_scales.tolist() if isinstance(sample_scales, torch.Tensor) else list(sample_scales)
    centres = sample_pos.tolist()
    for s in range(S):
        sc = _F(scales[s])
        q.scale[s] = sc
        q.neigh_r[s] = tns * (tgt_r / sc) * (out_r / sup_r)
        q.neigh_c[s] = tns * (tgt_c / sc) * (out_c / sup_c)
        q.prev_r[s] = (pos_r - _F(centres[s][0])) / (ratio_r * sc)
        q.prev_c[s] = (pos_c - _F(centres[s][1])) / (ratio_c * sc)
    return q


@device_guarded
def _localize(self, scores, sample_pos, sample_scales):
    _require_device(scores)
    scores_hn = scores
    if self.output_window is not None and self.params.get('perform_hn_without_windowing', False):
        scores_hn = scores.clone()                           # the second peak is searched in the un-windowed map
        scores *= self.output_window
    H, W = scores.shape[-2:]
    sc3 = scores.reshape(-1, H, W)
    S = sc3.shape[0]
    if S > 8:
        raise NotImplementedError("more than 8 scales per call")
    if not sc3.is_contiguous():
        sc3 = sc3.contiguous()
    hn3 = None
    if scores_hn is not scores:
        hn3 = scores_hn.reshape(-1, H, W)
        hn3 = hn3 if hn3.is_contiguous() else hn3.contiguous()
    q = _frame_constants(self, (S, H, W), sample_pos, sample_scales)
    _, host, host_ptr = _host_out(scores.device)
    stream = torch.cuda.current_stream()
    rc = _lib.lib().pt_localize_decide_f32(_ptr(sc3), None if hn3 is None else _ptr(hn3), ctypes.byref(q), host_ptr, S, H, W,
                                           ctypes.c_void_p(stream.cuda_stream))
    _lib.check(rc, "pt_localize_decide_f32")
    stream.synchronize()                                     # the one synchronisation of the localisation step
    r = host.tolist()
    translation_vec = torch.tensor(r[4:6], dtype=torch.float32)
    max_disp = torch.tensor(r[2:4], dtype=torch.float32)
    return translation_vec, torch.tensor(int(r[1])), scores_hn, _lib.PT_LOC_FLAGS[int(r[0])], max_disp


def localize_advanced(self, scores, sample_pos, sample_scales):
    """Drop-in for `DiMP.localize_advanced` (dimp.py:238-303): (translation_vec, scale_ind, scores, flag)."""
    return _localize(self, scores, sample_pos, sample_scales)[:4]


def localize_advanced_tomp(self, scores, sample_pos, sample_scales):
    """Drop-in for `ToMP.localize_advanced`: the same with the chosen peak's [row, col] appended."""
    return _localize(self, scores, sample_pos, sample_scales)
