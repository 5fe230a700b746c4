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


_HOST_OUT = {}
_NEG_INF = -float('inf')


def _host_out(device):
    """16 floats of pinned host memory per device that the localisation kernel writes directly (no copy launch)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)       # per (device, stream), like filter.workspace
    buf = _HOST_OUT.get(key)
    if buf is None:
        t = torch.zeros(16, dtype=torch.float32).pin_memory()
        buf = (t, t.numpy(), ctypes.c_void_p(t.data_ptr()))
        _HOST_OUT[key] = buf
    return buf


def tracker_state(self, sample_pos, sample_scales, st=None):
    """`pt_localize_state` (include/pt_hot.h): the host values `localize_advanced` reads from the tracker, unprocessed --
    the float32 arithmetic on them (dimp.py:241-244,268,285,291) happens inside the library in the reference's operation
    order.  Optional thresholds the parameter file does not define are -inf (their test can never fire)."""
    prm = self.params
    st = _lib.LocalizeState() if st is None else st
    st.target_not_found_threshold = prm.target_not_found_threshold
    st.uncertain_threshold = prm.get('uncertain_threshold', _NEG_INF)
    st.hard_sample_threshold = prm.get('hard_sample_threshold', _NEG_INF)
    st.distractor_threshold = prm.distractor_threshold
    st.hard_negative_threshold = prm.hard_negative_threshold
    st.target_neighborhood_scale = prm.target_neighborhood_scale
    st.dispalcement_scale = prm.dispalcement_scale
    st.kernel_size[:] = self.kernel_size.tolist() if isinstance(self.kernel_size, torch.Tensor) else self.kernel_size
    st.img_support_sz[:] = self.img_support_sz.tolist()
    st.target_sz[:] = self.target_sz.tolist()
    st.pos[:] = self.pos.tolist()
    scales = sample_scales.reshape(-1).tolist() if isinstance(sample_scales, torch.Tensor) else list(sample_scales)
    S = len(scales)
    if S > 8:
        raise NotImplementedError("more than 8 scales per call")
    st.sample_scales[:S] = scales
    st.sample_pos[:2 * S] = sample_pos.reshape(-1).tolist()
    return st, S


@device_guarded
def _localize(self, scores, sample_pos, sample_scales):
    _require_device(scores)
    scores_hn = scores
    if self.output_window is not None and self.params.get('perform_hn_without_windowing', False):
        scores_hn = scores.clone()                           # the second peak is searched in the un-windowed map
        scores *= self.output_window
    H, W = scores.shape[-2:]
    sc3 = scores if scores.is_contiguous() else scores.contiguous()
    hn3 = None
    if scores_hn is not scores:
        hn3 = scores_hn if scores_hn.is_contiguous() else scores_hn.contiguous()
    st, S = tracker_state(self, sample_pos, sample_scales)
    if S * H * W != sc3.numel():
        raise ValueError("localize_advanced: one score map per sample scale expected")
    _, host, host_ptr = _host_out(scores.device)
    # launches and returns when the 16 results have landed in the pinned buffer: the one wait of the localisation step
    rc = _lib.lib().pt_localize_advanced_sync_f32(sc3.data_ptr(), None if hn3 is None else hn3.data_ptr(), ctypes.byref(st),
                                                  host_ptr, S, H, W, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "pt_localize_advanced_sync_f32")
    r = torch.from_numpy(host.copy())                        # private copy of the 16 results
    return r[4:6], torch.tensor(int(host[1])), scores_hn, _lib.PT_LOC_FLAGS[int(host[0])], r[2:4]


def localize_advanced(self, scores, sample_pos, sample_scales):
    """Drop-in for `DiMP.localize_advanced` (dimp.py:238-303): (translation_vec, scale_ind, scores, flag)."""
    return _localize(self, scores, sample_pos, sample_scales)[:4]


def localize_advanced_tomp(self, scores, sample_pos, sample_scales):
    """Drop-in for `ToMP.localize_advanced`: the same with the chosen peak's [row, col] appended."""
    return _localize(self, scores, sample_pos, sample_scales)
