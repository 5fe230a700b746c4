"""Host-side mirror of the score-map localisation: `dcf.max2d` (pytracking/libs/dcf.py:156-164) and the body of
`DiMP.localize_advanced` / `ToMP.localize_advanced` (pytracking/tracker/dimp/dimp.py:238-303, tracker/tomp/tomp.py).

`localize_advanced(self, scores, sample_pos, sample_scales)` is written to be bound as the tracker's method: it reads
the same attributes (`params`, `kernel_size`, `img_support_sz`, `target_sz`, `pos`, `output_window`) and returns the same
tuple, but finds both peaks in ONE device launch and makes ONE device-to-host copy of 8 floats, where the reference
issues two max2d's, a clone, a masked fill and about ten `.item()` / `.cpu()` synchronisations.  The decision ladder
(not_found / uncertain / hard_negative / normal) is the reference's, evaluated on those 8 numbers with the same
float32 tensor arithmetic.
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


def _localize(self, scores, sample_pos, sample_scales):
    sz = scores.shape[-2:]
    score_sz = torch.Tensor(list(sz))
    output_sz = score_sz - (self.kernel_size + 1) % 2
    score_center = (score_sz - 1) / 2

    scores_hn = scores
    if self.output_window is not None and self.params.get('perform_hn_without_windowing', False):
        scores_hn = scores.clone()                           # dimp.py:247-250, verbatim (device elementwise)
        scores *= self.output_window

    S = scores.shape[0]
    scale_of = lambda s: sample_scales[s]
    neigh = [self.params.target_neighborhood_scale * (self.target_sz / scale_of(s)) * (output_sz / self.img_support_sz)
             for s in range(S)]                              # dimp.py:268 for every candidate scale
    v = two_peaks(scores, scores_hn, neigh)

    scale_ind = v[3].long()
    sample_scale = sample_scales[scale_ind]
    max_score1 = v[0]
    max_disp1 = v[1:3].clone()
    target_disp1 = max_disp1 - score_center
    translation_vec1 = target_disp1 * (self.img_support_sz / output_sz) * sample_scale

    if max_score1.item() < self.params.target_not_found_threshold:
        return translation_vec1, scale_ind, scores_hn, 'not_found', max_disp1
    if max_score1.item() < self.params.get('uncertain_threshold', -float('inf')):
        return translation_vec1, scale_ind, scores_hn, 'uncertain', max_disp1
    if max_score1.item() < self.params.get('hard_sample_threshold', -float('inf')):
        return translation_vec1, scale_ind, scores_hn, 'hard_negative', max_disp1

    max_score2 = v[4]
    max_disp2 = v[5:7].clone()
    target_disp2 = max_disp2 - score_center
    translation_vec2 = target_disp2 * (self.img_support_sz / output_sz) * sample_scale

    prev_target_vec = (self.pos - sample_pos[scale_ind, :]) / ((self.img_support_sz / output_sz) * sample_scale)

    if max_score2 > self.params.distractor_threshold * max_score1:
        disp_norm1 = torch.sqrt(torch.sum((target_disp1 - prev_target_vec) ** 2))
        disp_norm2 = torch.sqrt(torch.sum((target_disp2 - prev_target_vec) ** 2))
        disp_threshold = self.params.dispalcement_scale * math.sqrt(sz[0] * sz[1]) / 2

        if disp_norm2 > disp_threshold and disp_norm1 < disp_threshold:
            return translation_vec1, scale_ind, scores_hn, 'hard_negative', max_disp1
        if disp_norm2 < disp_threshold and disp_norm1 > disp_threshold:
            return translation_vec2, scale_ind, scores_hn, 'hard_negative', max_disp2
        if disp_norm2 > disp_threshold and disp_norm1 > disp_threshold:
            return translation_vec1, scale_ind, scores_hn, 'uncertain', max_disp1
        return translation_vec1, scale_ind, scores_hn, 'uncertain', max_disp1

    if max_score2 > self.params.hard_negative_threshold * max_score1 and max_score2 > self.params.target_not_found_threshold:
        return translation_vec1, scale_ind, scores_hn, 'hard_negative', max_disp1

    return translation_vec1, scale_ind, scores_hn, 'normal', max_disp1


def localize_advanced(self, scores, sample_pos, sample_scales):
    """Drop-in for `DiMP.localize_advanced` (dimp.py:238-303): (translation_vec, scale_ind, scores, flag)."""
    return _localize(self, scores, sample_pos, sample_scales)[:4]


def localize_advanced_tomp(self, scores, sample_pos, sample_scales):
    """Drop-in for `ToMP.localize_advanced`: the same with the chosen peak's [row, col] appended."""
    return _localize(self, scores, sample_pos, sample_scales)
