"""Device-resident state of one synthetic tracking sequence and the per-frame hot-path step
(SURVEY.md section 8d): classify -> arg-max -> memory insert -> steepest-descent solve, executed by
`pt_track_frame_f32` on the current stream with no host synchronisation (graph-capturable).

Used by bench.py, __graft_entry__.smoke() and the GPU tests.  One `TrackState` = one video sequence = one
GPU (sequences are independent).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib, synth
from .filter import _ptr, _stream, device_guarded


class TrackState:
    def __init__(self, cfg, n, seed, device="cuda", kind="dimp"):
        self.cfg, self.n, self.kind = dict(cfg), int(n), kind
        c = self.cfg
        dev = torch.device(device)
        w0, feat, bb, sw = synth.dimp_problem(seed, n, c)
        self.mem_feat = torch.from_numpy(feat).to(dev)
        self.mem_bb = torch.from_numpy(bb).to(dev)
        self.sample_weight = torch.from_numpy(sw).to(dev)
        self.filter = torch.from_numpy(w0 if kind == "dimp" else w0 * 0).to(dev)
        OH = c["H"] + (c["K"] + 1) % 2
        OW = c["W"] + (c["K"] + 1) % 2
        self.scores = torch.zeros(OH, OW, dtype=torch.float32, device=dev)
        self.peak = torch.zeros(2, dtype=torch.float32, device=dev)
        p = _lib.SdParams()
        p.feat_stride = float(c["feat_stride"])
        p.step_length = float(c["init_step_length"])
        p.reg = max(c["init_filter_reg"] ** 2, c["min_filter_reg"] ** 2)
        p.alpha_eps = float(c["alpha_eps"])
        if kind == "dimp":
            p.kind = _lib.PT_SD_DIMP
            # ONE contiguous array (label | mask | spatial): the init stage then requests the tables with its first loads (k_fast_init2: lut3)
            lut3 = torch.from_numpy(np.concatenate((
                synth.gauss_lut(c["num_dist_bins"], c["bin_displacement"], c["init_gauss_sigma"]),
                synth.mask_lut(c["num_dist_bins"], c["bin_displacement"], c["mask_init_factor"], c["mask_act"]),
                np.ones(c["num_dist_bins"], np.float32))).astype(np.float32)).to(dev)
            nb = c["num_dist_bins"]
            self._lut3 = lut3
            self._luts = [lut3[0:nb], lut3[nb:2 * nb], lut3[2 * nb:3 * nb]]
            p.num_bins = c["num_dist_bins"]
            p.bin_displacement = float(c["bin_displacement"])
            p.label_lut, p.mask_lut, p.spatial_lut = (t.data_ptr() for t in self._luts)
            p.mask_act = _lib.PT_MASK_SIGMOID if c["mask_act"] == "sigmoid" else _lib.PT_MASK_LINEAR
            p.score_act = _lib.PT_ACT_RELU
            p.act_param = 1.0
        else:
            p.kind = _lib.PT_SD_PRDIMP
            p.gauss_sigma = float(c["gauss_sigma"])
            p.normalize_label = int(bool(c["normalize_label"]))
            p.uni_weight = float(c["init_uni_weight"] or 0.0)
            p.label_shrink = float(c["label_shrink"])
            p.has_softmax_reg = int(c["softmax_reg"] is not None)
            p.softmax_reg = float(c["softmax_reg"] or 0.0)
            p.label_threshold = float(c["label_threshold"])
        self.params = p
        L = _lib.lib()
        nb = L.pt_track_frame_ws_bytes(n, c["C"], c["H"], c["W"], c["K"])
        if nb == 0:
            raise RuntimeError("pt_track_frame_ws_bytes rejected the configuration")
        self.ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        self.pending = _lib.FramePending()                          # frame chains (pt_track_frame_chain_f32): host-side, zero = nothing pending

    @device_guarded
    def step(self, test_feat, slot, num_iter, defer=False):
        """test_feat (C,H,W) device tensor.  Asynchronous.
        defer=True: frame-chain mode (`pt_track_frame_chain_f32`, include/pt_hot.h): the solve's last filter update stays pending and
        rides on the NEXT step's first correlation; `self.filter` is complete only after `flush()` (or the next non-deferred step).
        A pending update of the previous step is always applied first, whatever `defer` says."""
        c = self.cfg
        assert test_feat.is_contiguous() and test_feat.dtype == torch.float32
        if defer or self.pending.iters > 0:
            rc = _lib.lib().pt_track_frame_chain_f32(
                ctypes.byref(self.params), _ptr(self.filter), _ptr(self.mem_feat), _ptr(self.mem_bb),
                _ptr(self.sample_weight), _ptr(test_feat), int(slot), self.n, c["C"], c["H"], c["W"], c["K"], int(num_iter),
                _ptr(self.scores), _ptr(self.peak), _ptr(self.ws), self.ws.numel(), ctypes.byref(self.pending), int(bool(defer)),
                _stream())
            if rc == _lib.PT_ERR_UNSUPPORTED and self.pending.iters == 0:
                defer = False                                       # shape outside the chain's path: the plain frame below
            else:
                _lib.check(rc, "pt_track_frame_chain_f32")
                return
        rc = _lib.lib().pt_track_frame_f32(
            ctypes.byref(self.params), _ptr(self.filter), _ptr(self.mem_feat), _ptr(self.mem_bb),
            _ptr(self.sample_weight), _ptr(test_feat), int(slot), self.n, c["C"], c["H"], c["W"], c["K"], int(num_iter),
            _ptr(self.scores), _ptr(self.peak), _ptr(self.ws), self.ws.numel(), _stream())
        _lib.check(rc, "pt_track_frame_f32")

    @device_guarded
    def flush(self):
        """Apply a pending last update (end of a chain of `step(..., defer=True)` calls).  Asynchronous; no-op when nothing is pending."""
        c = self.cfg
        rc = _lib.lib().pt_track_frame_flush_f32(ctypes.byref(self.pending), _ptr(self.filter), self.n, c["C"], c["H"], c["W"], c["K"],
                                                 _ptr(self.ws), self.ws.numel(), _stream())
        _lib.check(rc, "pt_track_frame_flush_f32")

    def attach_head(self, weight, norm_scale, norm_eps=1e-5):
        """Classification-feature head in front of the frame: weight (C, Cin, 3, 3) of the final conv
        (ltr/models/target_classifier/features.py:66-72); enables `step_from_backbone`."""
        c = self.cfg
        self.head_w = weight.detach().permute(0, 2, 3, 1).contiguous()        # (C, ky, kx, Cin)
        self.head_cin = int(weight.shape[1])
        self.head_scale, self.head_eps = float(norm_scale), float(norm_eps)
        nb = _lib.lib().pt_track_frame_head_ws_bytes(self.n, self.head_cin, c["C"], c["H"], c["W"], c["K"])
        if nb == 0:
            raise RuntimeError("pt_track_frame_head_ws_bytes rejected the configuration")
        self.ws_head = torch.empty(nb, dtype=torch.uint8, device=self.mem_feat.device)

    @device_guarded
    def step_from_backbone(self, backbone_feat, slot, num_iter):
        """backbone_feat (Cin,H,W): head -> memory slot -> classify -> arg-max -> solve, one C-ABI call.  Asynchronous."""
        c = self.cfg
        assert backbone_feat.is_contiguous() and backbone_feat.dtype == torch.float32
        rc = _lib.lib().pt_track_frame_head_f32(
            ctypes.byref(self.params), _ptr(self.filter), _ptr(self.mem_feat), _ptr(self.mem_bb), _ptr(self.sample_weight),
            _ptr(backbone_feat), _ptr(self.head_w), self.head_scale, self.head_eps, int(slot), self.n, self.head_cin,
            c["C"], c["H"], c["W"], c["K"], int(num_iter), _ptr(self.scores), _ptr(self.peak), _ptr(self.ws_head),
            self.ws_head.numel(), _stream())
        _lib.check(rc, "pt_track_frame_head_f32")

    # algorithmic work of one frame, SURVEY.md section 8(d)
    def bytes_per_solve(self, num_iter):
        c = self.cfg
        return 2 * num_iter * 4 * self.n * c["C"] * c["H"] * c["W"]

    def flops_per_solve(self, num_iter):
        c = self.cfg
        O = c["H"] + (c["K"] + 1) % 2
        return 3 * num_iter * 2 * self.n * c["C"] * c["K"] ** 2 * O * O
