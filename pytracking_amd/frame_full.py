"""Host-side mirror of `pt_track_frame_full_f32` (include/pt_hot.h): the whole DiMP frame behind the backbone -- classification-
feature head, classification + memory insert + re-optimisation, `localize_advanced`, the tracker's glue between localisation and
refinement, IoU-guided refinement of the proposals -- as ONE ctypes call with ONE host wait.

It replaces the call sequence of `DiMP.track` (pytracking/tracker/dimp/dimp.py:95-131): `get_classification_features`,
`classify_target`, `localize_target`, `update_state`, `refine_target_box` (up to and including `optimize_boxes`).  `track()` itself is
Python in the reference, so this entry point is for callers that own their frame loop (bench.py's `dimp50_frame_after_backbone`, an
embedding that wants the 2 host round trips + the Python between them off the critical path); the unmodified tracker classes keep
going through the per-call symbols `install()` rebinds.  The tracker object is read exactly like the rebound methods read it:
`params`, `pos`, `target_sz`, `kernel_size`, `img_support_sz`, `img_sample_sz`, `image_sz`, `net.bb_regressor`, `iou_modulation`.
"""
import ctypes
import os

import torch

from . import _lib, iou_refine as _ir
from .filter import _require_device, device_guarded

_NEG_INF = -float("inf")


class FramePipeline:
    """One sequence: `state` = bench_frame.TrackState with a head attached (filter, sample memory, solver parameters)."""

    def __init__(self, state, num_iter, overlap=False, reordered_update_ok=False, graph=False):
        """overlap: run the localisation + refinement chain on a second stream, concurrently with the steepest-descent iterations
        (`pt_frame_full.aux_stream`, include/pt_hot.h).  That REORDERS the memory update relative to `DiMP.track`
        (dimp.py:139-145 labels the new sample with the refined state; here the label comes from the classification peak, as in the
        synthetic frame of SURVEY.md section 8d), so with `num_iter > 0` the library refuses the call unless `reordered_update_ok=True`
        says the caller wants exactly that.  The reference's order -- update AFTER refinement -- is overlap=False.
        graph: enables `run_graph` -- the frame's launches are captured ONCE into a hipGraph whose per-frame values (memory slot, tracker
        state, thresholds, random numbers, sequence number) are read from a device block (`pt_frame_full.dyn`, include/pt_hot.h); per
        frame the host fills that block, replays the graph and waits for the result word.  Inputs then live in fixed buffers
        (`graph_inputs()`), one stream only."""
        if not hasattr(state, "head_w"):
            raise ValueError("FramePipeline: TrackState.attach_head() first")
        self.st, self.num_iter = state, int(num_iter)
        dev = state.mem_feat.device
        self.loc = _lib.LocalizeState()
        self.glue = _lib.FrameGlue()
        self.ff = f = _lib.FrameFull()
        c = state.cfg
        f.sd = ctypes.pointer(state.params)
        f.filter, f.mem_feat, f.mem_bb = state.filter.data_ptr(), state.mem_feat.data_ptr(), state.mem_bb.data_ptr()
        f.sample_weight, f.head_weight_tap_major = state.sample_weight.data_ptr(), state.head_w.data_ptr()
        f.norm_scale, f.norm_eps = state.head_scale, state.head_eps
        f.n, f.Cin, f.C, f.H, f.W, f.K, f.num_iter = state.n, state.head_cin, c["C"], c["H"], c["W"], c["K"], self.num_iter
        f.scores_out, f.peak_out = state.scores.data_ptr(), state.peak.data_ptr()
        f.loc, f.glue = ctypes.pointer(self.loc), ctypes.pointer(self.glue)
        host = torch.zeros(_lib.PT_FRAME_HOST_FLOATS, dtype=torch.float32).pin_memory()
        self._host, self._host_np, self._host_ptr = host, host.numpy(), ctypes.c_void_p(host.data_ptr())
        self._ws = None
        self._dims_key = None
        self._dev = dev
        self._aux = torch.cuda.Stream(device=dev) if overlap else None
        f.aux_stream = self._aux.cuda_stream if overlap else None
        f.aux_reordered_update_ok = int(bool(reordered_update_ok))
        self._bound = None
        if graph and overlap:
            raise ValueError("FramePipeline: graph replay runs on one stream (overlap=False)")
        self._graph_mode, self._graphs, self._seq, self._gin = bool(graph), {}, 0.0, None
        if graph:
            nb = _lib.lib().pt_track_frame_full_dyn_bytes()
            self._dyn_host = torch.zeros(nb, dtype=torch.uint8).pin_memory()
            # PT_FRAME_DYN_COPY=1 (A/B knob): the block in device memory, refreshed by a copy node at the head of the graph; default: the
            # kernels read the pinned host block directly (three small reads over the fabric instead of a ~10 us copy node in front of
            # everything, profiles/r06i_*)
            self._dyn_dev = torch.zeros(nb, dtype=torch.uint8, device=dev) if os.environ.get("PT_FRAME_DYN_COPY") == "1" else None
            self._gstream = torch.cuda.Stream(device=dev)
        self._call = _lib.lib().pt_track_frame_full_f32
        self._ffref = ctypes.byref(f)
        self._ws_ptr, self._ws_len = None, 0

    def __del__(self):
        try:
            _lib.lib().pt_host_buffer_forget(self._host_ptr)
        except Exception:                                         # noqa: BLE001 -- interpreter shutdown
            pass

    def _bind_iou(self, net, c3, c4, P):
        key = (id(net), c3.shape, c4.shape, P)
        if key != self._dims_key:
            self._dims = _lib.IouDims(c3.shape[1], c4.shape[1], net.fc3_rt.linear.out_features, net.fc4_rt.linear.out_features,
                                      c3.shape[2], c3.shape[3], c4.shape[2], c4.shape[3])
            self.ff.iou_dims = ctypes.pointer(self._dims)
            self.glue.num_random = P - 1
            nb = _lib.lib().pt_track_frame_full_ws_bytes(ctypes.byref(self.ff))
            if nb == 0:
                raise NotImplementedError("pt_track_frame_full_f32: configuration not covered by the gfx950 kernels")
            if self._ws is None or self._ws.numel() < nb:
                self._ws = torch.empty(nb, dtype=torch.uint8, device=self._dev)
            self._ws_ptr, self._ws_len = self._ws.data_ptr(), self._ws.numel()
            self._dims_key = key
        pack, prepared = _ir._packs(net, self._dims)
        self.ff.iou_params, self.ff.iou_prepared = pack.data_ptr(), prepared.data_ptr()

    def bind(self, tracker, iou_features, num_random):
        """Everything of the call that does not change from frame to frame: the tracker's parameters (thresholds, jitter, refinement
        settings, image / sample sizes) and the IoU network's packed weights.  `run` calls it when it sees a new tracker object, a new
        proposal count or new feature shapes; call it yourself after changing `tracker.params` or the network's weights."""
        p = tracker.params
        c3, c4 = iou_features
        net = tracker.net.bb_regressor
        if net.training:
            raise NotImplementedError("IoU refinement: eval-mode network")
        self._bind_iou(net, c3, c4, 1 + num_random)
        f, g, st = self.ff, self.glue, self.loc
        st.target_not_found_threshold = p.target_not_found_threshold
        st.uncertain_threshold = p.get('uncertain_threshold', _NEG_INF)
        st.hard_sample_threshold = p.get('hard_sample_threshold', _NEG_INF)
        st.distractor_threshold = p.distractor_threshold
        st.hard_negative_threshold = p.hard_negative_threshold
        st.target_neighborhood_scale = p.target_neighborhood_scale
        st.dispalcement_scale = p.dispalcement_scale
        st.kernel_size[:] = tracker.kernel_size.tolist() if isinstance(tracker.kernel_size, torch.Tensor) else tracker.kernel_size
        st.img_support_sz[:] = tracker.img_support_sz.tolist()
        g.image_sz[:] = tracker.image_sz.tolist()
        g.img_sample_sz[:] = tracker.img_sample_sz.tolist()
        g.target_inside_ratio = p.get('target_inside_ratio', 0.2)
        g.box_jitter_pos, g.box_jitter_sz = p.box_jitter_pos, p.box_jitter_sz
        g.use_classifier = int(bool(p.get('use_classifier', True)))
        sl = p.box_refinement_step_length
        f.step_length4[:] = [sl[0], sl[0], sl[1], sl[1]] if isinstance(sl, (tuple, list)) else [float(sl)] * 4
        f.step_decay = p.box_refinement_step_decay
        f.iou_iter = p.box_refinement_iter
        f.relative = int(p.get('box_refinement_space', 'default') == 'relative')
        self._bound = (tracker, num_random, c3.shape, c4.shape)
        self._graphs.clear()                                      # captured launches bake these parameters

    @device_guarded
    def run(self, tracker, backbone_feat, slot, iou_features, sample_pos, sample_scales, rand_u):
        """backbone_feat (Cin,H,W) device; iou_features (c3 (1,C3,H3,W3), c4 (1,C4,H4,W4)) device; sample_pos (1,2), sample_scales (1)
        host, as `track()` forms them; rand_u = `torch.rand(num_init_random_boxes, 4)` of this frame (host).
        -> dict(translation_vec (2), scale_ind, flag, pos (2) after update_state, init_box (4), boxes (P,4), iou (P)), CPU tensors.
        With overlap=True the call returns when the REFINEMENT chain has delivered its result block; the steepest-descent tail (filter,
        sample memory, `mem_bb`) may still be running on the current stream -- synchronise that stream before reading them from the host
        or from another stream.  Timings of this mode are return latencies, not frame periods."""
        P = self._fill_frame(tracker, backbone_feat, slot, iou_features, sample_pos, sample_scales, rand_u)
        self.ff.dyn = None
        rc = self._call(self._ffref, self._host_ptr, self._ws_ptr, self._ws_len, torch.cuda.current_stream().cuda_stream)
        if rc:
            _lib.check(rc, "pt_track_frame_full_f32")
        return self._results(P)

    def _fill_frame(self, tracker, backbone_feat, slot, iou_features, sample_pos, sample_scales, rand_u):
        """The per-frame fields of the call: where the target is, where the sample was taken, this frame's random numbers and tensors."""
        c3, c4 = iou_features
        _require_device(backbone_feat, c3, c4)
        num_random = int(rand_u.shape[0]) if rand_u is not None else 0
        b = self._bound
        if b is None or b[0] is not tracker or b[1] != num_random or b[2] != c3.shape or b[3] != c4.shape:
            self.bind(tracker, iou_features, num_random)
        f, g, st = self.ff, self.glue, self.loc
        st.target_sz[:] = tracker.target_sz.tolist()
        st.pos[:] = tracker.pos.tolist()
        scales = sample_scales.reshape(-1).tolist()
        if len(scales) != 1:
            raise NotImplementedError("pt_track_frame_full_f32: one sample scale")
        st.sample_scales[0] = scales[0]
        st.sample_pos[:2] = sample_pos.reshape(-1).tolist()
        if num_random:
            g.rand_u[:4 * num_random] = rand_u.reshape(-1).tolist()
        f.backbone_feat, f.slot = backbone_feat.data_ptr(), int(slot)
        f.c3, f.c4 = c3.data_ptr(), c4.data_ptr()
        mod3, mod4 = tracker.iou_modulation
        f.mod3, f.mod4 = mod3.data_ptr(), mod4.data_ptr()
        return 1 + num_random

    def _results(self, P):
        h = torch.from_numpy(self._host_np.copy())
        return {"translation_vec": h[4:6], "scale_ind": int(h[1]), "flag": _lib.PT_LOC_FLAGS[int(h[0])], "pos": h[16:18],
                "init_box": h[18:22], "boxes": h[32:32 + 4 * P].view(P, 4), "iou": h[96:96 + P], "peak": h[2:4]}

    def graph_inputs(self, backbone_feat, iou_features):
        """The fixed input buffers of graph mode (allocated on first use from the shapes given): (backbone_feat, (c3, c4)).  A caller whose
        backbone writes straight into them passes them to `run_graph` and no copy is made."""
        if self._gin is None or self._gin[0].shape != backbone_feat.shape or self._gin[1].shape != iou_features[0].shape \
                or self._gin[2].shape != iou_features[1].shape:
            self._gin = tuple(torch.empty_like(t).contiguous() for t in (backbone_feat, iou_features[0], iou_features[1]))
            self._graphs.clear()
        return self._gin[0], (self._gin[1], self._gin[2])

    @device_guarded
    def run_graph(self, tracker, backbone_feat, slot, iou_features, sample_pos, sample_scales, rand_u):
        """`run` as ONE graph replay: same arguments, same result.  The launches are captured once per (tracker parameters, num_iter,
        proposal count); per frame: inputs into the fixed buffers (skipped when the caller already wrote them there), the per-frame
        block filled on the host (`pt_track_frame_full_dyn_fill_f32`), replay, wait for the result word."""
        if not self._graph_mode:
            raise RuntimeError("FramePipeline(graph=True) first")
        L = _lib.lib()
        gb, (g3, g4) = self.graph_inputs(backbone_feat, iou_features)
        for dst, src in ((gb, backbone_feat), (g3, iou_features[0]), (g4, iou_features[1])):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        P = self._fill_frame(tracker, gb, slot, (g3, g4), sample_pos, sample_scales, rand_u)
        f = self.ff
        f.dyn = self._dyn_dev.data_ptr() if self._dyn_dev is not None else self._dyn_host.data_ptr()
        self._seq = self._seq + 1.0 if self._seq < 8388000.0 else 1.0
        rc = L.pt_track_frame_full_dyn_fill_f32(self._ffref, self._seq, self._host_ptr, self._ws_ptr, self._ws_len, self._dyn_host.data_ptr())
        if rc:
            f.dyn = None
            _lib.check(rc, "pt_track_frame_full_dyn_fill_f32")
        key = (int(f.num_iter), P, int(f.iou_iter), int(f.relative), float(f.step_decay), tuple(f.step_length4), f.mod3, f.mod4, f.iou_params)
        g = self._graphs.get(key)
        if g is None:
            cur = torch.cuda.current_stream()
            s = self._gstream
            s.wait_stream(cur)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s):
                    if self._dyn_dev is not None:
                        self._dyn_dev.copy_(self._dyn_host, non_blocking=True)   # the copy node that refreshes the per-frame block
                    rc = L.pt_track_frame_full_launch_f32(self._ffref, self._host_ptr, self._ws_ptr, self._ws_len, s.cuda_stream)
            cur.wait_stream(s)
            if rc:
                f.dyn = None
                _lib.check(rc, "pt_track_frame_full_launch_f32 (capture)")
            self._graphs[key] = g
        g.replay()
        f.dyn = None
        rc = L.pt_host_wait_word_f32(self._host_ptr.value + 127 * 4, self._seq, torch.cuda.current_stream().cuda_stream)
        if rc:
            _lib.check(rc, "pt_host_wait_word_f32")
        return self._results(P)
