"""Host-side mirror of the IoU-guided box refinement: `DiMP.optimize_boxes_default` / `optimize_boxes_relative`
(pytracking/tracker/dimp/dimp.py:725-788) and `ATOM.optimize_boxes` (pytracking/tracker/atom/atom.py:758-836) on
`AtomIoUNet.predict_iou` (ltr/models/bbreg/atom_iou_net.py:96-136).

The two functions are written to be bound as the tracker's methods: they read `self.net.bb_regressor` (the reference's
AtomIoUNet instance -- its `fc3_rt`, `fc4_rt`, `iou_predictor` parameters are packed once and re-packed when they change),
`self.iou_modulation`, `self.params.box_refinement_{iter,step_length,step_decay}` and return the reference's tuple
`(boxes (P,4) on the CPU, iou (P) on the CPU)`.  All iterations run on the device without a host synchronisation; the
reference builds and differentiates an autograd graph per iteration.
"""
import ctypes
import weakref

import torch

from . import _lib
from .filter import _ptr, _require_device, _stream, device_guarded, workspace
from .transformer import _Pack

_CACHE = weakref.WeakKeyDictionary()          # AtomIoUNet instance -> (pack, prepared buffer, key)
_SHAPES = weakref.WeakKeyDictionary()         # AtomIoUNet instance -> {(c3 shape, c4 shape, P): (dims, workspace bytes, parameter count)}
_HOST_OUT = {}                                # (device index, stream) -> pinned result buffer of pt_iou_refine_sync_f32
HOST_FLOATS = 96                              # PT_IOU_HOST_FLOATS (include/pt_hot.h)
SYNC_MAX_P = 16


def _host_out(device):
    """One pinned buffer per (device, stream), like `filter.workspace`: the sequence word is a per-buffer protocol, two trackers on two
    streams of one device must not share it."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _HOST_OUT.get(key)
    if buf is None:
        t = torch.zeros(HOST_FLOATS, dtype=torch.float32).pin_memory()
        buf = (t, ctypes.c_void_p(t.data_ptr()))
        _HOST_OUT[key] = buf
    return buf


def _iou_tensors(net):
    """The 14 parameter / buffer tensors of the IoU head in the order of the packed layout (`iou_layout`, csrc/iou_refine.hip).
    Runs every frame: read through the modules' registries (what `nn.Module.__getattr__` resolves to, without ~40 Python-level
    `__getattr__` calls); anything that is not laid out like the reference's AtomIoUNet takes the attribute path."""
    try:
        mods = net._modules
        out = []
        for name in ("fc3_rt", "fc4_rt"):
            blk = mods[name]._modules
            lin, bn = blk["linear"], blk["bn"]
            out += [lin._parameters["weight"], lin._parameters["bias"], bn._parameters["weight"], bn._parameters["bias"],
                    bn._buffers["running_mean"], bn._buffers["running_var"]]
        head = mods["iou_predictor"]._parameters
        out += [head["weight"], head["bias"]]
        if all(isinstance(t, torch.Tensor) for t in out):
            return out
    except (AttributeError, KeyError, TypeError):
        pass
    out = []
    for blk in (net.fc3_rt, net.fc4_rt):
        out += [blk.linear.weight, blk.linear.bias, blk.bn.weight, blk.bn.bias, blk.bn.running_mean, blk.bn.running_var]
    return out + [net.iou_predictor.weight, net.iou_predictor.bias]


def _packs(net, dims):
    st = _CACHE.get(net)
    if st is None:
        st = {"pack": _Pack(), "prepared": None, "key": None}
        _CACHE[net] = st
    pack = st["pack"].get(_iou_tensors(net))
    if st["key"] != st["pack"].key:
        L = _lib.lib()
        st["prepared"] = torch.empty(L.pt_iou_prepared_floats(ctypes.byref(dims)), dtype=torch.float32, device=pack.device)
        _lib.check(L.pt_iou_prepare_f32(ctypes.byref(dims), _ptr(pack), _ptr(st["prepared"]), _stream()),
                   "pt_iou_prepare_f32")
        st["key"] = st["pack"].key
    return pack, st["prepared"]


@device_guarded
def refine_boxes(net, modulation, iou_features, init_boxes, num_iter, step_length, step_decay, relative, backtrack=False,
                 to_host=False):
    """-> (boxes (P,4), iou (P)) device tensors.  net: AtomIoUNet; modulation: (mod3, mod4) of one target;
    iou_features: (c3_t (1,C3,H3,W3), c4_t (1,C4,H4,W4)); init_boxes (P,4) xywh.
    to_host: the trackers' per-frame call -- CPU tensors are returned; with CPU `init_boxes` and P <= 16 the proposals travel in a
    kernel argument block and the results land in pinned host memory (`pt_iou_refine_sync_f32`: no copies, no stream sync)."""
    if net.training or net.fc3_rt.bn is None or net.fc3_rt.relu is None:
        raise NotImplementedError("IoU refinement: eval-mode LinearBlocks with BatchNorm + ReLU (the reference's network)")
    c3, c4 = [f.contiguous() for f in iou_features]
    mod3, mod4 = [m.reshape(-1).contiguous() for m in modulation]
    sync = to_host and init_boxes.device.type == 'cpu' and init_boxes.numel() <= 4 * SYNC_MAX_P
    if sync:
        boxes = init_boxes.reshape(-1, 4).to(torch.float32).contiguous()
        _require_device(c3, c4, mod3, mod4)
    else:
        boxes = init_boxes.reshape(-1, 4).to(c3.device, torch.float32).contiguous()
        _require_device(c3, c4, mod3, mod4, boxes)
    if c3.shape[0] != 1 or c4.shape[0] != 1:
        raise NotImplementedError("IoU refinement runs on one test image (the trackers' call)")
    L = _lib.lib()
    P = boxes.shape[0]
    # everything that depends on the network object and the shapes only is checked once (this runs every frame)
    key = (c3.shape, c4.shape, P)
    shape_cache = _SHAPES.setdefault(net, {})
    hit = shape_cache.get(key)
    if hit is None:
        for pool, size, scale in ((net.prroi_pool3t, 5, 1 / 8), (net.prroi_pool4t, 3, 1 / 16)):
            if (getattr(pool, "pooled_height", size), getattr(pool, "pooled_width", size)) != (size, size) or \
                    abs(getattr(pool, "spatial_scale", scale) - scale) > 1e-12:
                raise NotImplementedError("IoU refinement: pools other than 5x5 @ 1/8 and 3x3 @ 1/16")
        dims = _lib.IouDims(c3.shape[1], c4.shape[1], net.fc3_rt.linear.out_features, net.fc4_rt.linear.out_features,
                            c3.shape[2], c3.shape[3], c4.shape[2], c4.shape[3])
        nb = L.pt_iou_refine_ws_bytes(ctypes.byref(dims), P)
        if nb == 0:
            raise NotImplementedError("IoU refinement: configuration not covered by the gfx950 kernels")
        hit = shape_cache[key] = (dims, nb, L.pt_iou_param_floats(ctypes.byref(dims)))
    dims, nb, n_param = hit
    pack, prepared = _packs(net, dims)
    if pack.numel() != n_param:
        raise ValueError("AtomIoUNet parameters do not match the feature dimensions")
    if isinstance(step_length, (tuple, list)):
        steps = [step_length[0], step_length[0], step_length[1], step_length[1]]
    else:
        steps = [float(step_length)] * 4
    ws = workspace(nb, c3.device)
    if sync:
        host, host_ptr = _host_out(c3.device)
        rc = L.pt_iou_refine_sync_f32(ctypes.byref(dims), _ptr(pack), _ptr(prepared), _ptr(c3), _ptr(c4), _ptr(mod3), _ptr(mod4),
                                      ctypes.c_void_p(boxes.data_ptr()), host_ptr, P, int(num_iter), (ctypes.c_float * 4)(*steps),
                                      float(step_decay), int(bool(relative)), int(bool(backtrack)), _ptr(ws), ws.numel(), _stream())
        if rc != _lib.PT_ERR_UNSUPPORTED:
            _lib.check(rc, "pt_iou_refine_sync_f32")
            return host[:4 * P].clone().view(P, 4), host[64:64 + P].clone()
        # the fused iteration is switched off (PT_IOU_UNFUSED=1) or the stream is being captured: the regular device route
        boxes = boxes.to(c3.device)
    out_boxes = torch.empty_like(boxes)
    out_iou = torch.empty(P, dtype=torch.float32, device=c3.device)
    rc = L.pt_iou_refine_f32(ctypes.byref(dims), _ptr(pack), _ptr(prepared), _ptr(c3), _ptr(c4), _ptr(mod3), _ptr(mod4),
                             _ptr(boxes), _ptr(out_boxes), _ptr(out_iou), P, int(num_iter), (ctypes.c_float * 4)(*steps),
                             float(step_decay), int(bool(relative)), int(bool(backtrack)), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "pt_iou_refine_f32")
    if to_host:
        return out_boxes.cpu(), out_iou.cpu()
    return out_boxes, out_iou


def _optimize(self, iou_features, init_boxes, relative):
    p = self.params
    boxes, iou = refine_boxes(self.net.bb_regressor, self.iou_modulation, iou_features, init_boxes, p.box_refinement_iter,
                              p.box_refinement_step_length, p.box_refinement_step_decay, relative, to_host=True)
    return boxes.view(-1, 4), iou.view(-1)


def optimize_boxes_atom(self, iou_features, init_boxes):
    """Drop-in for `ATOM.optimize_boxes` (pytracking/tracker/atom/atom.py:758-836): the network is `self.iou_predictor`, the
    modulation `self.target_feat`, and every proposal backtracks on its own when its predicted IoU stops improving."""
    p = self.params
    space = p.get('box_refinement_space', 'default')
    if space not in ('default', 'relative'):
        raise ValueError('Unknown box_refinement_space {}'.format(space))
    boxes, iou = refine_boxes(self.iou_predictor, self.target_feat, iou_features, init_boxes, p.box_refinement_iter,
                              p.box_refinement_step_length, p.box_refinement_step_decay, space == 'relative', backtrack=True,
                              to_host=True)
    return boxes.view(-1, 4), iou.view(-1)


def optimize_boxes_default(self, iou_features, init_boxes):
    """Drop-in for `DiMP.optimize_boxes_default` (dimp.py:734-759)."""
    return _optimize(self, iou_features, init_boxes, False)


def optimize_boxes_relative(self, iou_features, init_boxes):
    """Drop-in for `DiMP.optimize_boxes_relative` (dimp.py:762-795)."""
    return _optimize(self, iou_features, init_boxes, True)
