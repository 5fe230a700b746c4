"""TEST INFRASTRUCTURE ONLY -- torch (CPU, float64-capable) restatement of the IoU-guided box refinement:
  AtomIoUNet.predict_iou                                   ltr/models/bbreg/atom_iou_net.py:96-136
  LinearBlock (Linear -> BatchNorm2d(eval) -> ReLU)        ltr/models/layers/blocks.py:23-36
  DiMP.optimize_boxes_default / optimize_boxes_relative    pytracking/tracker/dimp/dimp.py:725-788
  ATOM.optimize_boxes                                      pytracking/tracker/atom/atom.py:758-836
  rect_to_rel / rel_to_rect                                ltr/data/bounding_box_utils.py:4-33
PrRoIPool = oracle/prroi_torch.py ("parity unpinned" upstream, self-pinned there).  A floating-point kernel with a
gradient: torch autograd supplies d IoU / d box exactly as the reference obtains it.
Pinned by tests/golden/iou_refine.npz (oracle/make_golden.py: gen_iou_refine runs the reference's own methods).
Parameters: dict keyed by the state_dict names of AtomIoUNet below `bb_regressor.`.
"""
import torch

from oracle.prroi_torch import prroi_pool2d


def _linear_block(p, pre, x):
    y = x.reshape(x.shape[0], -1) @ p[pre + "linear.weight"].t() + p[pre + "linear.bias"]
    y = (y - p[pre + "bn.running_mean"]) / torch.sqrt(p[pre + "bn.running_var"] + 1e-5) * p[pre + "bn.weight"] + p[pre + "bn.bias"]
    return torch.relu(y)


def predict_iou(p, modulation, feat, proposals):
    """modulation: (mod3 (1,C3), mod4 (1,C4)); feat: (c3_t (1,C3,H3,W3), c4_t (1,C4,H4,W4)); proposals (1,P,4) xywh."""
    c3 = feat[0] * modulation[0].reshape(1, -1, 1, 1)
    c4 = feat[1] * modulation[1].reshape(1, -1, 1, 1)
    xyxy = torch.cat((proposals[0, :, 0:2], proposals[0, :, 0:2] + proposals[0, :, 2:4]), dim=1)
    roi = torch.cat((torch.zeros(xyxy.shape[0], 1, dtype=xyxy.dtype), xyxy), dim=1)
    y3 = _linear_block(p, "fc3_rt.", prroi_pool2d(c3, roi, 5, 5, 1 / 8))
    y4 = _linear_block(p, "fc4_rt.", prroi_pool2d(c4, roi, 3, 3, 1 / 16))
    return (torch.cat((y3, y4), dim=1) @ p["iou_predictor.weight"].t() + p["iou_predictor.bias"]).reshape(1, -1)


def rect_to_rel(bb, sz_norm):
    return torch.cat(((bb[..., :2] + 0.5 * bb[..., 2:]) / sz_norm, torch.log(bb[..., 2:])), dim=-1)


def rel_to_rect(bb, sz_norm):
    sz = torch.exp(bb[..., 2:])
    return torch.cat((bb[..., :2] * sz_norm - 0.5 * sz, sz), dim=-1)


def refine(p, modulation, feat, init_boxes, num_iter, step_length, step_decay, relative):
    """-> (boxes (P,4), iou of the last forward pass (P,)); dimp.py:734-788."""
    boxes = init_boxes.reshape(1, -1, 4).clone()
    step = step_length
    if relative:
        sz_norm = boxes[:, :1, 2:].clone()
        rel = rect_to_rel(boxes, sz_norm)
    out = None
    for _ in range(num_iter):
        if relative:
            var = rel.clone().detach().requires_grad_(True)
            out = predict_iou(p, modulation, feat, rel_to_rect(var, sz_norm))
        else:
            var = boxes.clone().detach().requires_grad_(True)
            out = predict_iou(p, modulation, feat, var)
        out.backward(gradient=torch.ones_like(out))
        if relative:
            rel = (var + step * var.grad).detach()
        else:
            boxes = (var + step * var.grad * var[:, :, 2:].repeat(1, 1, 2)).detach()
        step = step * step_decay
    if relative:
        boxes = rel_to_rect(rel, sz_norm)
    return boxes.reshape(-1, 4).detach(), out.detach().reshape(-1)


def refine_atom(p, modulation, feat, init_boxes, num_iter, step_length, step_decay, relative):
    """ATOM.optimize_boxes (atom.py:758-836): per-proposal step length; a proposal whose predicted IoU did not improve
    shrinks its step length and takes the previous step back.  Also returns how many (iteration, proposal) pairs
    backtracked, so that tests can check the branch was exercised."""
    boxes = init_boxes.reshape(1, -1, 4).clone()
    P = boxes.shape[1]
    slen = step_length * torch.ones(1, P, 1, dtype=boxes.dtype)
    prev = -99999999 * torch.ones(1, P, dtype=boxes.dtype)
    step = torch.zeros_like(boxes)
    if relative:
        sz_norm = boxes[:, :1, 2:].clone()
        var0 = rect_to_rel(boxes, sz_norm)
    else:
        var0 = boxes
    out, backtracks = None, 0
    for _ in range(num_iter):
        var = var0.clone().detach().requires_grad_(True)
        out = predict_iou(p, modulation, feat, rel_to_rect(var, sz_norm) if relative else var)
        out.backward(gradient=torch.ones_like(out))
        mask = (out.detach() > prev) | (step_decay >= 1)
        backtracks += int((~mask).sum())
        mf = mask.reshape(1, -1, 1).to(boxes.dtype)
        slen[~mask, :] *= step_decay
        prev = out.detach().clone()
        direction = var.grad if relative else var.grad * var[:, :, 2:].repeat(1, 1, 2)
        step = mf * slen * direction - (1.0 - mf) * step
        var0 = (var + step).detach()
    boxes = rel_to_rect(var0, sz_norm) if relative else var0
    return boxes.reshape(-1, 4).detach(), out.detach().reshape(-1), backtracks
