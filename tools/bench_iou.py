"""IoU-guided box refinement timing at the trackers' sizes (AtomIoUNet defaults: 256-channel IoU features 36x36 and
18x18, 10 proposals): DiMP-50 = 5 iterations in the default parametrisation, PrDiMP-50 = 10 iterations in the relative
one.  Fused device-side loop (pytracking_amd.iou_refine) against the stock formulation the reference runs -- one autograd
forward + backward of predict_iou per iteration -- on the same HIP PrRoIPool.   python tools/bench_iou.py
"""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import _lib  # noqa: E402
from pytracking_amd import iou_refine as IR  # noqa: E402
from pytracking_amd.prroi_pool import PrRoIPool2D  # noqa: E402


class Net(torch.nn.Module):
    def __init__(self, C=256, I=256):
        super().__init__()

        def block(k):
            m = torch.nn.Module()
            m.linear, m.bn, m.relu = torch.nn.Linear(C * k * k, I), torch.nn.BatchNorm2d(I), torch.nn.ReLU()
            return m
        self.fc3_rt, self.fc4_rt = block(5), block(3)
        self.iou_predictor = torch.nn.Linear(2 * I, 1)
        self.prroi_pool3t, self.prroi_pool4t = PrRoIPool2D(5, 5, 1 / 8), PrRoIPool2D(3, 3, 1 / 16)

    def predict_iou(self, mod, feat, proposals):                  # atom_iou_net.py:96-136 for one image
        c3, c4 = feat[0] * mod[0].reshape(1, -1, 1, 1), feat[1] * mod[1].reshape(1, -1, 1, 1)
        xyxy = torch.cat((proposals[0, :, :2], proposals[0, :, :2] + proposals[0, :, 2:]), 1)
        roi = torch.cat((torch.zeros_like(xyxy[:, :1]), xyxy), 1)
        out = []
        for blk, pool, c in ((self.fc3_rt, self.prroi_pool3t, c3), (self.fc4_rt, self.prroi_pool4t, c4)):
            y = blk.linear(pool(c, roi).reshape(roi.shape[0], -1))
            out.append(blk.relu(blk.bn(y.reshape(*y.shape, 1, 1))).reshape(y.shape))
        return self.iou_predictor(torch.cat(out, 1)).reshape(1, -1)


def stock_default(net, mod, feat, boxes, iters, step):
    b = boxes.view(1, -1, 4)
    for _ in range(iters):
        v = b.clone().detach().requires_grad_(True)
        out = net.predict_iou(mod, feat, v)
        out.backward(gradient=torch.ones_like(out))
        b = (v + step * v.grad * v[:, :, 2:].repeat(1, 1, 2)).detach()
    return b.view(-1, 4).cpu(), out.detach().view(-1).cpu()


def measure(dev, with_stock=True, reps=50):
    torch.manual_seed(0)
    net = Net().to(dev).eval()
    feat = (torch.randn(1, 256, 36, 36, device=dev), torch.randn(1, 256, 18, 18, device=dev))
    mod = (torch.rand(1, 256, device=dev) + 0.5, torch.rand(1, 256, device=dev) + 0.5)
    base = torch.tensor([100.0, 90.0, 80.0, 110.0])
    boxes = torch.stack([base] + [base + torch.cat((torch.rand(2) * 20 - 10, torch.rand(2) * 40 - 20)) for _ in range(9)])
    out = {}
    for tag, iters, step, rel in (("dimp50_default_5it", 5, 1.0, False), ("prdimp50_relative_10it", 10, 2.5e-3, True)):
        params = types.SimpleNamespace(box_refinement_iter=iters, box_refinement_step_length=step, box_refinement_step_decay=1)
        me = types.SimpleNamespace(params=params, net=types.SimpleNamespace(bb_regressor=net), iou_modulation=mod)
        fn = IR.optimize_boxes_relative if rel else IR.optimize_boxes_default
        runs = {"fused_us": lambda: fn(me, feat, boxes)}
        if not rel and with_stock:
            runs["stock_autograd_us"] = lambda: stock_default(net, mod, feat, boxes.to(dev), iters, step)
        for name, f in runs.items():
            for _ in range(5):
                r = f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = f()
            torch.cuda.synchronize()
            out[f"{tag}_{name}"] = round((time.perf_counter() - t0) / reps * 1e6, 1)
        if not rel and with_stock:
            a, b = fn(me, feat, boxes), stock_default(net, mod, feat, boxes.to(dev), iters, step)
            out[f"{tag}_max_box_diff"] = float((a[0] - b[0]).abs().max())
    out["workload"] = ("10 proposals (host memory), 256-channel IoU features 36x36 / 18x18; host wall time per call until the refined boxes "
                       "are readable on the host")
    return out


def main():
    if _lib.needs_build():
        _lib.build_library()
    print(json.dumps(measure(torch.device("cuda", 0))))


if __name__ == "__main__":
    main()
