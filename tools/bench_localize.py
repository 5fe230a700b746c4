"""Localisation step timing: pytracking_amd.localization.localize_advanced (one launch + one 32-byte copy) against the
same quantities obtained the stock way on a device tensor (two arg-max searches through torch.max, a clone + masked
fill, and the host reads the reference's method performs: dimp.py:252-281).   python tools/bench_localize.py
"""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import _lib  # noqa: E402
from pytracking_amd import localization as LM  # noqa: E402


class Params:
    target_not_found_threshold, distractor_threshold, hard_negative_threshold = 0.25, 0.8, 0.5
    target_neighborhood_scale, dispalcement_scale = 2.2, 0.8

    def get(self, name, default=None):
        return getattr(self, name, default)


def stock_two_peaks(scores, neigh):
    def max2d(a):
        mr, ar = torch.max(a, dim=-2)
        mv, ac = torch.max(mr, dim=-1)
        return mv, torch.stack((ar.view(ac.numel(), -1)[torch.arange(ac.numel()), ac.view(-1)], ac.view(-1)), -1)
    m1, d1 = max2d(scores)
    _, si = torch.max(m1, dim=0)
    m1 = m1[si]
    d1 = d1[si].float().cpu().view(-1)
    if m1.item() < 0.25:
        return None
    t, b = max(round(d1[0].item() - neigh[0] / 2), 0), min(round(d1[0].item() + neigh[0] / 2 + 1), scores.shape[-2])
    l, r = max(round(d1[1].item() - neigh[1] / 2), 0), min(round(d1[1].item() + neigh[1] / 2 + 1), scores.shape[-1])
    masked = scores[si:si + 1].clone()
    masked[..., t:b, l:r] = 0
    m2, d2 = max2d(masked)
    return m1.item(), d1, m2.item(), d2.float().cpu().view(-1)


def main():
    if _lib.needs_build():
        _lib.build_library()
    dev = torch.device("cuda", 0)
    H = W = 19
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    scores = (torch.exp(-((yy - 9.0) ** 2 + (xx - 11.0) ** 2) / 3.0) + 0.02 * torch.randn(H, W))[None].to(dev)
    me = types.SimpleNamespace(params=Params(), kernel_size=torch.Tensor([4, 4]), output_window=None,
                               img_support_sz=torch.Tensor([288.0, 288.0]), target_sz=torch.Tensor([60.0, 80.0]),
                               pos=torch.Tensor([150.0, 160.0]))
    sp, ss = torch.Tensor([[150.0, 160.0]]), torch.Tensor([1.0])
    out = {}
    for tag, fn in (("fused_us", lambda: LM.localize_advanced(me, scores, sp, ss)),
                    ("stock_torch_us", lambda: stock_two_peaks(scores, (8.25, 11.0)))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            fn()
        torch.cuda.synchronize()
        out[tag] = round((time.perf_counter() - t0) / 300 * 1e6, 1)
    out["workload"] = "localize_advanced on one 19x19 score map, host wall time per call incl. synchronisations"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
