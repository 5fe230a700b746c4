"""Experiment: phase time stamps of the fused IoU-refinement kernels (library variant built with -DPT_IOU_STAMPS):
   PT_VARIANT_SRC=iou_refine tools/build_variant.sh ioustamps -DPT_IOU_STAMPS
   PT_HOT_LIB=pytracking_amd/variants/libpt_hot_ioustamps.so python tools/exp_iou_stamps.py
Every wave records the 100 MHz device clock at up to 8 points; printed per stamp: min / mean / max over all waves, in us since the first
wave of the kernel entered."""
import ctypes
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import _lib  # noqa: E402
from pytracking_amd import iou_refine as IR  # noqa: E402
import bench_iou  # noqa: E402

NAMES = {"k_iou_fwd": ["entry", "geometry table + barrier", "pooled values written", "barrier", "end", "windows requested", "round 0 value", "all global loads landed"],
         "k_iou_bwd": ["entry", "geometry table + barrier", "windows requested", "matrix product done", "barrier", "end"]}


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = bench_iou.Net().to(dev).eval()
    feat = (torch.randn(1, 256, 36, 36, device=dev), torch.randn(1, 256, 18, 18, device=dev))
    mod = (torch.rand(1, 256, device=dev) + 0.5, torch.rand(1, 256, device=dev) + 0.5)
    base = torch.tensor([100.0, 90.0, 80.0, 110.0])
    boxes = torch.stack([base] + [base + torch.cat((torch.rand(2) * 20 - 10, torch.rand(2) * 40 - 20)) for _ in range(9)])
    params = types.SimpleNamespace(box_refinement_iter=5, box_refinement_step_length=1.0, box_refinement_step_decay=1)
    me = types.SimpleNamespace(params=params, net=types.SimpleNamespace(bb_regressor=net), iou_modulation=mod)
    L = _lib.lib()
    L.pt_debug_set_iou_stamps.argtypes = [ctypes.c_void_p]
    nwg = 100 + 36
    for _ in range(5):
        IR.optimize_boxes_default(me, feat, boxes)
    acc = []
    for rep in range(8):
        buf = torch.zeros(2 * nwg * 16 * 8, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        L.pt_debug_set_iou_stamps(ctypes.c_void_p(buf.data_ptr()))
        IR.optimize_boxes_default(me, feat, boxes)
        torch.cuda.synchronize()
        L.pt_debug_set_iou_stamps(None)
        acc.append(buf.cpu().numpy().reshape(2, nwg, 16, 8).astype(np.float64))
    for ki, name in enumerate(("k_iou_fwd", "k_iou_bwd")):
        ts = []
        for a in acc:
            t = a[ki].copy()
            t[t == 0] = np.nan
            ts.append((t - np.nanmin(t[:, :, 0])) * 0.01)
        t = np.nanmean(np.stack(ts), axis=0)
        print(name)
        for k, label in enumerate(NAMES[name]):
            v = t[:, :, k]
            if np.all(np.isnan(v)):
                continue
            print(f"  {label:40s} {np.nanmin(v):6.2f} / {np.nanmean(v):6.2f} / {np.nanmax(v):6.2f}")
        for lvl, sl in (("level 3 chunks", slice(0, 100)), ("level 4 chunks", slice(100, 136))):
            print(f"  {lvl}: mean end {np.nanmean(np.nanmax(t[sl, :, :], axis=2)):.2f} us")
        one = ts[-1]                                                    # a single run: which workgroups are late
        end = np.nanmax(one, axis=2)                                    # (chunk, wave slot)
        for grp in range(1):
            e = np.nanmax(end[:, 8 * grp:8 * grp + 8], axis=1)
            print(f"  group {grp}: end by XCD (chunk % 8): " + " ".join(f"{np.nanmean(e[x::8]):.1f}" for x in range(8)))
            if name == "k_iou_fwd":
                for z in (3, 50, 110):
                    print(f"    chunk {z}: per wave [windows requested, loads landed, round 0 value, pooled written]: " +
                          " | ".join(f"{one[z, wv, 5]:.2f} {one[z, wv, 7]:.2f} {one[z, wv, 6]:.2f} {one[z, wv, 2]:.2f}" for wv in range(8)))
            worst = np.argsort(-np.nan_to_num(e))[:6]
            for z in worst:
                print(f"    chunk {z:3d}: " + " | ".join(" ".join(f"{v:5.2f}" for v in one[z, 8 * grp + wv, :8]) for wv in (0, 4, 7)))


if __name__ == "__main__":
    main()
