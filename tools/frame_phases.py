"""Host-side phase times of the DiMP-50 frame after the backbone (tools/bench_dimp_frame_extended.py's frame, all four parts):
where the wall time between two frames goes, as the host sees it.   python tools/frame_phases.py [--frames 300]

  head_call      Python + launches of the classification-feature head (returns without waiting)
  step_call      pt_track_frame_f32: 18 launches (returns without waiting)
  localize_call  launch + wait for the localisation results (= the device finishing head + solver + localisation)
  iou_call       proposals -> refined boxes on the host (17 launches + wait)
"""
import argparse
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import _lib, bench_frame, synth  # noqa: E402
from pytracking_amd import features as FM, iou_refine as IR, localization as LM  # noqa: E402
from tools.bench_dimp_frame_extended import Params, iou_net  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--sync", action="store_true", help="a real stream synchronisation at the end of every frame")
    a = ap.parse_args()
    if _lib.needs_build():
        _lib.build_library()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    cfg = synth.DIMP50
    st = bench_frame.TrackState(cfg, cfg["memory"], seed=1234, device=dev)
    head = FM.residual_bottleneck(feature_dim=256, num_blocks=0, l2norm=True, final_conv=True,
                                  norm_scale=(1.0 / (512 * 16)) ** 0.5, out_dim=512).to(dev).eval()
    backbone_feat = [torch.randn(1, 1024, 18, 18, device=dev) for _ in range(8)]
    net = iou_net(dev)
    iou_feat = (torch.randn(1, 256, 36, 36, device=dev), torch.randn(1, 256, 18, 18, device=dev))
    params = Params(target_not_found_threshold=0.25, distractor_threshold=0.8, hard_negative_threshold=0.5,
                    target_neighborhood_scale=2.2, dispalcement_scale=0.8, box_refinement_iter=5,
                    box_refinement_step_length=1, box_refinement_step_decay=1)
    me = types.SimpleNamespace(params=params, kernel_size=torch.Tensor([4, 4]), output_window=None,
                               img_support_sz=torch.Tensor([288.0, 288.0]), target_sz=torch.Tensor([60.0, 70.0]),
                               pos=torch.Tensor([144.0, 144.0]), net=types.SimpleNamespace(bb_regressor=net),
                               iou_modulation=(torch.rand(1, 256, device=dev) + 0.5, torch.rand(1, 256, device=dev) + 0.5))
    sample_pos, sample_scales = torch.Tensor([[144.0, 144.0]]), torch.Tensor([1.0])
    base = torch.tensor([109.0, 114.0, 70.0, 60.0])
    boxes = torch.stack([base] + [base + torch.cat((torch.rand(2) * 14 - 7, torch.rand(2) * 30 - 15)) for _ in range(9)])
    acc = [0.0] * 4
    pc = time.perf_counter
    per_frame = []

    def frame(i, rec):
        t0 = pc()
        with torch.no_grad():
            x = head(backbone_feat[i % 8])[0]
        t1 = pc()
        st.step(x, slot=i % st.n, num_iter=5)
        t2 = pc()
        LM.localize_advanced(me, st.scores[None], sample_pos, sample_scales)
        t3 = pc()
        IR.optimize_boxes_default(me, iou_feat, boxes)
        t4 = pc()
        if rec:
            for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                acc[k] += d
            per_frame.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
        if a.sync:
            torch.cuda.synchronize()

    for i in range(30):
        frame(i, False)
    torch.cuda.synchronize()
    T0 = pc()
    for i in range(a.frames):
        frame(i, True)
    torch.cuda.synchronize()
    total = (pc() - T0) / a.frames * 1e6
    out = {k: round(v / a.frames * 1e6, 1) for k, v in zip(("head_call", "step_call", "localize_call", "iou_call"), acc)}
    out["frame_us"] = round(total, 1)
    import statistics
    for k, name in enumerate(("head_call", "step_call", "localize_call", "iou_call")):
        v = sorted(1e6 * f[k] for f in per_frame)
        out[name + "_dist"] = {"min": round(v[0], 1), "median": round(statistics.median(v), 1), "p90": round(v[int(0.9 * len(v))], 1),
                               "max": round(v[-1], 1)}
    out["head_call_first_40_frames"] = [round(1e6 * f[0]) for f in per_frame[:40]]
    # the IoU call alone, back to back (what tools/bench_iou.py times)
    for _ in range(10):
        IR.optimize_boxes_default(me, iou_feat, boxes)
    torch.cuda.synchronize()
    t = pc()
    for _ in range(100):
        IR.optimize_boxes_default(me, iou_feat, boxes)
    out["iou_call_alone"] = round((pc() - t) / 100 * 1e6, 1)
    # the localisation alone on a finished score map
    for _ in range(10):
        LM.localize_advanced(me, st.scores[None], sample_pos, sample_scales)
    t = pc()
    for _ in range(200):
        LM.localize_advanced(me, st.scores[None], sample_pos, sample_scales)
    out["localize_call_alone"] = round((pc() - t) / 200 * 1e6, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
