"""Which host-side torch op issues the device-to-device copies seen in the ToMP frame profile?  (measurement aid)"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import synth  # noqa: E402
from pytracking_amd import transformer as TM  # noqa: E402

dev = torch.device("cuda", 0)
cfg = synth.TOMP
tr = TM.Transformer(d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048)
pred = TM.FilterPredictor(tr, feature_sz=18).to(dev).eval()
cls = TM.LinearFilterClassifier(256).to(dev).eval()
reg = TM.DenseBoxRegressor(256).to(dev).eval()
train, test, lab, ltrb = [torch.from_numpy(x).to(dev) for x in synth.tomp_inputs(5, cfg)]


def frame():
    cw, bw, cenc, benc = pred.predict_cls_bbreg_filters_parallel(train, test, lab, 1, ltrb)
    return cls(cenc, cw), reg(benc, bw)


with torch.no_grad():
    for _ in range(3):
        frame()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(2):
            frame()
        torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if ("copy" in e.key.lower() or "clone" in e.key.lower()
                                                               or "contiguous" in e.key.lower() or "memcpy" in e.key.lower())]
for e in sorted(rows, key=lambda e: -e.count)[:12]:
    print(e.count, e.key, "|", " <- ".join(s.split("/")[-1] for s in e.stack[:4]))
