"""ToMP per-frame model prediction timing at BASELINE configs[3] geometry (tomp.py:282-303): 2 memory frames + the test
frame of 256 x 18 x 18 head features -> 972 tokens x 2 batch rows, 6 encoder + 6 decoder layers (8 heads, FFN 2048),
classifier + dense box regressor.   python tools/bench_tomp.py [--reps 50] [--graph]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import _lib, synth  # noqa: E402
from pytracking_amd import transformer as TM  # noqa: E402


def flops(cfg):
    D, ff, L = cfg["D"], cfg["ff"], (cfg["n_train"] + 1) * cfg["H"] * cfg["W"]
    B, HW = 2, cfg["H"] * cfg["W"]
    enc = cfg["n_enc"] * B * (2.0 * L * D * 3 * D + 4.0 * L * L * D + 2.0 * L * D * D + 4.0 * L * D * ff)
    tok = B * (L - HW) * (2.0 * (D // 4) * D + 2.0 * D * D)
    tower = 4 * 2.0 * HW * D * 9 * D + 2.0 * HW * 4 * 9 * D
    return enc + tok + tower


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--graph", action="store_true", help="replay the frame from a captured HIP graph")
    a = ap.parse_args()
    if _lib.needs_build():
        _lib.build_library()
    dev = torch.device("cuda", 0)
    cfg = synth.TOMP
    D = cfg["D"]
    tr = TM.Transformer(d_model=D, nhead=cfg["nhead"], num_encoder_layers=cfg["n_enc"], num_decoder_layers=cfg["n_dec"],
                        dim_feedforward=cfg["ff"])
    pred = TM.FilterPredictor(tr, feature_sz=cfg["feature_sz"]).to(dev).eval()
    cls = TM.LinearFilterClassifier(D).to(dev).eval()
    reg = TM.DenseBoxRegressor(D).to(dev).eval()
    train, test, lab, ltrb = [torch.from_numpy(x).to(dev) for x in synth.tomp_inputs(5, cfg)]

    def frame():
        cw, bw, cenc, benc = pred.predict_cls_bbreg_filters_parallel(train, test, lab, cfg["num_gth_frames"], ltrb)
        return cls(cenc, cw), reg(benc, bw)

    with torch.no_grad():
        for _ in range(3):
            frame()
        torch.cuda.synchronize()
        run = frame
        if a.graph:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                frame()
                with torch.cuda.graph(g, stream=s):
                    out = frame()
            run = g.replay
            run()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
    fl = flops(cfg)
    print(json.dumps({"workload": "ToMP predict_cls_bbreg_filters_parallel + classifier + bbreg, 2+1 frames 256x18x18, 6+6 layers",
                      "graph": bool(a.graph), "ms_per_frame": round(dt * 1e3, 4), "frames_per_s": round(1 / dt, 1),
                      "GFLOP_per_frame": round(fl / 1e9, 2), "TFLOPs": round(fl / dt / 1e12, 2),
                      "frac_of_157_TFLOPs_f32_mfma": round(fl / dt / 157.3e12, 3)}))


if __name__ == "__main__":
    main()
