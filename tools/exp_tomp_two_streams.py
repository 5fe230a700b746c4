"""Experiment (round 6, R): the two batch rows of ToMP's parallel predictor (classification row, box-regression row) are independent
through encoder and decoder.  One B = 2 call (every kernel covers both rows) against two B = 1 calls on two streams of one captured
graph (half-size kernels of the two rows overlapping, each other's launch / prologue / epilogue under the other's work)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import synth  # noqa: E402
from pytracking_amd import transformer as TM  # noqa: E402

dev = torch.device("cuda", 0)
cfg = synth.TOMP
D = cfg["D"]
tr = TM.Transformer(d_model=D, nhead=cfg["nhead"], num_encoder_layers=cfg["n_enc"], num_decoder_layers=cfg["n_dec"], dim_feedforward=cfg["ff"])
pred = TM.FilterPredictor(tr, feature_sz=cfg["feature_sz"]).to(dev).eval()
train, test, lab, ltrb = [torch.from_numpy(x).to(dev) for x in synth.tomp_inputs(5, cfg)]
main = torch.cuda.Stream(device=dev)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def one_call():
    return pred.predict_cls_bbreg_filters_parallel(train, test, lab, cfg["num_gth_frames"], ltrb)


def one_row():
    return pred.predict_filter(train, test, lab, ltrb)


def two_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = pred.predict_filter(train, test, lab, ltrb)
    with torch.cuda.stream(s2):
        b = pred.predict_filter(train, test, lab, ltrb)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b


def timed(fn, reps=200):
    with torch.no_grad(), torch.cuda.stream(main):
        for _ in range(3):
            fn()
        main.synchronize(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            fn()
        for _ in range(5):
            g.replay()
        main.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        main.synchronize()
        return round(1e3 * (time.perf_counter() - t0) / reps, 4)


out = {}
for rnd in range(3):
    for name, fn in (("one_call_B2_ms", one_call), ("two_streams_B1_B1_ms", two_streams), ("one_row_B1_ms", one_row)):
        out.setdefault(name, []).append(timed(fn))
print(json.dumps(out))
