"""Multi-sequence steepest-descent solve (optimizer.py:101-104 `num_sequences`): S sequences in ONE pt_sd_solve_batch_f32 call (spread
over the calling stream and up to three side streams) against S single-sequence calls back to back.   python tools/bench_multi_seq_solve.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import optimizer as OM, synth  # noqa: E402


def measure(dev, reps=30):
    cfg = synth.DIMP50
    mod = OM.DiMPSteepestDescentGN(num_iter=5, feat_stride=cfg["feat_stride"], init_step_length=cfg["init_step_length"],
                                   init_filter_reg=cfg["init_filter_reg"], init_gauss_sigma=cfg["init_gauss_sigma"],
                                   num_dist_bins=cfg["num_dist_bins"], bin_displacement=cfg["bin_displacement"],
                                   mask_init_factor=cfg["mask_init_factor"], score_act=cfg["score_act"], mask_act=cfg["mask_act"],
                                   min_filter_reg=cfg["min_filter_reg"], alpha_eps=cfg["alpha_eps"]).to(dev).eval()
    out = {}
    for S in (2, 4, 8):
        probs = [synth.dimp_problem(700 + s, cfg["memory"], cfg) for s in range(S)]
        T = lambda a: torch.from_numpy(a).to(dev)
        w0 = torch.stack([T(p[0]) for p in probs])
        feat = torch.stack([T(p[1]) for p in probs], dim=1).contiguous()
        bb = torch.stack([T(p[2]) for p in probs], dim=1).contiguous()
        sw = torch.stack([T(p[3]) for p in probs], dim=1).contiguous()
        singles = [(w0[s:s + 1], feat[:, s].contiguous(), bb[:, s].contiguous(), sw[:, s].contiguous()) for s in range(S)]

        def batch():
            mod(w0, feat, bb, sample_weight=sw, num_iter=5, compute_losses=False)

        def loop():
            for w, f, b, s_ in singles:
                mod(w, f, b, sample_weight=s_, num_iter=5, compute_losses=False)
        res = {}
        with torch.no_grad():
            # graph replays: the batch call captured (side streams -> parallel branches) against the single calls captured (one chain)
            side = torch.cuda.Stream(device=dev)
            graphs = {}
            for name, fn in (("graph_batch_call_us", batch), ("graph_single_calls_us", loop)):
                with torch.cuda.stream(side):
                    fn()
                    side.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        fn()
                    g.replay()
                    side.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        g.replay()
                    side.synchronize()
                    res[name] = round((time.perf_counter() - t0) / reps * 1e6, 1)
                    graphs[name] = g
            res["graph_speedup"] = round(res["graph_single_calls_us"] / res["graph_batch_call_us"], 3)
            for name, fn in (("one_batch_call_us", batch), ("single_calls_us", loop)):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                res[name] = round((time.perf_counter() - t0) / reps * 1e6, 1)
        res["speedup"] = round(res["single_calls_us"] / res["one_batch_call_us"], 3)
        res["us_per_sequence"] = round(res["one_batch_call_us"] / S, 1)
        out[f"S{S}"] = res
    out["workload"] = "DiMPSteepestDescentGN, 5 iterations, n = 50 x 512 x 18 x 18 per sequence, eager launches through the module mirror"
    return out


if __name__ == "__main__":
    print(json.dumps(measure(torch.device("cuda", 0))))
