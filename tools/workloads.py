"""Timing of the hot-path workloads besides the headline one, shared by `bench.py` (its `other_workloads` and `end_to_end`
keys) and the `tools/bench_*.py` drivers.  Every function runs a bounded number of repetitions on the given device and
returns a dict with the measured time, the ALGORITHMIC work of the unit and the roofline fraction:
    HBM-bound rows : bytes / seconds / 8.0 TB/s      MFMA-bound rows : flops / seconds / 157.3 TFLOP/s (fp32 matrix peak)
Inputs are synthetic (pytracking_amd/synth.py generators), resident in HBM before the timed region.
"""
import math
import os
import time

import numpy as np
import torch

from pytracking_amd import bench_frame, synth

HBM_PEAK = 8.0e12
MFMA_F32_PEAK = 157.3e12


def _timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def sd_frame(dev, kind="prdimp", frames=100, graph_frames=25, num_iter=5):
    """One tracking frame (classify + arg-max + memory insert + steepest-descent solve) of PrDiMP-50 (BASELINE configs[2]
    per-GPU workload: n = 50 x 512 x 22 x 22) or DiMP-50, hipGraph replay like the headline number."""
    cfg = synth.PRDIMP50 if kind == "prdimp" else synth.DIMP50
    n = cfg["memory"]
    st = bench_frame.TrackState(cfg, n, seed=2234, device=dev, kind=kind)
    pool = torch.from_numpy(synth.clf_features(np.random.default_rng(5321), n, cfg["C"], cfg["H"], cfg["W"], cfg["K"])).to(dev)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        for f in range(2):
            st.step(pool[f], f, num_iter)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for f in range(graph_frames):                       # a frame chain that ends with its flush, like bench.run_frames
                st.step(pool[f % n], f % n, num_iter, defer=True)
            st.flush()
        g.replay()
        stream.synchronize()
        reps = max(1, frames // graph_frames)
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        stream.synchronize()
        dt = (time.perf_counter() - t0) / (reps * graph_frames)
    byts = st.bytes_per_solve(num_iter)
    name = "PrDiMP-50" if kind == "prdimp" else "DiMP-50"
    return {"workload": f"{name} frame: classify + arg-max + insert + {num_iter} SD iterations over n={n}x{cfg['C']}x{cfg['H']}x{cfg['W']}, K=4 "
                        f"(hipGraph, {graph_frames} frames per graph, frame chain + flush)",
            "ms": round(dt * 1e3, 5), "frames_per_s": round(1 / dt, 1), "bound": "hbm", "algorithmic_bytes": byts,
            "achieved_GBs": round(byts / dt / 1e9, 1), "frac": round(byts / dt / HBM_PEAK, 4)}


def lwl(dev, n=32, iters=3, reps=20):
    """LWL few-shot learner (BASELINE configs[4]): GNSteepestDescent on LWTLResidual, 16 filters 3x3, 512 channels, 30x52
    maps, n samples, `iters` iterations (the reference: 3 per frame, lwl_ytvos.py:26; BASELINE's wording: 4)."""
    from pytracking_amd.steepestdescent import GNSteepestDescent, LWTLResidual
    F, C, H, W, K = 16, 512, 30, 52, 3
    rng = np.random.default_rng(3)
    feat = torch.from_numpy(synth.clf_features(rng, n, C, H, W, K)).to(dev)[:, None]
    label = torch.from_numpy(rng.uniform(0, 1, (n, 1, F, H, W)).astype(np.float32)).to(dev)
    sw = torch.from_numpy(rng.uniform(0.2, 1, (n, 1, F, H, W)).astype(np.float32)).to(dev)
    w0 = torch.zeros(1, F, C, K, K, device=dev)
    opt = GNSteepestDescent(LWTLResidual(0.05).to(dev), num_iter=iters, compute_losses=False, residual_batch_dim=1)
    with torch.no_grad():
        dt = _timed(lambda: opt(w0, feat=feat, label=label, sample_weight=sw), reps)
    passes = 2 * iters + 1
    flops = passes * 2.0 * n * F * C * K * K * H * W
    return {"workload": f"LWL GNSteepestDescent n={n} F=16 C=512 30x52 K=3, {iters} iterations ({passes} feature passes)",
            "ms": round(dt * 1e3, 4), "solves_per_s": round(1 / dt, 1), "bound": "mfma", "algorithmic_flops": flops,
            "achieved_TFLOPs": round(flops / dt / 1e12, 2), "frac": round(flops / dt / MFMA_F32_PEAK, 4),
            "feature_GBs": round(passes * 4.0 * n * C * H * W / dt / 1e9, 1)}


def atom_cg(dev, reps=50):
    """ATOM online filter update (BASELINE configs[0] shape / north_star "ATOM's conjugate-gradient update"):
    ConjugateGradient on ConvProblem, n = 250 samples of 64x18x18, 4x4 filter, 5 PR-CG iterations
    (pytracking/parameter/atom/default.py); 2 feature passes per iteration + 2 for the initial residual."""
    from pytracking_amd.optimization import ConjugateGradient, ConvProblem, MLU
    c = synth.ATOM18
    n = c["memory"]
    x0, samples, y, sw = synth.atom_problem(1, n)
    T = lambda a: torch.from_numpy(a).to(dev)
    x = [T(x0.copy())[None].clone()]
    prob = ConvProblem([T(samples)], [T(y)[:, None]], [c["filter_reg"]], [T(sw)], MLU(c["act_min_val"]))
    opt = ConjugateGradient(prob, x, fletcher_reeves=False, direction_forget_factor=0)
    dt = _timed(lambda: opt.run(c["cg_iter"]), reps)
    passes = 2 * c["cg_iter"] + 2
    byts = passes * 4.0 * n * c["C"] * c["H"] * c["W"]
    byts_8d = 2 * c["cg_iter"] * 4.0 * n * c["C"] * c["H"] * c["W"]     # 2 feature reads per CG iteration, nothing for the linearisation
    return {"workload": f"ATOM ConvProblem CG update n={n} C=64 18x18 K=4, {c['cg_iter']} iterations ({passes} feature passes)",
            "ms": round(dt * 1e3, 5), "updates_per_s": round(1 / dt, 1), "bound": "hbm", "algorithmic_bytes": byts,
            "achieved_GBs": round(byts / dt / 1e9, 1), "frac": round(byts / dt / HBM_PEAK, 4),
            "survey_8d_count": {"algorithmic_bytes": byts_8d, "achieved_GBs": round(byts_8d / dt / 1e9, 1), "frac": round(byts_8d / dt / HBM_PEAK, 4),
                                "note": "SURVEY.md 8(d) counts 2 x 4 n C H W bytes per CG iteration x 5 (the two passes of the "
                                        "linearisation are not in it); `frac` above counts the 12 passes the update really makes"}}


def atom_first_frame(dev, reps=5):
    """ATOM first-frame joint optimisation (atom.py:156-176): 30 augmented samples x 256 x 18 x 18, 64 compressed channels,
    init_CG_iter 60 / init_GN_iter 6; once per sequence, launch-chain bound."""
    from pytracking_amd.optimization import FactorizedConvProblem, GaussNewtonCG, MLU
    c = synth.ATOM18
    T = lambda a: torch.from_numpy(a).to(dev)
    rng = np.random.default_rng(5)
    na, M, Kc, K = 30, 256, 64, 4
    raw = T(rng.standard_normal((na, M, 18, 18), dtype=np.float32) * np.float32(0.1))
    _, _, y2, sw2 = synth.atom_problem(5, na)
    P0 = T(rng.standard_normal((Kc, M, 1, 1), dtype=np.float32) * np.float32(1.0 / np.sqrt(M)))
    jp = FactorizedConvProblem([raw], [T(y2)[:, None]], [c["filter_reg"]], [1e-4], None, [T(sw2)], None, MLU(c["act_min_val"]))

    def joint():
        GaussNewtonCG(jp, [torch.zeros(1, Kc, K, K, device=dev), P0.clone()]).run(10, 6)

    dt = _timed(joint, reps, warm=1)
    return {"workload": "ATOM FactorizedConvProblem GaussNewtonCG n=30 M=256 Kc=64 18x18 K=4, 6 x 10 CG", "ms": round(dt * 1e3, 3)}


def tomp_flops(cfg):
    D, ff, L = cfg["D"], cfg["ff"], (cfg["n_train"] + 1) * cfg["H"] * cfg["W"]
    B, HW = 2, cfg["H"] * cfg["W"]
    enc = cfg["n_enc"] * B * (2.0 * L * D * 3 * D + 4.0 * L * L * D + 2.0 * L * D * D + 4.0 * L * D * ff)
    tok = B * (L - HW) * (2.0 * (D // 4) * D + 2.0 * D * D)
    tower = 4 * 2.0 * HW * D * 9 * D + 2.0 * HW * 4 * 9 * D
    return enc + tok + tower


def tomp(dev, reps=20, graph=True):
    """ToMP model prediction per frame (BASELINE configs[3]; tompnet50 / tompnet101 share these dimensions): 2 memory frames +
    the test frame of 256x18x18 head features -> 972 tokens x 2 batch rows, 6 + 6 layers, classifier + dense box regressor."""
    from pytracking_amd import transformer as TM
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
        for _ in range(2):
            frame()
        torch.cuda.synchronize()
        run = frame
        if graph:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                frame()
                with torch.cuda.graph(g, stream=s):
                    frame()
            torch.cuda.current_stream().wait_stream(s)
            run = g.replay
        dt = _timed(run, reps, warm=2)
    fl = tomp_flops(cfg)
    return {"workload": "ToMP predict_cls_bbreg_filters_parallel + classifier + bbreg, 2+1 frames 256x18x18, 6+6 layers"
                        + (" (hipGraph)" if graph else " (eager)"),
            "ms": round(dt * 1e3, 4), "frames_per_s": round(1 / dt, 1), "bound": "mfma", "algorithmic_flops": fl,
            "achieved_TFLOPs": round(fl / dt / 1e12, 2), "frac": round(fl / dt / MFMA_F32_PEAK, 4)}


# ---------------------------------------------------------------------------------------------------------------------
# end to end with the stock backbone ("end-to-end fps with the stock ResNet-50 on a U(0,255) patch")
# ---------------------------------------------------------------------------------------------------------------------
class _Bottleneck(torch.nn.Module):
    def __init__(self, cin, planes, stride):
        super().__init__()
        nn = torch.nn
        self.conv1, self.bn1 = nn.Conv2d(cin, planes, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)
        self.down = None
        if stride != 1 or cin != planes * 4:
            self.down = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        y = torch.relu(self.bn1(self.conv1(x)))
        y = torch.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return torch.relu(y + (x if self.down is None else self.down(x)))


def resnet50_to_layer3():
    """ResNet-50 conv1 ... layer3 in plain torch.nn (the standard architecture the reference takes from torchvision,
    ltr/models/backbone/resnet.py; torchvision is not in this image), random init: 3x288x288 -> 1024x18x18."""
    nn = torch.nn
    layers = [nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1)]
    cin = 64
    for planes, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2)):
        for b in range(blocks):
            layers.append(_Bottleneck(cin, planes, stride if b == 0 else 1))
            cin = planes * 4
    return nn.Sequential(*layers)


def end_to_end(dev, frames=60, num_iter=5):
    """Per frame: U(0,255) 288x288 patch (already on the device) -> ImageNet normalisation -> stock-PyTorch ResNet-50
    (conv1..layer3, fp32, eager, MIOpen) -> pt_track_frame_head_f32 (classification-feature head writing the memory slot +
    classify + arg-max + 5 SD iterations over n = 50).  Reports the backbone's own time next to the whole frame."""
    cfg = synth.DIMP50
    n = cfg["memory"]
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")        # ~25 convolution shapes: no exhaustive MIOpen search per shape
    torch.manual_seed(7)
    net = resnet50_to_layer3().to(dev).eval()
    st = bench_frame.TrackState(cfg, n, seed=777, device=dev)
    rng = np.random.default_rng(779)
    w = torch.from_numpy(rng.standard_normal((cfg["C"], 1024, 3, 3), dtype=np.float32) * np.float32(0.02)).to(dev)
    st.attach_head(w, math.sqrt(1.0 / (cfg["C"] * cfg["K"] ** 2)))
    patches = torch.from_numpy(rng.uniform(0, 255, (4, 3, 288, 288)).astype(np.float32)).to(dev)
    mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1) * 255
    std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1) * 255

    def backbone(k):
        return net((patches[k % 4:k % 4 + 1] - mean) / std)[0].contiguous()

    k = [0]

    def frame():
        st.step_from_backbone(backbone(k[0]), k[0] % n, num_iter)
        k[0] += 1

    out = {"workload": "U(0,255) 288x288 patch -> stock PyTorch ResNet-50 conv1..layer3 (fp32, random init) -> head + classify + "
                       "arg-max + insert + 5 SD iterations (pt_track_frame_head_f32), n=50"}
    with torch.no_grad():
        assert backbone(0).shape == (1024, cfg["H"], cfg["W"])
        t_bb = _timed(lambda: backbone(0), frames, warm=5)
        t_all = _timed(frame, frames, warm=5)
        out["eager"] = {"backbone_ms": round(t_bb * 1e3, 4), "frame_ms": round(t_all * 1e3, 4), "frames_per_s": round(1 / t_all, 1),
                        "note": "host-launch bound: ~150 eager stock-PyTorch kernel launches per backbone pass; the hot path's "
                                "device time hides in the launch gaps"}
        # the same work as hipGraphs (stock torch.cuda.graph capture of the backbone; one graph per patch / slot pair):
        # device time of the backbone alone and of the whole frame
        try:
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb, stream=s):
                    keep = backbone(0)
                G = 8
                gf = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gf, stream=s):
                    for f in range(G):
                        st.step_from_backbone(backbone(f), f % n, num_iter)
                t_gb = _timed(gb.replay, frames, warm=3)
                t_gf = _timed(gf.replay, max(2, frames // G), warm=2) / G
            torch.cuda.current_stream().wait_stream(s)
            del keep
            out["graph"] = {"backbone_ms": round(t_gb * 1e3, 4), "frame_ms": round(t_gf * 1e3, 4), "frames_per_s": round(1 / t_gf, 1),
                            "hot_path_share": round(max(0.0, 1.0 - t_gb / t_gf), 4)}
        except Exception as exc:                                 # noqa: BLE001
            out["graph"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    best = out["graph"] if "frames_per_s" in out.get("graph", {}) else out["eager"]
    out["frames_per_s"], out["backbone_ms"], out["frame_ms"] = best["frames_per_s"], best["backbone_ms"], best["frame_ms"]
    return out


def iou_refine(dev):
    """IoU-guided box refinement as the trackers call it per frame: tools/bench_iou.py without the stock leg."""
    from tools import bench_iou
    return bench_iou.measure(dev, with_stock=False, reps=100)


def frame_after_backbone(dev):
    """Everything of a DiMP-50 frame behind the backbone (tools/bench_dimp_frame_extended.py)."""
    from tools import bench_dimp_frame_extended
    return bench_dimp_frame_extended.measure(dev, frames=200)


def all_other(dev):
    """The BASELINE configs beyond the headline one and the rows either side of the solver, in a few hundred ms of device time."""
    out = {}
    for key, fn in (("prdimp50_frame", lambda: sd_frame(dev, "prdimp")), ("tomp_predict", lambda: tomp(dev)),
                    ("lwl_n32_it3", lambda: lwl(dev, 32, 3)), ("lwl_n32_it4", lambda: lwl(dev, 32, 4)),
                    ("lwl_n8_it3", lambda: lwl(dev, 8, 3)), ("atom_cg_n250", lambda: atom_cg(dev)),
                    ("iou_refine", lambda: iou_refine(dev)), ("dimp50_frame_after_backbone", lambda: frame_after_backbone(dev))):
        try:
            out[key] = fn()
        except Exception as exc:                                 # noqa: BLE001 -- a failing side workload must not hide the headline
            out[key] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    return out
