"""TEST INFRASTRUCTURE (not collected by pytest) -- where does the wall time of an installed `track()` go?

    python -B tests/profile_installed_track.py [--frames 40] [--out profiles/r06_installed_track_breakdown.txt]      (GPU box, needs oracle/_ref)

The UNMODIFIED reference `DiMP` tracker (DiMP-50 parameter set, oracle/tracker_harness.py) on `cuda:0` through `pytracking_amd.install()`,
with everything the harness used to do per call moved OFF the clock: the stub backbone / IoU-feature maps of every call and the synthetic
frames are generated once, kept on the device (frames: host uint8 arrays, as `track()` receives them), and replayed from a cache in the
timed run.  Reported per `track()` call:
  * wall time (device synchronised), update frames and classification-only frames separately;
  * host time by owner from cProfile (tottime, i.e. exclusive): this repo's wrappers (pytracking_amd/), the reference's Python
    (pytracking/, ltr/), the harness (oracle/, tests/), torch and other library code -- and the top functions of each;
  * device time: sum of kernel durations per call when run under `rocprofv3 --kernel-trace` (tools/r06_run4.sh passes the figure in).
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def build(n_frames, cache, frames, seed=4100, device="cuda"):
    """The tracker of oracle.tracker_harness.run_dimp(record=False, device='cuda') with cached stub outputs."""
    import torch
    from oracle import tracker_harness as TH
    TH.ref_harness.install()
    from pytracking.tracker.dimp.dimp import DiMP
    dims = TH.DIMP50_TEST
    net = TH.build_dimp50(seed, dims).to(device)
    stub = TH.StubBackbone(seed, dims, device)
    orig_b, orig_i = stub.backbone, stub.iou_feat

    def backbone(n):
        key = ("b", stub.k_backbone, n)
        if key not in cache:
            cache[key] = orig_b(n)
        else:
            stub.k_backbone += 1
        return cache[key]

    def iou_feat(n):
        key = ("i", stub.k_iou, n)
        if key not in cache:
            cache[key] = orig_i(n)
        else:
            stub.k_iou += 1
        return cache[key]
    stub.backbone, stub.iou_feat = backbone, iou_feat
    ns = TH.NetStub(net, stub)
    params = TH.use_device(TH.dimp50_params(ns), device)
    tracker = DiMP(params)
    tracker.visdom = None
    net.bb_regressor.get_iou_feat = lambda feats: stub.iou_feat(feats[0].shape[0])
    torch.manual_seed(seed)
    if not frames:
        rng = np.random.default_rng(seed + 77)
        frames.extend(TH.synthetic_frame(rng) for _ in range(n_frames + 1))
    return tracker


def owner(path):
    p = path.replace("\\", "/")
    if "/pytracking_amd/" in p:
        return "this repo's wrappers (pytracking_amd/)"
    if "/oracle/_ref/" in p or "/reference/" in p:
        return "reference Python (pytracking/, ltr/)"
    if "/oracle/" in p or "/tests/" in p:
        return "harness (oracle/, tests/)"
    if "/torch/" in p:
        return "torch (Python side)"
    if p.startswith("~") or p.startswith("<") or p == "":
        return "built-ins / C calls (torch ops, ctypes calls into libpt_hot.so, numpy)"
    return "other library code"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--out", default=None)
    ap.add_argument("--device-us-per-call", type=float, default=None, help="sum of kernel durations per track() from a rocprofv3 run")
    ap.add_argument("--plain", action="store_true", help="timed loop only (for running under rocprofv3)")
    ap.add_argument("--device", default="cuda", help="cpu: control-flow dry run of this script (the reference's own branches run)")
    a = ap.parse_args()
    import torch
    from oracle import tracker_harness as TH
    TH.ref_harness.install()                                       # the reference tree (bundle) on sys.path, stub modules registered
    from pytracking_amd import install as amd
    sync = torch.cuda.synchronize if a.device == "cuda" else (lambda: None)
    amd.install()
    cache, frames = {}, []
    box = {"init_bbox": [200.0, 140.0, 70.0, 90.0]}
    # pass 1 fills the caches (and warms every code path); pass 2 is timed; pass 3 runs under cProfile
    walls, kinds = [], []
    prof = cProfile.Profile()
    for rnd in range(2 if a.plain else 3):
        tracker = build(a.frames, cache, frames, device=a.device)
        tracker.initialize(frames[0], dict(box))
        sync()
        for f in range(a.frames):
            n_upd0 = amd.stats.get("DiMPSteepestDescentGN.fast", 0)
            sync()
            if rnd == 2:
                prof.enable()
            t0 = time.perf_counter()
            tracker.track(frames[1 + f])
            sync()
            dt = time.perf_counter() - t0
            if rnd == 2:
                prof.disable()
            if rnd == 1:
                walls.append(dt)
                kinds.append("update" if amd.stats.get("DiMPSteepestDescentGN.fast", 0) > n_upd0 else "classify-only")
    stats = dict(amd.stats)
    amd.uninstall()
    if a.plain:
        print("plain run done:", len(walls), "timed track() calls")
        return 0
    lines = []
    w = np.array(walls) * 1e3
    lines.append(f"unmodified reference DiMP (DiMP-50 parameters, train_skipping 2 in the harness) on {torch.cuda.get_device_name(0) if a.device == 'cuda' else 'CPU (dry run)'} through install(); "
                 f"stub backbone / IoU features and frames pre-generated (device-resident, replayed from a cache): {len(w)} track() calls")
    lines.append(f"wall ms per track() (device synchronised): median {np.median(w):.3f}, mean {w.mean():.3f}, min {w.min():.3f}, max {w.max():.3f}")
    for kind in ("update", "classify-only"):
        sel = w[[k == kind for k in kinds]]
        if len(sel):
            lines.append(f"   {kind:14s}: {len(sel):3d} calls, median {np.median(sel):.3f} ms, min {sel.min():.3f}")
    if a.device_us_per_call is not None:
        lines.append(f"device time per track() (sum of kernel durations, rocprofv3 --kernel-trace of the same loop): {a.device_us_per_call:.1f} us")
    # ---- cProfile split (exclusive time per owner)
    st = pstats.Stats(prof)
    tot = {}
    per_fn = {}
    for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
        o = owner(fn)
        tot[o] = tot.get(o, 0.0) + tt
        per_fn.setdefault(o, []).append((tt, nc, f"{os.path.basename(fn)}:{line} {name}" if not fn.startswith("~") else name))
    calls = a.frames
    total = sum(tot.values())
    lines.append(f"\ncProfile of the same {calls} calls (profiler on: Python-heavy parts are inflated ~1.5-2x; exclusive time, ms per track()):")
    lines.append(f"   total under the profiler: {1e3 * total / calls:.3f} ms per call")
    for o, t in sorted(tot.items(), key=lambda kv: -kv[1]):
        lines.append(f"   {1e3 * t / calls:8.3f} ms  {100 * t / total:5.1f} %   {o}")
    for o in sorted(per_fn, key=lambda k: -tot[k]):
        lines.append(f"\n   top functions, {o}:")
        for tt, nc, name in sorted(per_fn[o], reverse=True)[:12]:
            lines.append(f"      {1e6 * tt / calls:9.1f} us/call  {nc / calls:7.1f} calls/track  {name}")
    # cumulative time of this repo's entry points (inclusive: what a rebound symbol costs end to end on the host)
    lines.append("\n   inclusive host time of this repo's functions (cumulative, us per track()):")
    rows = []
    for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
        if "/pytracking_amd/" in fn.replace("\\", "/"):
            rows.append((ct, nc, f"{os.path.basename(fn)}:{line} {name}"))
    for ct, nc, name in sorted(rows, reverse=True)[:25]:
        lines.append(f"      {1e6 * ct / calls:9.1f} us/call  {nc / calls:7.1f} calls/track  {name}")
    lines.append("\ninstall.stats over all passes: " + str(dict(sorted(stats.items()))))
    text = "\n".join(lines)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(text + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
