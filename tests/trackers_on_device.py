"""TEST INFRASTRUCTURE -- the UNMODIFIED reference trackers on the MI355X through `pytracking_amd.install()`.

    python -B tests/trackers_on_device.py [--stock] [--out FILE]       (GPU box; needs oracle/_ref, see oracle/make_ref_bundle.py)

`pytracking.tracker.dimp.DiMP` (DiMP-50 and PrDiMP-50 parameter sets), `tomp.ToMP` and `atom.ATOM` exactly as the reference ships them (byte-for-byte bundle,
sha256 manifest), constructed AFTER `install()`, `params.use_gpu = True`, network and features on `cuda:0` -- i.e. what a user of
the reference runs (pytracking/parameter/dimp/dimp50.py:10, evaluation/tracker.py `create_tracker`).  The stock-PyTorch parts in
front of the hot path (backbone, IoU-feature convolutions) are the seeded stubs of oracle/tracker_harness.py, the same ones the
committed CPU logs tests/golden/tracker_{dimp50,prdimp50,tomp50,atom18}.npz were recorded with, so every call across the hot-path boundary
can be compared event by event with the CPU run of the same tracker WITHOUT install():

  * same event sequence (a flipped flag / schedule decision would change it),
  * every float payload (score maps, filters, refined boxes, IoU, translation vectors, output boxes) within 1e-4,
  * every integer / string payload (scale index, flag, memory slot) equal,
  * `install.stats`: the gfx950 branch of every rebound symbol the tracker reaches was taken, the reference branch never.

`--stock`: additionally run the same trackers on the GPU WITHOUT install() (the reference's own stock PyTorch-ROCm ops) and
report wall time per track() call next to the installed run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import tracker_replay as TR  # noqa: E402

GOLDEN = os.path.join(HERE, "golden")

# payload keys that are inputs the harness regenerates or bookkeeping of the recording itself
_SKIP_KEYS = {"kind"}
# |sum| of a whole feature map (float64 accumulate of ~1e5 values): compared relatively
_REL_KEYS = {"checksum": 2e-5}
# Pixel coordinates (magnitudes of several hundred px: one float32 ulp is 3e-5 at 256).  The run is CLOSED LOOP: the inputs of a
# boundary call carry the deviations of all earlier calls (PrDiMP: `refine.init_boxes` already differ by 6e-5 from the CPU run, and
# the reference's own stock-PyTorch GPU run deviates from its CPU run by 9.2e-5 on `refine.boxes`), so these payloads get 1e-4 plus
# 1e-6 relative (~3 ulp) on top.  On identical inputs the 1e-4 bound is asserted without it by the replay tests
# (tests/test_tracker_trajectory.py) and the golden tests of the refinement.
# LWL only: outputs of the STOCK-PyTorch decoder behind the path (480 x 832 raw scores, stored as float16 in the log) and the box /
# mask area the tracker derives from them -- not hot-path payloads; the hot-path payloads of the same events (filters, mask encodings)
# keep the 1e-4 bound
STOCK_OVERRIDES = {"lwl": {"scores": 2e-3, "target_bbox": 5e-2, "mask_area": 64.0}}
_COORD_RTOL = {"boxes": 1e-6, "init_boxes": 1e-6, "target_bbox": 1e-6, "bb": 1e-6, "pos": 1e-6, "target_sz": 1e-6, "bbox": 1e-6}


def compare_logs(got, want, atol=1e-4, overrides=None):
    """Event-by-event comparison; returns {"kind.key": max abs deviation}.  Raises AssertionError with the first mismatch."""
    kg, kw = [e["kind"] for e in got], [e["kind"] for e in want]
    assert kg == kw, "event sequence differs from the CPU run:\n  got  %s\n  want %s" % (kg, kw)
    dev = {}
    for i, (a, b) in enumerate(zip(got, want)):
        for key, wv in b.items():
            if key in _SKIP_KEYS:
                continue
            assert key in a, (i, b["kind"], key)
            wv = np.asarray(wv)
            gv = np.asarray(a[key])
            tag = f"{b['kind']}.{key}"
            if wv.dtype.kind in "US":
                assert str(gv) == str(wv), (i, tag, str(gv), str(wv))
            elif wv.dtype.kind in "iub":
                assert np.array_equal(gv.astype(np.int64), wv.astype(np.int64)), (i, tag, gv, wv)
            else:
                gv, wv = gv.astype(np.float64), wv.astype(np.float64)
                assert gv.shape == wv.shape, (i, tag, gv.shape, wv.shape)
                if key in _REL_KEYS:
                    err = float(np.abs(gv - wv).max() / max(np.abs(wv).max(), 1e-30))
                    assert err <= _REL_KEYS[key], (i, tag, err)
                else:
                    fin = np.isfinite(wv)
                    assert np.array_equal(fin, np.isfinite(gv)), (i, tag, "non-finite pattern")
                    err = float(np.abs(gv[fin] - wv[fin]).max()) if fin.any() else 0.0
                    tol = atol + _COORD_RTOL.get(key, 0.0) * (float(np.abs(wv[fin]).max()) if fin.any() else 0.0)
                    tol = max(tol, (overrides or {}).get(key, 0.0))
                    assert err <= tol, (i, tag, err, tol)
                dev[tag] = max(dev.get(tag, 0.0), err)
    return dev


def _golden(name):
    return TR.events_from_npz(dict(np.load(os.path.join(GOLDEN, name + ".npz"))))


def _timed_track(tracker_cls):
    """Wrap `track` of a tracker class so that the wall time of every call (device synchronised) is collected."""
    import torch
    times = []
    orig = tracker_cls.track

    def track(self, image, info=None):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = orig(self, image, info)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        return out
    tracker_cls.track = track
    return times, lambda: setattr(tracker_cls, "track", orig)


# the branch every tracker must reach on the device, and the ones it must never fall out of
EXPECT_FAST = {
    "dimp": ("residual_bottleneck", "PrRoIPool2D", "DiMPSteepestDescentGN", "apply_filter", "DiMP.localize_advanced",
             "DiMP.optimize_boxes_default"),
    "prdimp": ("residual_bottleneck", "PrRoIPool2D", "PrDiMPSteepestDescentNewton", "apply_filter", "DiMP.localize_advanced",
               "DiMP.optimize_boxes_relative"),
    "tomp": ("residual_bottleneck", "ToMP.localize_advanced"),
    "lwl": ("GNSteepestDescent", "apply_filter"),
    "atom": ("GaussNewtonCG", "ConjugateGradient", "operation.conv2d[same]", "ATOM.optimize_boxes", "PrRoIPool2D"),
}


def run(which, installed=True, device="cuda"):
    """One tracker run on `device`; returns (events, stats dict, seconds per track() call, extras)."""
    from oracle import tracker_harness as TH
    TH.ref_harness.install()
    from pytracking_amd import install as amd
    if installed:
        amd.install()
        amd.stats.clear()
    try:
        if which in ("dimp", "prdimp"):
            from pytracking.tracker.dimp.dimp import DiMP as cls
            times, undo = _timed_track(cls)
            try:
                outs, rec, (tracker, net) = TH.run_dimp(device=device, **(TH.DIMP_RUN if which == "dimp" else TH.PRDIMP_RUN))
            finally:
                undo()
            extra = {"filter_optimizer": type(net.classifier.filter_optimizer).__mro__[1].__module__,
                     "head": type(net.classifier.feature_extractor).__mro__[1].__module__}
        elif which == "tomp":
            from pytracking.tracker.tomp.tomp import ToMP as cls
            times, undo = _timed_track(cls)
            try:
                outs, rec, (tracker, net) = TH.run_tomp(device=device, **TH.TOMP_RUN)
            finally:
                undo()
            extra = {"filter_predictor": type(net.head.filter_predictor).__module__,
                     "transformer": type(net.head.filter_predictor.transformer).__module__,
                     "classifier": type(net.head.classifier).__module__, "bb_regressor": type(net.head.bb_regressor).__module__}
        elif which == "lwl":
            from pytracking.tracker.lwl.lwl import LWL as cls
            times, undo = _timed_track(cls)
            try:
                outs, rec, (tracker, net) = TH.run_lwl(device=device, **TH.LWL_RUN)
            finally:
                undo()
            extra = {"filter_optimizer": type(net.target_model.filter_optimizer).__mro__[1].__module__,
                     "residual_module": type(net.target_model.filter_optimizer.residual_module).__mro__[1].__module__}
        elif which == "atom":
            from pytracking.tracker.atom.atom import ATOM as cls
            times, undo = _timed_track(cls)
            try:
                outs, rec, (tracker, net) = TH.run_atom(device=device, **TH.ATOM_RUN)
            finally:
                undo()
            extra = {"optimizers": sorted({f"{k}:{m}" for k, m in rec.notes})}
        else:
            raise ValueError(which)
        stats = dict(amd.stats) if installed else {}
    finally:
        if installed:
            amd.uninstall()
    # through the same serialisation as the committed logs (dtypes, scalars -> 0-d arrays)
    events = TR.events_from_npz({k: np.asarray(v) for k, v in rec.to_npz_dict().items()})
    return events, stats, times, extra


GOLDEN_NAME = {"dimp": "tracker_dimp50", "prdimp": "tracker_prdimp50", "tomp": "tracker_tomp50", "atom": "tracker_atom18", "lwl": "tracker_lwl"}


def check(which, atol=1e-4):
    events, stats, times, extra = run(which, installed=True)
    dev = compare_logs(events, _golden(GOLDEN_NAME[which]), atol=atol, overrides=STOCK_OVERRIDES.get(which))
    for name in EXPECT_FAST[which]:
        assert stats.get(name + ".fast", 0) > 0, f"{which}: gfx950 branch of `{name}` never taken: {stats}"
    fell = {k: v for k, v in stats.items() if k.endswith(".reference") and not k.startswith(("sample_patch", "max2d"))}
    assert not fell, f"{which}: calls handed back to the reference's own code on the device: {fell}"
    return dev, stats, times, extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stock", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default="dimp,prdimp,tomp,atom,lwl")
    args = ap.parse_args()
    import torch
    lines = [f"unmodified reference trackers on {torch.cuda.get_device_name(0)} through pytracking_amd.install(); "
             f"compared event by event with the CPU run of the same trackers without install() (tests/golden/tracker_*.npz)"]
    rc = 0
    for which in args.only.split(","):
        try:
            dev, stats, times, extra = check(which)
            lines.append(f"\n== {which}: PASS (every payload <= 1e-4 [pixel coordinates: + 1e-6 relative, closed loop], flags / indices / slots equal, same event sequence)")
            lines.append("   max |deviation| per boundary payload: " + json.dumps({k: float(f"{v:.3g}") for k, v in sorted(dev.items())}))
            lines.append("   install.stats (branch taken per rebound symbol): " + json.dumps(dict(sorted(stats.items()))))
            lines.append("   classes built by the reference's constructors: " + json.dumps(extra))
            lines.append("   track() wall ms per call (stub backbone, device synchronised): " + json.dumps([round(1e3 * t, 2) for t in times]))
        except AssertionError as exc:
            rc = 1
            lines.append(f"\n== {which}: FAIL {str(exc)[:3000]}")
        except Exception as exc:                                    # noqa: BLE001
            import traceback
            rc = 1
            lines.append(f"\n== {which}: ERROR {type(exc).__name__}: {exc}\n{traceback.format_exc()[-4000:]}")
        if args.stock:
            try:
                ev, _, t_stock, _ = run(which, installed=False)
                lines.append(f"   same tracker WITHOUT install() on the GPU (reference's stock PyTorch-ROCm ops, PrRoIPool = autograd "
                             f"restatement): track() wall ms per call: " + json.dumps([round(1e3 * t, 2) for t in t_stock]))
                try:
                    d2 = compare_logs(ev, _golden(GOLDEN_NAME[which]), atol=1e-3, overrides=STOCK_OVERRIDES.get(which))
                    lines.append("   its deviation from the CPU run: " + json.dumps({k: float(f"{v:.3g}") for k, v in sorted(d2.items())}))
                except AssertionError as exc:
                    lines.append(f"   its log does not match the CPU run within 1e-3: {str(exc)[:500]}")
            except Exception as exc:                                # noqa: BLE001
                lines.append(f"   stock GPU run failed: {type(exc).__name__}: {str(exc)[:500]}")
    text = "\n".join(lines)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            fh.write(text + "\n")
    return rc


if __name__ == "__main__":
    sys.exit(main())
