"""Trajectory-level integration parity: an UNMODIFIED reference tracker -- `pytracking.tracker.dimp.DiMP`
`initialize()` + 10 x `track()` on a random-init dimpnet50 with the stock-PyTorch parts stubbed by seeded features --
was run on CPU by oracle/tracker_harness.py and every call across the hot-path boundary recorded
(tests/golden/tracker_dimp50.npz).

  * CPU, reference mounted:  the log replays through the reference's own modules to float32 summation order (validates
    log + player; bit-exact at the recording's thread count);
    the same tracker with `pytracking_amd.install()` active produces the same trajectory (everything off the hot path
    falls back to the reference: "trackers run unchanged after install()").
  * GPU:  the log replays through the gfx950 modules in the tracker's call order, solver state (filter, memory, head
    features) carried closed-loop over all frames, every score map / filter / refined box within 1e-4.
"""
import numpy as np
import pytest

from conftest import load_golden
import tracker_replay as TR


def _events():
    return TR.events_from_npz(load_golden("tracker_dimp50"))


def test_log_covers_every_boundary_call():
    evs = _events()
    kinds = [e["kind"] for e in evs]
    for k in ("head", "get_filter", "classify", "localize", "refine", "memory", "optimize", "state"):
        assert k in kinds, k
    flags = {str(e["flag"]) for e in evs if e["kind"] == "localize"}
    assert {"normal", "hard_negative"} <= flags
    iters = {int(e["num_iter"]) for e in evs if e["kind"] == "optimize"}
    assert iters == {1, 2}                                       # hard-negative update and the regular cadence
    assert int(evs[0]["n_frames"]) == 10 and sum(k == "classify" for k in kinds) == 10


def _need_reference():
    from oracle import ref_harness
    if not ref_harness.available():
        pytest.skip("reference tree not mounted")


def test_reference_modules_replay_the_log():
    """Bit-exact when the CPU thread count equals the recording's (oneDNN partitions its sums by thread); 2e-5 covers
    any other partitioning (the 256-core host of the GPU box, with the reference bundle: 6.9e-6 on the ATOM log).  Flags and scale
    indices must match exactly either way."""
    _need_reference()
    from oracle import tracker_harness as TH
    net = TH.build_dimp50(TH.DIMP_RUN["seed"], TH.DIMP_RUN["dims"])
    tracker = type("T", (), {})()
    tracker.params = TH.dimp50_params(None)
    for k, v in TH.DIMP_RUN["thresholds"].items():
        setattr(tracker.params, k, v)
    dev = TR.replay(_events(), TH.RefOps(net, tracker), atol=2e-5)
    assert set(dev) == {"get_filter", "classify", "localize", "refine_iou", "refine_boxes", "optimize"}


def test_reference_tracker_runs_unchanged_after_install():
    """install() must be transparent for an unmodified tracker whose tensors are off the hot path (CPU): same boxes,
    same flags, same score maps as the recorded run, through the rebound classes / functions / methods."""
    _need_reference()
    from oracle import tracker_harness as TH
    from pytracking_amd import install as amd
    TH.ref_harness.install()
    amd.install()
    try:
        outs, rec, (tracker, net) = TH.run_dimp(**TH.DIMP_RUN)
        from pytracking_amd import optimizer as OM
        assert isinstance(net.classifier.filter_optimizer, OM.DiMPSteepestDescentGN)     # the rebound class was built
    finally:
        amd.uninstall()
    want = _events()
    got = rec.events
    assert [e["kind"] for e in got] == [e["kind"] for e in want]
    for a, b in zip(got, want):
        for k in ("scores", "filter", "boxes", "iou", "target_bbox", "tv"):
            if k in b:
                np.testing.assert_allclose(np.asarray(a[k], dtype=np.float64), np.asarray(b[k], dtype=np.float64), atol=1e-6,
                                           err_msg=f"{a['kind']}.{k}")
        if "flag" in b:
            assert str(a["flag"]) == str(b["flag"])


@pytest.mark.gpu
def test_gfx950_modules_replay_the_tracker_log():
    evs = _events()
    dev = TR.replay(evs, TR.MirrorOps(evs, "cuda"), atol=1e-4)
    assert set(dev) == {"get_filter", "classify", "localize", "refine_iou", "refine_boxes", "optimize"}
    print("max deviation per boundary call over the 10-frame trajectory:", {k: f"{v:.2e}" for k, v in dev.items()})


# ------------------------------------------------------------------------------------------------------
# ToMP-50: initialize() + 6 x track() of the unmodified reference tracker (tests/golden/tracker_tomp50.npz)
# ------------------------------------------------------------------------------------------------------
def _tomp_events():
    return TR.events_from_npz(load_golden("tracker_tomp50"))


def test_tomp_log_covers_both_memory_sizes():
    evs = _tomp_events()
    ns = {int(e["n"]) for e in evs if e["kind"] == "tomp_classify"}
    assert ns == {1, 2} and sum(e["kind"] == "memory" for e in evs) >= 2
    assert sum(e["kind"] == "tomp_classify" for e in evs) == int(evs[0]["n_frames"]) == 6


def test_reference_tomp_modules_replay_the_log():
    _need_reference()
    from oracle import tracker_harness as TH
    net = TH.build_tomp50(TH.TOMP_RUN["seed"], TH.TOMP_RUN["dims"])
    tracker = type("T", (), {})()
    tracker.params = TH.tomp50_params(None)
    for k, v in TH.TOMP_RUN["thresholds"].items():
        setattr(tracker.params, k, v)
    dev = TR.replay_tomp(_tomp_events(), TH.TompRefOps(net, tracker), atol=2e-5)
    assert set(dev) == {"scores", "log_bbox", "localize", "score_loc"}


@pytest.mark.gpu
def test_gfx950_modules_replay_the_tomp_tracker_log():
    evs = _tomp_events()
    dev = TR.replay_tomp(evs, TR.TompMirrorOps(evs, "cuda"), atol=1e-4)
    print("ToMP: max deviation per boundary call over the 6-frame trajectory:", {k: f"{v:.2e}" for k, v in dev.items()})


# ------------------------------------------------------------------------------------------------------
# ATOM: initialize() + 8 x track() of the unmodified reference tracker (tests/golden/tracker_atom18.npz).  `ATOM.track()` goes
# through pytracking/libs/fourier.py:24,31 (`torch.rfft` / `torch.irfft`, removed in torch 1.8): oracle/ref_harness.py supplies
# them on top of torch.fft for the recording run (harness only).
# ------------------------------------------------------------------------------------------------------
def _atom_events():
    return TR.events_from_npz(load_golden("tracker_atom18"))


def test_atom_log_covers_every_boundary_call():
    evs = _atom_events()
    kinds = [e["kind"] for e in evs]
    for k in ("atom_gn", "atom_init_done", "atom_classify", "atom_localize", "atom_refine", "atom_memory", "atom_cg", "state"):
        assert k in kinds, k
    gn = next(e for e in evs if e["kind"] == "atom_gn")
    assert int(gn["num_cg_iter"]) == 10 and int(gn["num_gn_iter"]) == 6 and int(gn["n_aug"]) == 11      # atom/default.py:27-28
    assert sum(k == "atom_classify" for k in kinds) == int(evs[0]["n_frames"]) == 8
    assert {str(e["flag"]) for e in evs if e["kind"] == "atom_localize"} >= {"hard_negative", "uncertain"}
    assert sum(k == "atom_cg" for k in kinds) >= 5 and sum(k == "atom_memory" for k in kinds) >= 4


def test_reference_atom_modules_replay_the_log():
    _need_reference()
    from oracle import tracker_harness as TH
    iounet = TH.build_atom_iounet(TH.ATOM_RUN["seed"], TH.ATOM_RUN["dims"])
    params = TH.atom_params(None, TH.ATOM_RUN["thresholds"])
    dev = TR.replay_atom(_atom_events(), TH.AtomRefOps(iounet, params), atol=2e-5)
    assert set(dev) == {"gn_filter", "gn_projection", "classify", "refine_iou", "refine_boxes", "cg_filter"}


@pytest.mark.gpu
def test_gfx950_modules_replay_the_atom_tracker_log():
    """Closed loop over the 8-frame ATOM trajectory on the GPU: joint Gauss-Newton at the deployed 6 x 10 schedule on the 11
    first-frame samples, every classification score map, every backtracking box refinement and every CG update of the filter over the
    250-slot memory (the fused C = 64 path of csrc/atom_cg.hip) within 1e-4 of the unmodified tracker's CPU run."""
    evs = _atom_events()
    dev = TR.replay_atom(evs, TR.AtomMirrorOps(evs, "cuda"), atol=1e-4)
    assert set(dev) == {"gn_filter", "gn_projection", "classify", "refine_iou", "refine_boxes", "cg_filter"}
    print("ATOM: max deviation per boundary call over the 8-frame trajectory:", {k: f"{v:.2e}" for k, v in dev.items()})


# ------------------------------------------------------------------------------------------------------
# PrDiMP-50: the DiMP tracker class with pytracking/parameter/dimp/prdimp50.py (Newton / KL optimiser, soft-max score preprocessing,
# 22x22 maps, relative box refinement, 10 iterations): tests/golden/tracker_prdimp50.npz.  The whole-tracker comparison on the device
# is tests/test_trackers_on_device.py.
# ------------------------------------------------------------------------------------------------------
def test_prdimp_log_covers_both_outcomes_and_the_relative_refinement():
    evs = TR.events_from_npz(load_golden("tracker_prdimp50"))
    kinds = [e["kind"] for e in evs]
    assert sum(k == "classify" for k in kinds) == int(evs[0]["n_frames"]) == 8
    assert {str(e["flag"]) for e in evs if e["kind"] == "localize"} == {"normal", "hard_negative"}
    assert {str(e["method"]) for e in evs if e["kind"] == "refine"} == {"optimize_boxes_relative"}
    assert {int(e["num_iter"]) for e in evs if e["kind"] == "optimize"} == {1, 2}
    assert next(e for e in evs if e["kind"] == "classify")["scores"].shape[-2:] == (23, 23)


def test_reference_prdimp_tracker_reproduces_its_log():
    """The recorded run is reproducible here (same seeds, same reference): guards the harness against drift."""
    _need_reference()
    from oracle import tracker_harness as TH
    import torch
    torch.set_num_threads(8)
    outs, rec, _ = TH.run_dimp(**TH.PRDIMP_RUN)
    want = TR.events_from_npz(load_golden("tracker_prdimp50"))
    got = TR.events_from_npz({k: np.asarray(v) for k, v in rec.to_npz_dict().items()})
    assert [e["kind"] for e in got] == [e["kind"] for e in want]
    for a, b in zip(got, want):
        for k in ("scores", "filter", "boxes", "iou", "target_bbox", "tv"):
            if k in b:
                np.testing.assert_allclose(np.asarray(a[k], dtype=np.float64), np.asarray(b[k], dtype=np.float64), atol=2e-5,
                                           err_msg=f"{a['kind']}.{k}")
        if "flag" in b:
            assert str(a["flag"]) == str(b["flag"])


# ------------------------------------------------------------------------------------------------------
# LWL: initialize() (box-shaped first-frame mask) + 6 x track() of the unmodified reference tracker on a stubbed backbone
# (tests/golden/tracker_lwl.npz; filters logged on every 4th channel, decoder scores on every 8th pixel).  The whole-tracker comparison on
# the device is tests/test_trackers_on_device.py.
# ------------------------------------------------------------------------------------------------------
def test_lwl_log_covers_the_growing_memory():
    evs = TR.events_from_npz(load_golden("tracker_lwl"))
    kinds = [e["kind"] for e in evs]
    assert kinds.count("lwl_init") == 1 and kinds.count("lwl_segment") == 6
    init = next(e for e in evs if e["kind"] == "lwl_init")
    assert int(init["num_iter"]) == 20 and init["filter"].shape == (1, 16, 128, 3, 3)            # 20 iterations, 16 filters 3x3
    ups = [e for e in evs if e["kind"] == "lwl_update"]
    assert [int(e["n"]) for e in ups] == [2, 3, 4, 5, 6] and {int(e["num_iter"]) for e in ups} == {3}
    assert next(e for e in evs if e["kind"] == "lwl_segment")["mask_encoding"].shape == (1, 1, 16, 30, 52)
