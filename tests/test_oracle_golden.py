"""The numpy oracle (oracle/np_oracle.py) against the golden vectors produced by the unmodified
reference (oracle/make_golden.py).  CPU only.  Tolerances: the oracle is run in float64 on the
float32 inputs, the reference computed in float32, so agreement is limited by the reference's own
rounding (measured 1e-7..1e-6 abs on these magnitudes); 2e-5 abs/rel is asserted."""
import numpy as np
import pytest

from oracle import np_oracle as O
from pytracking_amd import synth
from conftest import load_golden

ATOL = 2e-5


def close(a, b, atol=ATOL, rtol=2e-5):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), atol=atol, rtol=rtol)


@pytest.mark.parametrize("tag", ["k4", "k3", "k1", "k5"])
def test_filter_ops_single(tag):
    g = load_golden("filter_ops")
    feat, filt = g[f"{tag}_feat"].astype(np.float64), g[f"{tag}_filt"].astype(np.float64)
    close(O.apply_filter(feat, filt), g[f"{tag}_scores"])
    adj = O.apply_feat_transpose(feat, g[f"{tag}_inp"].astype(np.float64), filt.shape[-1])
    close(adj, g[f"{tag}_adj_v2"], atol=1e-4)
    close(adj, g[f"{tag}_adj_v3"], atol=1e-4)


def test_filter_ops_multifilter_and_sequences():
    g = load_golden("filter_ops")
    f64 = lambda k: g[k].astype(np.float64)
    close(O.apply_filter(f64("mf_feat"), f64("mf_filt")), g["mf_scores"])
    close(O.apply_feat_transpose(f64("mf_feat"), f64("mf_inp"), 3), g["mf_adj"], atol=1e-4)
    for s in range(2):
        close(O.apply_filter(f64("s2_feat")[:, s], f64("s2_filt")[s]), g["s2_scores"][:, s])
        close(O.apply_feat_transpose(f64("s2_feat")[:, s], f64("s2_inp")[:, s], 4), g["s2_adj"][s], atol=1e-4)


def _dimp_kwargs(cfg, **over):
    kw = dict(step_length=cfg["init_step_length"], filter_reg=cfg["init_filter_reg"],
              min_filter_reg=cfg["min_filter_reg"], feat_stride=cfg["feat_stride"],
              label_w=synth.gauss_lut(cfg["num_dist_bins"], cfg["bin_displacement"], cfg["init_gauss_sigma"]),
              mask_w=synth.mask_lut(cfg["num_dist_bins"], cfg["bin_displacement"], cfg["mask_init_factor"]),
              spatial_w=np.ones(cfg["num_dist_bins"], np.float32), bin_displacement=cfg["bin_displacement"],
              alpha_eps=cfg["alpha_eps"])
    kw.update(over)
    return kw


@pytest.mark.parametrize("name", ["dimp_sd_small_w", "dimp_sd_small_now", "dimp_sd_mid"])
def test_dimp_sd_small(name):
    g = load_golden(name)
    over = {k: float(g[k]) for k in ("filter_reg", "min_filter_reg", "alpha_eps") if k in g}
    sw = g["sw"].astype(np.float64) if "sw" in g else None
    its, losses = O.dimp_sd(g["w0"].astype(np.float64), g["feat"].astype(np.float64), g["bb"].astype(np.float64), sw,
                            num_iter=int(g["num_iter"]), **_dimp_kwargs(synth.DIMP50, **over))
    close(its, g["iterates"])
    close(losses, g["losses"], atol=1e-5, rtol=1e-4)
    close(O.apply_filter(g["feat"].astype(np.float64), its[-1]), g["scores"])


def test_dimp_sd_cfg2_n15_full_size():
    g = load_golden("dimp_sd_cfg2_n15")
    w0, feat, bb, sw = synth.dimp_problem(int(g["seed"]), int(g["n"]))
    # float64 oracle on the float32 inputs: a float32 numpy run can flip sign(s) of a near-zero score
    # (LeakyReluParDeriv is discontinuous, activation.py:43-44) and jump by ~3e-5 at a later iteration.
    f64 = lambda a: a.astype(np.float64)
    its, losses = O.dimp_sd(f64(w0), f64(feat), f64(bb), f64(sw), num_iter=5, **_dimp_kwargs(synth.DIMP50))
    close(its, g["iterates"], atol=2e-6)
    close(losses, g["losses"], atol=1e-5, rtol=1e-4)


def test_dimp_sd_cfg2_n15_twenty_iterations():
    """DiMP at 20 iterations (4x the tracker's setting; the ABI allows 64): no drift of the restatement against the
    reference over a long run -- iterates 5 / 10 / 20, all 21 losses, final scores."""
    g = load_golden("dimp_sd_cfg2_n15_it20")
    w0, feat, bb, sw = synth.dimp_problem(int(g["seed"]), int(g["n"]))
    f64 = lambda a: a.astype(np.float64)
    its, losses = O.dimp_sd(f64(w0), f64(feat), f64(bb), f64(sw), num_iter=20, **_dimp_kwargs(synth.DIMP50))
    close(its[g["which"]], g["iterates"], atol=5e-6)
    close(losses, g["losses"], atol=1e-5, rtol=1e-4)
    close(O.apply_filter(f64(feat), its[-1]), g["scores"], atol=2e-5)


def test_dimp_l2_small():
    g = load_golden("dimp_l2_small")
    f64 = lambda k: g[k].astype(np.float64)
    its, losses = O.dimp_l2_sd(f64("w0"), f64("feat"), f64("bb"), f64("sw"), num_iter=int(g["num_iter"]),
                               step_length=float(g["step_length"]), filter_reg=float(g["filter_reg"]),
                               min_filter_reg=float(g["min_filter_reg"]), feat_stride=16,
                               gauss_sigma=float(g["gauss_sigma"]), hinge_threshold=float(g["hinge_threshold"]))
    close(its, g["iterates"])
    close(losses, g["losses"], atol=1e-5, rtol=1e-4)


def _prdimp_kwargs(cfg, **over):
    kw = dict(step_length=cfg["init_step_length"], filter_reg=cfg["init_filter_reg"],
              min_filter_reg=cfg["min_filter_reg"], feat_stride=cfg["feat_stride"], gauss_sigma=cfg["gauss_sigma"],
              alpha_eps=cfg["alpha_eps"], normalize_label=cfg["normalize_label"])
    kw.update(over)
    return kw


def test_prdimp_small_and_options():
    g = load_golden("prdimp_sd_small")
    f64 = lambda k: g[k].astype(np.float64)
    its, losses = O.prdimp_sd(f64("w0"), f64("feat"), f64("bb"), f64("sw"), num_iter=int(g["num_iter"]),
                              **_prdimp_kwargs(synth.PRDIMP50))
    close(its, g["iterates"])
    close(losses, g["losses"], atol=1e-5, rtol=1e-4)
    g = load_golden("prdimp_sd_opts")
    its, losses = O.prdimp_sd(f64("w0"), f64("feat"), f64("bb"), None, num_iter=int(g["num_iter"]),
                              **_prdimp_kwargs(synth.PRDIMP50, softmax_reg_val=float(g["softmax_reg"]),
                                               uni_weight=float(g["uni_weight"]), label_shrink=float(g["label_shrink"]),
                                               label_threshold=float(g["label_threshold"])))
    close(its, g["iterates"])
    close(losses, g["losses"], atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("name", ["atom_cg_small_pr", "atom_cg_small_fr"])
def test_atom_cg_small(name):
    g = load_golden(name)
    f64 = lambda k: g[k].astype(np.float64)
    x = f64("x0")
    state = None
    for call in range(g["x_out"].shape[0]):
        x, state = O.atom_cg(x, f64("samples"), f64("y"), f64("sw"), filter_reg=synth.ATOM18["filter_reg"],
                             act_min_val=synth.ATOM18["act_min_val"], num_iter=int(g["num_iter"]),
                             fletcher_reeves=bool(g["fletcher_reeves"]), state=state)
        close(x, g["x_out"][call], atol=1e-5, rtol=1e-4)


def test_torch_port_matches_reference():
    """The torch-CPU port that bench.py times as `cpu_baseline` reproduces the reference's iterates."""
    import torch
    from oracle.frame_port import TorchCpuTracker
    g = load_golden("dimp_sd_cfg2_n15")
    tr = TorchCpuTracker(synth.DIMP50, int(g["n"]), int(g["seed"]))
    with torch.no_grad():
        w = tr.solve(tr.filter, tr.mem_feat, tr.mem_bb, tr.sw, 5)
    close(w[0].numpy(), g["iterates"][-1], atol=2e-6)
    # the float64 / dense-contraction form the closed-loop GPU test steps (same sums, no grouped convolutions): against
    # the reference golden, against the convolution form op by op (even and odd filter sizes), and over frames
    tr64 = TorchCpuTracker(synth.DIMP50, int(g["n"]), int(g["seed"]), dtype=torch.float64, gemm=True)
    with torch.no_grad():
        w64 = tr64.solve(tr64.filter, tr64.mem_feat, tr64.mem_bb, tr64.sw, 5)
    close(w64[0].numpy(), g["iterates"][-1], atol=2e-6)
    rng = np.random.default_rng(0)
    for K in (4, 3, 1, 2):
        conv, gem = (TorchCpuTracker(dict(synth.DIMP50, C=8, K=K), 3, 5, dtype=torch.float64, gemm=m) for m in (False, True))
        feat = torch.from_numpy(rng.standard_normal((3, 8, 7, 9)))
        wk = torch.from_numpy(rng.standard_normal((1, 8, K, K)))
        s0, s1 = conv.corr(feat, wk), gem.corr(feat, wk)
        assert s0.shape == s1.shape
        close(s1.numpy(), s0.numpy(), atol=1e-12)
        r = torch.from_numpy(rng.standard_normal(tuple(s0.shape)))
        close(gem.adj(feat, r, K).numpy(), conv.adj(feat, r, K).numpy(), atol=1e-12)


def test_torch_port_prdimp_matches_reference():
    """The PrDiMP variant of the frame port (softmax-Newton solve, optimizer.py:355-439; zero start filter, 22x22 maps)
    reproduces the reference's iterates: BASELINE configs[2] size (`prdimp_sd_cfg3_n50`) in float32 / convolutions and
    in float64 / dense contractions (the form the closed-loop GPU test steps), and the option branches (softmax
    regularisation, uniform weight, label shrink / threshold, one-hot label) on the small goldens."""
    import torch
    from oracle.frame_port import TorchCpuTracker
    g = load_golden("prdimp_sd_cfg3_n50")
    for kw in (dict(), dict(dtype=torch.float64, gemm=True)):
        tr = TorchCpuTracker(synth.PRDIMP50, int(g["n"]), int(g["seed"]), kind="prdimp", **kw)
        assert float(tr.filter.abs().max()) == 0.0
        with torch.no_grad():
            w = tr.solve_prdimp(tr.filter, tr.mem_feat, tr.mem_bb, tr.sw, int(g["num_iter"]))
        close(w[0].numpy(), g["iterates"][-1], atol=2e-6)

    def run_small(name, cfg, with_sw):
        gs = load_golden(name)
        tr = TorchCpuTracker(dict(cfg, C=16, H=10, W=10), gs["feat"].shape[0], 0, kind="prdimp", dtype=torch.float64, gemm=True)
        T = lambda a: torch.from_numpy(np.asarray(a, np.float64))
        sw = T(gs["sw"]) if with_sw else torch.full((gs["feat"].shape[0],), 1.0 / gs["feat"].shape[0], dtype=torch.float64)
        with torch.no_grad():
            w = tr.solve_prdimp(T(gs["w0"])[None], T(gs["feat"]), T(gs["bb"]), sw, int(gs["num_iter"]))
        close(w[0].numpy(), gs["iterates"][-1], atol=2e-6)
        return gs
    run_small("prdimp_sd_small", synth.PRDIMP50, True)
    go = load_golden("prdimp_sd_opts")
    run_small("prdimp_sd_opts", dict(synth.PRDIMP50, softmax_reg=float(go["softmax_reg"]), init_uni_weight=float(go["uni_weight"]),
                                     label_shrink=float(go["label_shrink"]), label_threshold=float(go["label_threshold"])), False)
    run_small("prdimp_sd_sigma0", dict(synth.PRDIMP50, gauss_sigma=0.0), True)


@pytest.mark.parametrize("name", ["lwl_gn_small_full", "lwl_gn_small_img", "lwl_gn_small_none", "lwl_gn_mid"])
def test_lwl_gn_sd(name):
    """LWL few-shot learner (GNSteepestDescent on LWTLResidual, steepestdescent.py:32-105) vs the reference run."""
    g = load_golden(name)
    sw = g["sw"] if g["sw"].size else None
    its, losses = O.lwl_gn_sd(g["w0"].astype(np.float64), g["feat"].astype(np.float64), g["label"].astype(np.float64),
                              None if sw is None else sw.astype(np.float64), num_iter=int(g["num_iter"]),
                              filter_reg=float(g["filter_reg"]), steplength_reg=float(g["steplength_reg"]))
    np.testing.assert_allclose(its, g["iterates"], atol=2e-5)
    np.testing.assert_allclose(np.array(losses), g["losses"], rtol=1e-4, atol=1e-7)
    s = O.apply_filter(g["feat"].astype(np.float64), its[-1])
    np.testing.assert_allclose(s, g["scores"], atol=2e-5)


@pytest.mark.parametrize("name", ["atom_gn_small_fr", "atom_gn_small_pr", "atom_gn_mid"])
def test_atom_joint_gauss_newton(name):
    """ATOM first-frame GaussNewtonCG on FactorizedConvProblem (optimization.py:328-421, atom/optim.py:6-68)."""
    g = load_golden(name)
    f64 = lambda a: a.astype(np.float64)
    f, P = O.atom_gn_cg(f64(g["f0"]), f64(g["P0"]), f64(g["samples"]), f64(g["y"]), f64(g["sw"]),
                        filter_reg=float(g["filter_reg"]), projection_reg=float(g["projection_reg"]),
                        act_min_val=float(g["act_min_val"]), cg_iters=[int(v) for v in g["cg_iters"]],
                        fletcher_reeves=bool(int(g["fletcher_reeves"])))
    np.testing.assert_allclose(f, g["f_out"], atol=5e-5)
    np.testing.assert_allclose(P, g["P_out"], atol=5e-5)


@pytest.mark.parametrize("name,cfg", [("tomp_small", "TOMP_SMALL"), ("tomp_full", "TOMP")])
def test_tomp_model_predictor(name, cfg):
    """ToMP FilterPredictor (both entry points) + LinearFilterClassifier + DenseBoxRegressor
    (filter_predictor.py:50-150, transformer.py:66-262, heads.py:83-141)."""
    from oracle import tomp_oracle as TO
    g = load_golden(name)
    cfg = getattr(synth, cfg)
    p = {k: v.astype(np.float64) for k, v in synth.tomp_params(int(g["seed"]), cfg).items()}
    train, test, lab, ltrb = [a.astype(np.float64) for a in synth.tomp_inputs(int(g["seed"]) + 1, cfg)]
    if "train" in g:
        np.testing.assert_array_equal(g["train"], train.astype(np.float32))
    a = (cfg["nhead"], cfg["n_enc"], cfg["n_dec"], cfg["feature_sz"])
    np.testing.assert_allclose(TO.posenc(cfg["H"], cfg["W"], cfg["D"], cfg["feature_sz"]).T.reshape(g["pos"].shape),
                               g["pos"], atol=2e-6)
    cw, bw, cenc, benc = TO.predict_cls_bbreg_filters_parallel(p, train, test, lab, cfg["num_gth_frames"], ltrb, *a)
    np.testing.assert_allclose(cw, g["cls_filter"], atol=2e-5)
    np.testing.assert_allclose(bw, g["bbreg_filter"], atol=2e-5)
    np.testing.assert_allclose(cenc, g["cls_enc"], atol=2e-5)
    np.testing.assert_allclose(benc, g["bbreg_enc"], atol=2e-5)
    np.testing.assert_allclose(TO.linear_filter_classifier(p, cenc, cw), g["scores"], atol=5e-5)
    np.testing.assert_allclose(TO.dense_box_regressor(p, benc, bw), g["ltrb"], rtol=1e-4, atol=1e-5)
    if cfg is synth.TOMP_SMALL:
        w1, enc1 = TO.predict_filter(p, train, test, lab, ltrb, *a)
        np.testing.assert_allclose(w1[0], g["single_filter"], atol=2e-5)
        np.testing.assert_allclose(enc1, g["single_enc"], atol=2e-5)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_clf_head(tag):
    """Conv2d 3x3 (no bias) + InstanceL2Norm as residual_bottleneck builds it (features.py:49-73)."""
    g = load_golden("clf_head")
    y = O.clf_head(g[f"{tag}_x"].astype(np.float64), g[f"{tag}_w"].astype(np.float64), float(g[f"{tag}_scale"]))
    np.testing.assert_allclose(y, g[f"{tag}_y"], atol=2e-6, rtol=2e-5)


def test_localize_frame_constants_and_decision():
    """Host logic of pytracking_amd.localization (the per-frame constants handed to the localisation kernel) + the oracle's
    restatement of the outcome (np_oracle.localize_decide), against DiMP.localize_advanced run by the unmodified
    reference (tests/golden/localize.npz): flag, scale, translation vector."""
    import ctypes
    import torch
    from pytracking_amd import localization as LM
    from localize_cases import cases, constants
    seen = set()
    for me, c in cases(load_golden("localize")):
        S, H, W = c["scores"].shape
        _, qd = constants(me, (S, H, W), torch.from_numpy(c["sample_pos"]), torch.from_numpy(c["sample_scales"]))
        out = O.localize_decide(c["scores"], c["scores"], qd)
        flag = O.LOC_FLAGS[int(out[0])]
        assert flag == str(c["flag"]) and int(out[1]) == int(c["scale_ind"])
        np.testing.assert_array_equal(out[4:6].astype(np.float32), c["tv"])       # same float32 operations, same order
        seen.add((flag, int(out[12])))
    assert {f for f, _ in seen} == {"normal", "not_found", "uncertain", "hard_negative"} and ("hard_negative", 2) in seen


@pytest.mark.parametrize("tag,relative", [("default", False), ("default_decay", False), ("relative", True)])
def test_iou_refinement(tag, relative):
    """IoU-guided box refinement (dimp.py:725-788 on atom_iou_net.py:96-136): float64 restatement vs the reference run."""
    import torch
    from oracle import iou_oracle as IO
    g = load_golden("iou_refine")
    t64 = lambda a: torch.from_numpy(a.astype(np.float64))
    p = {k[2:]: t64(v) for k, v in g.items() if k.startswith("w_")}
    iters, step, decay = g[f"{tag}_cfg"]
    boxes, iou = IO.refine(p, (t64(g["mod3"]), t64(g["mod4"])), (t64(g["c3"]), t64(g["c4"])), t64(g["boxes"]), int(iters),
                           float(step), float(decay), relative)
    np.testing.assert_allclose(boxes.numpy(), g[f"{tag}_boxes"], rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(iou.numpy(), g[f"{tag}_iou"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("tag,relative", [("atom_default", False), ("atom_relative", True), ("atom_nodecay", False)])
def test_iou_refinement_atom_backtracking(tag, relative):
    """ATOM.optimize_boxes (atom.py:758-836) incl. the per-proposal backtracking, vs the reference run."""
    import torch
    from oracle import iou_oracle as IO
    g = load_golden("iou_refine")
    t64 = lambda a: torch.from_numpy(a.astype(np.float64))
    p = {k[2:]: t64(v) for k, v in g.items() if k.startswith("w_")}
    iters, step, decay = g[f"{tag}_cfg"]
    boxes, iou, backtracks = IO.refine_atom(p, (t64(g["mod3"]), t64(g["mod4"])), (t64(g["c3"]), t64(g["c4"])), t64(g["boxes"]),
                                            int(iters), float(step), float(decay), relative)
    np.testing.assert_allclose(boxes.numpy(), g[f"{tag}_boxes"], rtol=2e-5, atol=5e-4)
    np.testing.assert_allclose(iou.numpy(), g[f"{tag}_iou"], rtol=2e-5, atol=2e-5)
    assert (backtracks > 0) == (decay < 1), backtracks


# ------------------------------------------------------------------------------------------------------
# round 2: full-size / deployed-size vectors and the branches added by oracle/make_golden.py gen_branches()
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["dimp_sd_bentpar", "dimp_sd_linmask", "dimp_sd_bentpar_linmask"])
def test_dimp_sd_activation_branches(name):
    """score_act='bentpar' (activation.py:47-66) and mask_act='linear' (optimizer.py:57-66)."""
    g = load_golden(name)
    f64 = lambda k: g[k].astype(np.float64)
    mask_act, score_act = str(g["mask_act"]), str(g["score_act"])
    cfg = synth.DIMP50
    over = dict(mask_act=mask_act, score_act=score_act, act_param=float(g["act_param"]) or None,
                mask_w=synth.mask_lut(cfg["num_dist_bins"], cfg["bin_displacement"], cfg["mask_init_factor"], mask_act))
    its, losses = O.dimp_sd(f64("w0"), f64("feat"), f64("bb"), f64("sw"), num_iter=int(g["num_iter"]),
                            **_dimp_kwargs(cfg, **over))
    close(its, g["iterates"])
    close(losses, g["losses"], atol=1e-5, rtol=1e-4)


def test_prdimp_sd_one_hot_label():
    """gauss_sigma = 0: one-hot label at the nearest cell (optimizer.py:334-341)."""
    g = load_golden("prdimp_sd_sigma0")
    f64 = lambda k: g[k].astype(np.float64)
    c = synth.PRDIMP50
    its, losses = O.prdimp_sd(f64("w0"), f64("feat"), f64("bb"), f64("sw"), num_iter=int(g["num_iter"]),
                              step_length=c["init_step_length"], filter_reg=c["init_filter_reg"],
                              min_filter_reg=c["min_filter_reg"], feat_stride=c["feat_stride"], gauss_sigma=0.0,
                              alpha_eps=c["alpha_eps"], normalize_label=c["normalize_label"])
    close(its, g["iterates"])
    close(losses, g["losses"], atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("name", ["dimp_sd_two_sequences", "prdimp_sd_two_sequences"])
def test_sd_two_sequences(name):
    """Two sequences in one call: each sequence is an independent solve; the loss is averaged over sequences
    (optimizer.py:143, :396)."""
    g = load_golden(name)
    S = g["w0"].shape[0]
    tot = 0.0
    for s in range(S):
        a = [g[k].astype(np.float64) for k in ("w0", "feat", "bb", "sw")]
        args = (a[0][s], a[1][:, s], a[2][:, s], a[3][:, s])
        if name.startswith("dimp"):
            its, losses = O.dimp_sd(*args, num_iter=int(g["num_iter"]), **_dimp_kwargs(synth.DIMP50))
        else:
            c = synth.PRDIMP50
            its, losses = O.prdimp_sd(*args, num_iter=int(g["num_iter"]), step_length=c["init_step_length"],
                                      filter_reg=c["init_filter_reg"], min_filter_reg=c["min_filter_reg"],
                                      feat_stride=c["feat_stride"], gauss_sigma=c["gauss_sigma"], alpha_eps=c["alpha_eps"],
                                      normalize_label=c["normalize_label"])
        close(its, g["iterates"][:, s])
        tot = tot + np.array(losses)
    close(tot / S, g["losses"][:, 0], atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("tag", ["pr", "fr"])
def test_atom_cg_direction_forgetting(tag):
    """direction_forget_factor != 0: (p, rho, r_prev) carried across three run() calls (optimization.py:82-85)."""
    g = load_golden(f"atom_cg_forget_{tag}")
    f64 = lambda k: g[k].astype(np.float64)
    x, state = f64("x0"), None
    for call, iters in enumerate(g["iters"]):
        x, state = O.atom_cg(x, f64("samples"), f64("y"), f64("sw"), filter_reg=synth.ATOM18["filter_reg"],
                             act_min_val=synth.ATOM18["act_min_val"], num_iter=int(iters),
                             fletcher_reeves=bool(int(g["fletcher_reeves"])), state=state,
                             direction_forget_factor=float(g["forget"]))
        close(x, g["x_out"][call], atol=2e-5, rtol=1e-4)
    # and the state really mattered: a reset before the second call gives a different iterate
    x1, _ = O.atom_cg(f64("x0"), f64("samples"), f64("y"), f64("sw"), filter_reg=synth.ATOM18["filter_reg"],
                      act_min_val=synth.ATOM18["act_min_val"], num_iter=3, fletcher_reeves=bool(int(g["fletcher_reeves"])))
    x2, _ = O.atom_cg(x1, f64("samples"), f64("y"), f64("sw"), filter_reg=synth.ATOM18["filter_reg"],
                      act_min_val=synth.ATOM18["act_min_val"], num_iter=2, fletcher_reeves=bool(int(g["fletcher_reeves"])))
    assert np.abs(x2 - g["x_out"][1]).max() > 1e-4


@pytest.mark.parametrize("tag,fr", [("pr", False), ("fr", True)])
def test_atom_gn_first_frame_schedule(tag, fr):
    """ATOM first frame at the deployed 6 x 10 schedule (parameter/atom/default.py:27-28), 30 x 256 x 18 x 18."""
    g = load_golden("atom_gn_first_frame")
    f0, P0, samples, y, sw = synth.atom_gn_problem(int(g["seed"]))
    f64 = lambda a: a.astype(np.float64)
    rf, rP = O.atom_gn_cg(f64(f0), f64(P0), f64(samples), f64(y), f64(sw), filter_reg=float(g["filter_reg"]),
                          projection_reg=float(g["projection_reg"]), act_min_val=float(g["act_min_val"]),
                          cg_iters=[10] * 6, fletcher_reeves=fr)
    close(rf, g[f"f_out_{tag}"], atol=2e-5)
    close(rP, g[f"P_out_{tag}"], atol=2e-5)


def test_lwl_gn_config5_full_size_first_iteration():
    """BASELINE configs[4] size (n=32, 512x30x52, 16 filters): the first GN iteration and both initial losses; the
    later iterates are checked on the GPU (the float64 numpy passes take ~10 s each here)."""
    g = load_golden("lwl_gn_cfg5_n32")
    w0, feat, label, sw = synth.lwl_problem(int(g["seed"]))
    f64 = lambda a: a.astype(np.float64)
    its, losses = O.lwl_gn_sd(f64(w0), f64(feat), f64(label), f64(sw), num_iter=1, filter_reg=float(g["filter_reg"]))
    close(its[1], g["iterate1"], atol=2e-5)
    close(losses, g["losses4"][:2], atol=1e-7, rtol=1e-4)
    np.testing.assert_array_equal(g["losses3"], g["losses4"][:4])          # 3 iterations are the prefix of 4


def test_lwl_gn_config5_first_frame_twenty_iterations():
    """LWL's first-frame optimisation: `net_opt_iter = 20` (lwl_ytvos.py:30) on ONE sample at the configs[4] geometry
    (512 x 30 x 52, 16 filters), float64 restatement against the reference's autograd run; the n = 32 run of the same
    length is checked on the GPU only (its float64 numpy passes take minutes here)."""
    g = load_golden("lwl_gn_cfg5_n1_it20")
    w0, feat, label, sw = synth.lwl_problem(int(g["seed"]), dict(synth.LWL, n=int(g["n"])))
    f64 = lambda a: a.astype(np.float64)
    its, losses = O.lwl_gn_sd(f64(w0), f64(feat), f64(label), f64(sw), num_iter=20, filter_reg=float(g["filter_reg"]))
    close(its[-1], g["final"], atol=2e-5)
    close(its[10], g["iterate10"].astype(np.float64), atol=2e-3, rtol=2e-3)         # stored as float16
    close(losses, g["losses"], atol=1e-7, rtol=1e-4)
    close(O.apply_filter(f64(feat[:1]), its[-1]), g["scores_first"], atol=2e-5)


@pytest.mark.parametrize("tag,relative", [("default", False), ("relative", True)])
def test_iou_refinement_deployed_size(tag, relative):
    """256-channel IoU features, 256-wide LinearBlocks, 10 proposals (dimpnet.py:189-190): the float64 oracle against
    the reference's float32 run.  The reference's own rounding at these box magnitudes (100-200 px, ulp 1.5e-5) is
    what the 2e-4 px bound covers (measured 2.7e-5 / 9.3e-5 px)."""
    import torch
    from oracle import iou_oracle as IO
    g = load_golden("iou_refine_full")
    p = synth.iou_net_params(int(g["param_seed"]))
    c3, c4, m3, m4, boxes = synth.iou_inputs(int(g["input_seed"]))
    t64 = lambda a: torch.from_numpy(a.astype(np.float64))
    iters, step, decay = g[f"{tag}_cfg"]
    rb, ri = IO.refine({k: t64(v) for k, v in p.items()}, (t64(m3), t64(m4)), (t64(c3), t64(c4)), t64(boxes), int(iters),
                       float(step), float(decay), relative)
    close(ri.numpy(), g[f"{tag}_iou"], atol=2e-5)
    close(rb.numpy(), g[f"{tag}_boxes"], atol=2e-4, rtol=0)
    assert np.abs(g[f"{tag}_boxes"] - boxes).max() > 2.0


def test_sample_patch_geometry_and_pixels():
    """`sample_patch` / `sample_patch_multiscale` (preprocessing.py:33-148): the host geometry of
    pytracking_amd/preprocessing.py and the numpy restatement of the pixel part against 24 reference runs covering every
    border mode, pre-downsampling strides 1..8 and crops hanging over every border."""
    from pytracking_amd import preprocessing as PP
    g = load_golden("sample_patch")
    im = g["im"][0]
    strides = set()
    for k in range(int(g["n"])):
        msc = float(g[f"c{k}_msc"])
        geom, coord = PP.patch_geometry(im.shape[-2:], g[f"c{k}_pos"], g[f"c{k}_ssz"], g[f"c{k}_osz"], str(g[f"c{k}_mode"]),
                                        None if msc < 0 else msc)
        np.testing.assert_array_equal(np.array(coord, np.float32), g[f"c{k}_coord"][0])
        out = O.sample_patch_pixels(im, geom.df, geom.os0, geom.os1, geom.tl0, geom.tl1, geom.crop_h, geom.crop_w,
                                    tuple(int(v) for v in g[f"c{k}_osz"]))
        # pixel values are 0..255: with the source index formed by ONE fused multiply-add, as this image's ATen does, the
        # restatement is within 3 ulp of a pixel (3.1e-5; 17 of 24 patches bit-identical); with the unfused index it was 5e-4
        np.testing.assert_allclose(out, g[f"c{k}_patch"][0], atol=1e-4, rtol=0)
        strides.add(geom.df)
    assert {1, 2, 3} <= strides


@pytest.mark.parametrize("tag", ["s", "n", "d"])
def test_augmentation_set(tag):
    """`sample_patch_transformed` (preprocessing.py:13-30) over lists of the reference's transforms (augmentation.py: Identity,
    Translation, FlipHorizontal, FlipVertical, Blur, Scale; with / without output size; shifts that crop and that pad; `d` = the
    DiMP-50 first-frame list at 288 / 576): the numpy restatement against the reference's output.  Pixels 0..255, bound 1e-4
    (measured <= 4.6e-5 = 3 ulp: blend contraction / convolution summation order; crops, flips and shifts are exact)."""
    from pytracking_amd import preprocessing as PP
    from augment_cases import read_case
    c = read_case(load_golden("augment"), tag)
    im = c["im"][0]
    ssz = [c["scale"] * float(c["image_sz"][0]), c["scale"] * float(c["image_sz"][1])]
    geom, _ = PP.patch_geometry(im.shape[-2:], c["pos"], ssz, c["image_sz"])
    patch = O.sample_patch_pixels(im, geom.df, geom.os0, geom.os1, geom.tl0, geom.tl1, geom.crop_h, geom.crop_w,
                                  tuple(int(v) for v in c["image_sz"]))
    out = O.augment_patch(patch, c["specs"])
    assert out.shape == c["shape"]
    st = c["stride"]
    np.testing.assert_allclose(out[:, :, ::st, ::st], c["out"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out.astype(np.float64).sum(axis=(1, 2, 3)), c["sums"], rtol=1e-7)
    # the host descriptors of the product agree with the oracle's reading of the same list (output size, pads)
    descs, taps, hw = PP.transform_descriptors(c["objs"], patch.shape[1:])
    assert hw == c["shape"][2:]
    for d, sp in zip(descs, c["specs"]):
        th = d.th
        pad_h = 0.0 if sp["output_sz"] is None else (sp["output_sz"][0] - th) / 2
        assert d.pad_top == int(np.floor(pad_h)) + sp["shift"][0]


@pytest.mark.parametrize("tag", ["pr", "fr", "pr0", "fr0"])
def test_atom_cg_compressed_channels(tag):
    """The oracle's CG (optimization.py:72-163, 227-289) at ATOM's compressed-channel shape class (C = 64, 4x4), PR / FR, with and
    without direction forgetting, three run() calls with the state carried over."""
    g = load_golden("atom_cg_c64")
    x0, samples, y, sw = synth.atom_problem(int(g["seed"]), int(g["n"]), small=dict(C=64, H=int(g["H"]), W=int(g["W"])))
    f64 = lambda a: a.astype(np.float64)
    x, state = f64(x0), None
    for call, iters in enumerate(g["iters"]):
        x, state = O.atom_cg(x, f64(samples), f64(y), f64(sw), filter_reg=synth.ATOM18["filter_reg"],
                             act_min_val=synth.ATOM18["act_min_val"], num_iter=int(iters), fletcher_reeves=bool(int(g[f"{tag}_fr"])),
                             direction_forget_factor=float(g[f"{tag}_forget"]), state=state)
        close(x, g[f"{tag}_x_out"][call], atol=2e-5, rtol=1e-4)
