"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz by running the UNMODIFIED reference.

Run in the build container only (needs /root/reference):
    python -B oracle/make_golden.py
Inputs come from the seeded generators in pytracking_amd/synth.py; for the small cases the inputs
are stored next to the outputs, for the BASELINE-sized cases only (seed, shape) + outputs are
stored and the inputs are regenerated from the seed by the tests.

Reference entry points executed (all on CPU, fp32, torch.no_grad()):
  ltr.models.layers.filter.apply_filter / apply_feat_transpose          (filter.py:5,91)
  ltr.models.target_classifier.optimizer.DiMPSteepestDescentGN          (optimizer.py:11)
  ltr.models.target_classifier.optimizer.DiMPL2SteepestDescentGN        (optimizer.py:174)
  ltr.models.target_classifier.optimizer.PrDiMPSteepestDescentNewton    (optimizer.py:294)
  pytracking.libs.optimization.ConjugateGradient + atom.optim.ConvProblem (optimization.py:227, optim.py:71)
  ltr.models.target_classifier.initializer.FilterInitializerLinear      (initializer.py:118; PrRoIPool = restatement)
  ltr.models.bbreg.atom_iou_net.AtomIoUNet.predict_iou                  (atom_iou_net.py:96; PrRoIPool = restatement)
  ltr.models.transformer.filter_predictor.FilterPredictor, heads.{LinearFilterClassifier, DenseBoxRegressor}
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402

ref_harness.install()

from pytracking_amd import synth  # noqa: E402
import ltr.models.layers.filter as rfilter  # noqa: E402
import ltr.models.target_classifier.optimizer as roptim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
T = torch.from_numpy


def save(name, **kw):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in kw.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def gen_filter_ops():
    rng = np.random.default_rng(11)
    out = {}
    cases = [("k4", 3, 8, 6, 7, 4), ("k3", 2, 5, 9, 9, 3), ("k1", 2, 6, 5, 4, 1), ("k5", 2, 4, 8, 6, 5)]
    for tag, n, C, H, W, K in cases:
        feat = rng.standard_normal((n, C, H, W), dtype=np.float32)
        filt = rng.standard_normal((1, C, K, K), dtype=np.float32)
        with torch.no_grad():
            s = rfilter.apply_filter(T(feat), T(filt))                         # (n,1,O,O)
            inp = torch.from_numpy(rng.standard_normal(tuple(s.shape), dtype=np.float32))
            g2 = rfilter.apply_feat_transpose(T(feat), inp, (K, K), training=False)
            g3 = rfilter.apply_feat_transpose(T(feat), inp, (K, K), training=True)
        out.update({f"{tag}_feat": feat, f"{tag}_filt": filt[0], f"{tag}_scores": s[:, 0].numpy(),
                    f"{tag}_inp": inp[:, 0].numpy(), f"{tag}_adj_v2": g2[0].numpy(), f"{tag}_adj_v3": g3[0].numpy()})
    # multi-filter (LWL-style): feat (n,S=1,C,H,W), filter (1,F,C,K,K)
    n, C, H, W, K, F = 2, 6, 7, 8, 3, 4
    feat = rng.standard_normal((n, 1, C, H, W), dtype=np.float32)
    filt = rng.standard_normal((1, F, C, K, K), dtype=np.float32)
    with torch.no_grad():
        s = rfilter.apply_filter(T(feat), T(filt))                              # (n,1,F,H,W)
        inp = torch.from_numpy(rng.standard_normal(tuple(s.shape), dtype=np.float32))
        g = rfilter.apply_feat_transpose(T(feat), inp, (K, K), training=False)  # (1,F,C,K,K)
    out.update(mf_feat=feat[:, 0], mf_filt=filt[0], mf_scores=s[:, 0].numpy(), mf_inp=inp[:, 0].numpy(),
               mf_adj=g[0].numpy())
    # two sequences (S=2) exercise the grouped path
    n, S, C, H, W, K = 3, 2, 4, 6, 6, 4
    feat = rng.standard_normal((n, S, C, H, W), dtype=np.float32)
    filt = rng.standard_normal((S, C, K, K), dtype=np.float32)
    with torch.no_grad():
        s = rfilter.apply_filter(T(feat), T(filt))
        inp = torch.from_numpy(rng.standard_normal(tuple(s.shape), dtype=np.float32))
        g = rfilter.apply_feat_transpose(T(feat), inp, (K, K), training=False)
    out.update(s2_feat=feat, s2_filt=filt, s2_scores=s.numpy(), s2_inp=inp.numpy(), s2_adj=g.numpy())
    save("filter_ops", **out)


def _dimp_module(cfg):
    m = roptim.DiMPSteepestDescentGN(
        num_iter=cfg["num_iter"], feat_stride=cfg["feat_stride"], init_step_length=cfg["init_step_length"],
        init_filter_reg=cfg["init_filter_reg"], init_gauss_sigma=cfg["init_gauss_sigma"],
        num_dist_bins=cfg["num_dist_bins"], bin_displacement=cfg["bin_displacement"],
        mask_init_factor=cfg["mask_init_factor"], score_act=cfg["score_act"], mask_act=cfg["mask_act"],
        min_filter_reg=cfg["min_filter_reg"], alpha_eps=cfg["alpha_eps"])
    return m.eval()


def _prdimp_module(cfg):
    m = roptim.PrDiMPSteepestDescentNewton(
        num_iter=cfg["num_iter"], feat_stride=cfg["feat_stride"], init_step_length=cfg["init_step_length"],
        init_filter_reg=cfg["init_filter_reg"], gauss_sigma=cfg["gauss_sigma"], min_filter_reg=cfg["min_filter_reg"],
        alpha_eps=cfg["alpha_eps"], init_uni_weight=cfg["init_uni_weight"], normalize_label=cfg["normalize_label"],
        label_shrink=cfg["label_shrink"], softmax_reg=cfg["softmax_reg"], label_threshold=cfg["label_threshold"])
    return m.eval()


def _run_opt(mod, w0, feat, bb, sw, num_iter):
    with torch.no_grad():
        w, its, losses = mod(T(w0)[None], T(feat), T(bb), sample_weight=None if sw is None else T(sw),
                             num_iter=num_iter, compute_losses=True)
        scores = rfilter.apply_filter(T(feat), w)
    return (torch.stack([i[0] for i in its]).numpy(), torch.stack([l.reshape(()) for l in losses]).numpy(),
            scores[:, 0].numpy())


def gen_dimp():
    cfg = synth.DIMP50
    small = dict(C=16, H=10, W=10)
    for tag, n, sm, ww, it in [("small_w", 4, small, True, 3), ("small_now", 3, small, False, 2)]:
        w0, feat, bb, sw = synth.dimp_problem(21, n, cfg, with_weights=ww, small=sm)
        its, losses, scores = _run_opt(_dimp_module(cfg), w0, feat, bb, sw, it)
        kw = dict(w0=w0, feat=feat, bb=bb, iterates=its, losses=losses, scores=scores, num_iter=it)
        if sw is not None:
            kw["sw"] = sw
        save(f"dimp_sd_{tag}", **kw)
    # mutated run-time attributes (dimp.py:589-602) + mid-sized map
    w0, feat, bb, sw = synth.dimp_problem(22, 6, cfg, small=dict(C=32, H=18, W=18))
    mod = _dimp_module(cfg)
    mod.filter_reg.data[0] = 0.05
    mod.min_filter_reg = 0.2
    mod.alpha_eps = 0.01
    its, losses, scores = _run_opt(mod, w0, feat, bb, sw, 4)
    save("dimp_sd_mid", w0=w0, feat=feat, bb=bb, sw=sw, iterates=its, losses=losses, scores=scores, num_iter=4,
         filter_reg=0.05, min_filter_reg=0.2, alpha_eps=0.01)
    # BASELINE config 2: n=50, 512x18x18, K=4, 5 iterations -- inputs regenerated from the seed
    for seed, n in [(1234, 50), (1235, 15)]:
        w0, feat, bb, sw = synth.dimp_problem(seed, n, cfg)
        its, losses, scores = _run_opt(_dimp_module(cfg), w0, feat, bb, sw, 5)
        save(f"dimp_sd_cfg2_n{n}", seed=seed, n=n, iterates=its, losses=losses, scores=scores, num_iter=5)


def gen_dimp_l2():
    cfg = synth.DIMP50
    w0, feat, bb, sw = synth.dimp_problem(31, 5, cfg, small=dict(C=16, H=10, W=10))
    mod = roptim.DiMPL2SteepestDescentGN(num_iter=3, feat_stride=16, init_step_length=1.0, gauss_sigma=0.9,
                                         hinge_threshold=0.05, init_filter_reg=0.1, min_filter_reg=1e-3,
                                         alpha_eps=0.0).eval()
    its, losses, scores = _run_opt(mod, w0, feat, bb, sw, 3)
    save("dimp_l2_small", w0=w0, feat=feat, bb=bb, sw=sw, iterates=its, losses=losses, scores=scores, num_iter=3,
         gauss_sigma=0.9, hinge_threshold=0.05, step_length=1.0, filter_reg=0.1, min_filter_reg=1e-3)


def gen_prdimp():
    cfg = synth.PRDIMP50
    w0, feat, bb, sw = synth.dimp_problem(41, 4, cfg, small=dict(C=16, H=10, W=10))
    w0 = w0 * 0
    its, losses, scores = _run_opt(_prdimp_module(cfg), w0, feat, bb, sw, 3)
    save("prdimp_sd_small", w0=w0, feat=feat, bb=bb, sw=sw, iterates=its, losses=losses, scores=scores, num_iter=3)
    # exercise softmax_reg / uniform weight / shrink / threshold options + no sample weights
    c2 = dict(cfg, softmax_reg=-1.0, init_uni_weight=0.1, label_shrink=0.05, label_threshold=1e-4)
    w0, feat, bb, _ = synth.dimp_problem(42, 3, cfg, with_weights=False, small=dict(C=16, H=10, W=10))
    its, losses, scores = _run_opt(_prdimp_module(c2), w0, feat, bb, None, 2)
    save("prdimp_sd_opts", w0=w0, feat=feat, bb=bb, iterates=its, losses=losses, scores=scores, num_iter=2,
         softmax_reg=-1.0, uni_weight=0.1, label_shrink=0.05, label_threshold=1e-4)
    # BASELINE config 3 shape: n=50, 512x22x22
    w0, feat, bb, sw = synth.dimp_problem(2234, 50, cfg)
    w0 = w0 * 0
    its, losses, scores = _run_opt(_prdimp_module(cfg), w0, feat, bb, sw, 5)
    save("prdimp_sd_cfg3_n50", seed=2234, n=50, iterates=its, losses=losses, scores=scores, num_iter=5)


def gen_atom_cg():
    from pytracking import TensorList
    from pytracking.libs import optimization
    from pytracking.tracker.atom.optim import ConvProblem
    from ltr.models.layers import activation
    cfg = synth.ATOM18

    def run(tag, seed, n, small, iters, fletcher_reeves=False, calls=1):
        x0, samples, y, sw = synth.atom_problem(seed, n, cfg, small=small)
        act = activation.MLU(cfg["act_min_val"])
        prob = ConvProblem(TensorList([T(samples)]), TensorList([T(y)[:, None]]), TensorList([cfg["filter_reg"]]),
                           TensorList([T(sw)]), act)
        x = TensorList([T(x0.copy())[None].clone()])
        opt = optimization.ConjugateGradient(prob, x, fletcher_reeves=fletcher_reeves, direction_forget_factor=0)
        outs = []
        for _ in range(calls):
            opt.run(iters)
            outs.append(x[0].detach()[0].numpy().copy())
        kw = dict(x_out=np.stack(outs), num_iter=iters, fletcher_reeves=int(fletcher_reeves))
        if small is not None:
            kw.update(x0=x0, samples=samples, y=y, sw=sw)
        else:
            kw.update(seed=seed, n=n)
        save(f"atom_cg_{tag}", **kw)

    run("small_pr", 51, 5, dict(C=8, H=10, W=10), 4, calls=2)
    run("small_fr", 52, 4, dict(C=8, H=9, W=11), 3, fletcher_reeves=True)
    run("cfg1_n250", 1250, 250, None, 5)


def gen_atom_gn():
    """ATOM first-frame joint optimisation: the reference's GaussNewtonCG on FactorizedConvProblem (CPU, autograd)."""
    from pytracking import TensorList
    from pytracking.libs import optimization
    from pytracking.tracker.atom.optim import FactorizedConvProblem
    from ltr.models.layers import activation
    cfg = synth.ATOM18

    def run(tag, seed, n, M, Kc, H, W, cg_iters, fr):
        rng = np.random.default_rng(seed)
        samples = (rng.standard_normal((n, M, H, W), dtype=np.float32) * np.float32(0.1))
        _, _, y, sw = synth.atom_problem(seed, n, cfg, small=dict(C=Kc, H=H, W=W))
        f0 = rng.standard_normal((Kc, 4, 4), dtype=np.float32) * np.float32(0.05)
        P0 = rng.standard_normal((Kc, M), dtype=np.float32) * np.float32(1.0 / np.sqrt(M))
        act = activation.MLU(cfg["act_min_val"])
        prob = FactorizedConvProblem(TensorList([T(samples)]), TensorList([T(y)[:, None]]), TensorList([cfg["filter_reg"]]),
                                     TensorList([1e-4]), None, TensorList([T(sw)]), lambda x: x, act)
        filt = T(f0.copy())[None].clone()
        proj = T(P0.copy())[:, :, None, None].clone()
        var = TensorList([filt]).concat(TensorList([proj]))
        opt = optimization.GaussNewtonCG(prob, var, fletcher_reeves=fr)
        opt.run(list(cg_iters))
        save(f"atom_gn_{tag}", samples=samples, y=y, sw=sw, f0=f0, P0=P0, f_out=var[0].detach()[0].numpy(),
             P_out=var[1].detach()[:, :, 0, 0].numpy(), cg_iters=np.array(cg_iters), fletcher_reeves=int(fr),
             filter_reg=cfg["filter_reg"], projection_reg=1e-4, act_min_val=cfg["act_min_val"])

    run("small_fr", 61, 4, 12, 8, 10, 10, [3, 3], True)
    run("small_pr", 62, 3, 16, 8, 9, 11, [2, 2, 2], False)
    run("mid", 63, 6, 32, 16, 18, 18, [4, 4], True)


def gen_prroi_consumers():
    """Reference modules that consume PrRoIPool, executed with the restatement plugged in."""
    from ltr.models.target_classifier.initializer import FilterInitializerLinear
    from ltr.models.bbreg.atom_iou_net import AtomIoUNet
    torch.manual_seed(7)
    rng = np.random.default_rng(61)
    # filter initializer (initializer.py:151-173): conv3x3 -> PrRoIPool(4,4,1/16) -> mean over images
    init = FilterInitializerLinear(filter_size=4, filter_norm=False, feature_dim=16).eval()
    feat = synth.clf_features(rng, 5, 16, 18, 18, 4)
    bb = synth.target_boxes(rng, 5)
    with torch.no_grad():
        w = init(T(feat)[:, None], T(bb)[:, None])
        conv = init.filter_conv(T(feat))
    save("filter_init_linear", feat=feat, bb=bb, conv_w=init.filter_conv.weight.detach().numpy(),
         conv_b=init.filter_conv.bias.detach().numpy(), conv_out=conv.numpy(), weights=w.numpy())
    # IoU predictor test branch (atom_iou_net.py:96-136) incl. gradient w.r.t. the proposals
    net = AtomIoUNet(input_dim=(32, 64), pred_input_dim=(16, 16), pred_inter_dim=(16, 16)).eval()
    c3 = torch.from_numpy(rng.standard_normal((1, 16, 36, 36), dtype=np.float32))
    c4 = torch.from_numpy(rng.standard_normal((1, 16, 18, 18), dtype=np.float32))
    mod3 = torch.from_numpy(rng.standard_normal((1, 16), dtype=np.float32))
    mod4 = torch.from_numpy(rng.standard_normal((1, 16), dtype=np.float32))
    props = torch.from_numpy(np.concatenate((rng.uniform(60, 140, (10, 2)), rng.uniform(40, 120, (10, 2))), 1)
                             .astype(np.float32))[None]
    props.requires_grad_(True)
    iou = net.predict_iou((mod3, mod4), (c3, c4), props)
    iou.backward(gradient=torch.ones_like(iou))
    sd = {k: v.detach().numpy() for k, v in net.state_dict().items()
          if k.startswith(("fc3_rt", "fc4_rt", "iou_predictor"))}
    save("iou_predict", c3=c3.numpy(), c4=c4.numpy(), mod3=mod3.numpy(), mod4=mod4.numpy(),
         proposals=props.detach().numpy(), iou=iou.detach().numpy(), grad=props.grad.numpy(),
         **{"sd_" + k.replace(".", "__"): v for k, v in sd.items()})


def gen_lwl():
    """LWL few-shot learner: the reference's GNSteepestDescent on LWTLResidual (autograd double-backward) on CPU."""
    from pytracking import TensorList
    from ltr.models.meta.steepestdescent import GNSteepestDescent
    from ltr.models.lwl.loss_residual_modules import LWTLResidual
    rng = np.random.default_rng(41)

    def run(n, F, C, H, W, K, num_iter, sw_mode, reg, slreg, w0_zero):
        feat = synth.clf_features(rng, n, C, H, W, K)
        label = rng.uniform(0.0, 1.0, (n, F, H, W)).astype(np.float32)
        if sw_mode == "full":
            sw = rng.uniform(0.2, 1.0, (n, F, H, W)).astype(np.float32)
        elif sw_mode == "per_image":
            sw = rng.uniform(0.2, 1.0, (n,)).astype(np.float32)
        else:
            sw = None
        w0 = np.zeros((F, C, K, K), np.float32) if w0_zero else \
            (rng.standard_normal((F, C, K, K), dtype=np.float32) * np.float32(0.02))
        res = LWTLResidual(init_filter_reg=reg)
        opt = GNSteepestDescent(residual_module=res, num_iter=num_iter, compute_losses=True, steplength_reg=slreg,
                                residual_batch_dim=1)          # as ltr/models/lwl/lwl_net.py:192-194 constructs it
        w, its, losses = opt(TensorList([T(w0)[None]]), feat=T(feat)[:, None], label=T(label)[:, None],
                             sample_weight=None if sw is None else (T(sw)[:, None] if sw.ndim == 4 else T(sw)))
        with torch.no_grad():
            scores = rfilter.apply_filter(T(feat)[:, None], w[0].detach())
        return dict(feat=feat, label=label, w0=w0, sw=np.zeros(0, np.float32) if sw is None else sw,
                    iterates=torch.stack([i[0][0].detach() for i in its]).numpy(),
                    losses=torch.stack([l.detach().reshape(()) for l in losses]).numpy(),
                    scores=scores[:, 0].numpy(), num_iter=num_iter, filter_reg=reg, steplength_reg=slreg)

    save("lwl_gn_small_full", **run(3, 4, 8, 7, 9, 3, 3, "full", 0.1, 0.0, False))
    save("lwl_gn_small_img", **run(2, 16, 12, 6, 8, 3, 2, "per_image", 0.05, 0.1, True))
    save("lwl_gn_small_none", **run(4, 5, 8, 8, 6, 1, 3, "none", 0.1, 0.0, False))
    # mid size on the MFMA path geometry (C multiple of 16, 16 filters, 3x3), small enough to commit (~120 KiB)
    g = run(2, 16, 32, 10, 13, 3, 3, "full", 0.05, 0.0, True)
    save("lwl_gn_mid", **g)


def build_reference_tomp(cfg, params):
    """The reference's FilterPredictor + LinearFilterClassifier + DenseBoxRegressor (tompnet.py:106-118) carrying
    the seeded parameters, in eval mode."""
    import ltr.models.transformer.transformer as trans
    import ltr.models.transformer.filter_predictor as fp
    import ltr.models.transformer.heads as heads
    D = cfg["D"]
    tr = trans.Transformer(d_model=D, nhead=cfg["nhead"], num_encoder_layers=cfg["n_enc"],
                           num_decoder_layers=cfg["n_dec"], dim_feedforward=cfg["ff"])
    pred = fp.FilterPredictor(tr, feature_sz=cfg["feature_sz"], use_test_frame_encoding=True)
    cls = heads.LinearFilterClassifier(num_channels=D)
    reg = heads.DenseBoxRegressor(num_channels=D)
    for mod, pre in ((pred, "fp."), (cls, "cls."), (reg, "reg.")):
        sd = {k[len(pre):]: T(v.copy()) for k, v in params.items() if k.startswith(pre)}
        if pre == "fp.":
            sd["query_embed_fg_decoder.weight"] = sd["query_embed_fg.weight"]      # same module under two names (:35)
            for idx in (1, 4):
                sd[f"box_encoding.{idx}.num_batches_tracked"] = torch.tensor(0)
        mod.load_state_dict(sd, strict=True)
        mod.eval()
    return pred, cls, reg


def gen_tomp():
    """ToMP model predictor (filter_predictor.py:50-150) + heads (heads.py:83-141), reference on CPU."""
    def run(tag, cfg, seed, store_inputs):
        params = synth.tomp_params(seed, cfg)
        train, test, lab, ltrb = synth.tomp_inputs(seed + 1, cfg)
        pred, cls, reg = build_reference_tomp(cfg, params)
        with torch.no_grad():
            cw, bw, cenc, benc = pred.predict_cls_bbreg_filters_parallel(T(train), T(test), T(lab), cfg["num_gth_frames"],
                                                                         T(ltrb))
            scores = cls(cenc, cw)
            boxes = reg(benc, bw)
            w1, enc1 = pred.predict_filter(T(train), T(test), T(lab), T(ltrb))
            pos = pred.get_positional_encoding(T(test))
        out = dict(seed=seed, cls_filter=cw.numpy().reshape(-1), bbreg_filter=bw.numpy().reshape(-1),
                   cls_enc=cenc.numpy(), bbreg_enc=benc.numpy(), scores=scores.numpy(), ltrb=boxes.numpy(),
                   single_filter=w1.numpy().reshape(-1), single_enc=enc1.numpy(), pos=pos.numpy()[0, 0])
        if store_inputs:
            out.update(train=train, test=test, label=lab, ltrb_target=ltrb)
        save(f"tomp_{tag}", **out)

    run("small", synth.TOMP_SMALL, 81, True)
    run("full", synth.TOMP, 83, False)          # BASELINE configs[3] geometry: 256 channels, 8 heads, 6+6 layers, 18x18


def gen_localize():
    """`DiMP.localize_advanced` (dimp.py:238-303) of the unmodified reference on CPU, called unbound on a stand-in object
    that carries exactly the attributes the method reads."""
    import types
    from pytracking.tracker.dimp.dimp import DiMP
    from pytracking.utils import TrackerParams
    rng = np.random.default_rng(131)
    cases = []
    for i in range(48):
        S = 1 if i % 3 else 2
        H = W = 19 if i % 2 else 23
        K = 4 if i % 2 else 5
        params = TrackerParams()
        params.target_not_found_threshold = 0.25
        params.distractor_threshold = 0.8
        params.hard_negative_threshold = 0.5
        params.target_neighborhood_scale = 2.2
        params.dispalcement_scale = 0.8
        if i % 7 == 3:
            params.uncertain_threshold = 0.45
        if i % 11 == 5:
            params.hard_sample_threshold = 0.5
        base = rng.standard_normal((S, H, W)).astype(np.float32) * np.float32(0.05)
        peak = float(rng.choice([0.15, 0.4, 0.7, 1.0]))
        yy, xx = np.mgrid[0:H, 0:W]
        r1, c1 = rng.integers(3, H - 3, 2)
        base[rng.integers(0, S)] += (peak * np.exp(-((yy - r1) ** 2 + (xx - c1) ** 2) / 3.0)).astype(np.float32)
        if i % 2 == 0:                                           # a distractor of comparable height somewhere else
            r2, c2 = rng.integers(2, H - 2, 2)
            base[rng.integers(0, S)] += (peak * float(rng.uniform(0.4, 1.05)) *
                                         np.exp(-((yy - r2) ** 2 + (xx - c2) ** 2) / 3.0)).astype(np.float32)
        if i % 13 == 0:                                          # exact ties: the first peak twice
            base[0, 2, 5] = base[0, 7, 5] = base[0, 7, 3] = np.float32(peak + 0.5)
        me = types.SimpleNamespace(params=params, kernel_size=torch.Tensor([K, K]), output_window=None,
                                   img_support_sz=torch.Tensor([288.0, 288.0]) * (22 / 18 if H == 23 else 1),
                                   target_sz=torch.Tensor(rng.uniform(30, 120, 2).astype(np.float32)),
                                   pos=torch.Tensor(rng.uniform(100, 200, 2).astype(np.float32)))
        sample_scales = torch.Tensor(rng.uniform(0.8, 1.3, S).astype(np.float32))
        sample_pos = me.pos.reshape(1, 2) + torch.Tensor(rng.uniform(-25, 25, (S, 2)).astype(np.float32))
        tv, scale_ind, _, flag = DiMP.localize_advanced(me, T(base.copy()), sample_pos, sample_scales)
        cases.append(dict(scores=base, K=K, img_support_sz=me.img_support_sz.numpy(), target_sz=me.target_sz.numpy(),
                          pos=me.pos.numpy(), sample_scales=sample_scales.numpy(), sample_pos=sample_pos.numpy(),
                          uncertain=params.get('uncertain_threshold', -np.inf), hard_sample=params.get('hard_sample_threshold', -np.inf),
                          tv=tv.numpy(), scale_ind=int(scale_ind), flag=flag))
    flags = sorted(set(c["flag"] for c in cases))
    print("flags covered:", {f: sum(c["flag"] == f for c in cases) for f in flags})
    out = {"n": len(cases)}
    for i, c in enumerate(cases):
        for k, v in c.items():
            out[f"c{i}_{k}"] = v
    save("localize", **out)


def gen_iou_refine():
    """IoU-guided box refinement: the reference's DiMP.optimize_boxes_default / optimize_boxes_relative (dimp.py:725-788)
    driving the reference's AtomIoUNet.predict_iou (atom_iou_net.py:96-136; PrRoIPool = the restatement) on CPU."""
    import types
    from pytracking.tracker.dimp.dimp import DiMP
    from pytracking.utils import TrackerParams
    from ltr.models.bbreg.atom_iou_net import AtomIoUNet
    torch.manual_seed(23)
    rng = np.random.default_rng(71)
    C, I = 64, 32
    net = AtomIoUNet(input_dim=(32, 64), pred_input_dim=(C, C), pred_inter_dim=(I, I)).eval()
    with torch.no_grad():                                      # move BN statistics / biases off their initial values
        for blk in (net.fc3_rt, net.fc4_rt):
            blk.bn.running_mean.copy_(torch.randn(I) * 0.1)
            blk.bn.running_var.copy_(torch.rand(I) + 0.5)
            blk.bn.bias.copy_(torch.randn(I) * 0.1)
            blk.linear.bias.copy_(torch.randn(I) * 0.1)
        net.iou_predictor.bias.fill_(0.3)
    c3 = T(rng.standard_normal((1, C, 36, 36), dtype=np.float32))
    c4 = T(rng.standard_normal((1, C, 18, 18), dtype=np.float32))
    mod3 = T((rng.standard_normal((1, C), dtype=np.float32) * np.float32(0.5) + 1).astype(np.float32))
    mod4 = T((rng.standard_normal((1, C), dtype=np.float32) * np.float32(0.5) + 1).astype(np.float32))
    base = np.array([100.0, 90.0, 80.0, 110.0], np.float32)
    boxes = np.stack([base] + [base + np.concatenate((rng.uniform(-12, 12, 2), rng.uniform(-25, 25, 2))).astype(np.float32)
                               for _ in range(9)])
    sd = {k: v.detach().numpy() for k, v in net.state_dict().items()
          if k.startswith(("fc3_rt", "fc4_rt", "iou_predictor")) and "num_batches" not in k}
    out = dict(c3=c3.numpy(), c4=c4.numpy(), mod3=mod3.numpy(), mod4=mod4.numpy(), boxes=boxes,
               **{"w_" + k: v for k, v in sd.items()})
    for tag, iters, step, decay, method in (("default", 5, 1.0, 1.0, DiMP.optimize_boxes_default),
                                            ("default_decay", 3, 0.8, 0.5, DiMP.optimize_boxes_default),
                                            ("relative", 10, 2.5e-3, 1.0, DiMP.optimize_boxes_relative)):
        params = TrackerParams()
        params.device = "cpu"
        params.box_refinement_iter, params.box_refinement_step_length, params.box_refinement_step_decay = iters, step, decay
        me = types.SimpleNamespace(params=params, net=types.SimpleNamespace(bb_regressor=net), iou_modulation=(mod3, mod4))
        b, iou = method(me, (c3, c4), T(boxes.copy()))
        out.update({f"{tag}_boxes": b.numpy(), f"{tag}_iou": iou.numpy(), f"{tag}_cfg": np.array([iters, step, decay])})
    from pytracking.tracker.atom.atom import ATOM
    for tag, iters, step, decay, space in (("atom_default", 6, 3.0, 0.5, "default"), ("atom_relative", 6, 6e-2, 0.5, "relative"),
                                           ("atom_nodecay", 5, 1.0, 1.0, "default")):
        params = TrackerParams()
        params.device = "cpu"
        params.box_refinement_iter, params.box_refinement_step_length, params.box_refinement_step_decay = iters, step, decay
        params.box_refinement_space = space
        me = types.SimpleNamespace(params=params, iou_predictor=net, target_feat=(mod3, mod4))
        b, iou = ATOM.optimize_boxes(me, (c3, c4), T(boxes.copy()))
        out.update({f"{tag}_boxes": b.numpy(), f"{tag}_iou": iou.numpy(), f"{tag}_cfg": np.array([iters, step, decay])})
    save("iou_refine", **out)


def gen_clf_head():
    """Classification-feature head: the reference's residual_bottleneck(num_blocks=0, final_conv=True, l2norm=True)
    (features.py:49-73) on CPU."""
    import ltr.models.target_classifier.features as rfeat
    rng = np.random.default_rng(97)
    out = {}
    for tag, n, cin, cout, H, W, scale in (("a", 2, 64, 32, 6, 7, 0.0625), ("b", 3, 128, 24, 5, 5, 1.0)):
        head = rfeat.residual_bottleneck(feature_dim=cin // 4, num_blocks=0, l2norm=True, final_conv=True,
                                         norm_scale=scale, out_dim=cout).eval()
        w = (rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) * np.float32(0.05))
        head[0].weight.data.copy_(T(w))
        x = rng.standard_normal((n, cin, H, W), dtype=np.float32)
        with torch.no_grad():
            y = head(T(x))
        out.update({f"{tag}_x": x, f"{tag}_w": w, f"{tag}_y": y.numpy(), f"{tag}_scale": scale})
    save("clf_head", **out)


# ------------------------------------------------------------------------------------------------------
# round 2: full-size / deployed-size cases and the branches that had no reference vector
# ------------------------------------------------------------------------------------------------------
def gen_lwl_full():
    """BASELINE configs[4] size: n=32 samples of 512x30x52, 16 filters 3x3, zero initial filter; 4 GN steepest-descent
    iterations (BASELINE's wording) -- the reference's per-frame setting of 3 is the prefix of the same run
    (ltr/models/meta/steepestdescent.py:32-105 is sequential in the iterate).  Inputs are regenerated from the seed."""
    from pytracking import TensorList
    from ltr.models.meta.steepestdescent import GNSteepestDescent
    from ltr.models.lwl.loss_residual_modules import LWTLResidual
    cfg = synth.LWL
    seed = 5032
    w0, feat, label, sw = synth.lwl_problem(seed, cfg)
    out = dict(seed=seed, filter_reg=cfg["filter_reg"])
    for it in (3, 4):
        res = LWTLResidual(init_filter_reg=cfg["filter_reg"])
        opt = GNSteepestDescent(residual_module=res, num_iter=it, compute_losses=True, steplength_reg=0.0,
                                residual_batch_dim=1)
        w, its, losses = opt(TensorList([T(w0)[None]]), feat=T(feat)[:, None], label=T(label)[:, None],
                             sample_weight=T(sw)[:, None])
        out[f"losses{it}"] = torch.stack([l.detach().reshape(()) for l in losses]).numpy()
        out[f"final{it}"] = w[0].detach()[0].numpy()
        if it == 4:
            with torch.no_grad():
                s = rfilter.apply_filter(T(feat[:2])[:, None], w[0].detach())
            out["scores_first2"] = s[:, 0].numpy()
            out["iterate1"] = its[1][0][0].detach().numpy()
    save("lwl_gn_cfg5_n32", **out)


def gen_long_runs():
    """Iteration counts the trackers really use beyond the per-frame setting:
    LWL's first-frame optimisation `net_opt_iter = 20` (pytracking/parameter/lwl/lwl_ytvos.py:30) at the BASELINE
    configs[4] geometry with n = 1 (the first frame of every sequence) and n = 32, and DiMP at 20 iterations
    (the C ABI allows 64) on the configs[1] geometry, n = 15.  What a fused solver that carries scores by recurrence
    could get wrong is drift: the final filter, the losses of all 21 iterates and the scores under the final filter
    are stored.  Inputs are regenerated from the seed by the tests."""
    from pytracking import TensorList
    from ltr.models.meta.steepestdescent import GNSteepestDescent
    from ltr.models.lwl.loss_residual_modules import LWTLResidual
    for n, seed in ((1, 5101), (32, 5132)):
        cfg = dict(synth.LWL, n=n)
        w0, feat, label, sw = synth.lwl_problem(seed, cfg)
        res = LWTLResidual(init_filter_reg=cfg["filter_reg"])
        opt = GNSteepestDescent(residual_module=res, num_iter=20, compute_losses=True, steplength_reg=0.0,
                                residual_batch_dim=1)
        w, its, losses = opt(TensorList([T(w0)[None]]), feat=T(feat)[:, None], label=T(label)[:, None],
                             sample_weight=T(sw)[:, None])
        with torch.no_grad():
            s = rfilter.apply_filter(T(feat[:1])[:, None], w[0].detach())
        save(f"lwl_gn_cfg5_n{n}_it20", seed=seed, n=n, filter_reg=cfg["filter_reg"], num_iter=20,
             losses=torch.stack([l.detach().reshape(()) for l in losses]).numpy(), final=w[0].detach()[0].numpy(),
             iterate10=its[10][0][0].detach().numpy().astype(np.float16),      # coarse mid-run check, half the bytes
             scores_first=s[:, 0].numpy())
    cfg = synth.DIMP50
    w0, feat, bb, sw = synth.dimp_problem(1235, 15, cfg)
    its, losses, scores = _run_opt(_dimp_module(cfg), w0, feat, bb, sw, 20)
    save("dimp_sd_cfg2_n15_it20", seed=1235, n=15, num_iter=20, losses=losses, scores=scores,
         iterates=its[[5, 10, 20]], which=np.array([5, 10, 20]))


def _iou_net(cfg, params):
    from ltr.models.bbreg.atom_iou_net import AtomIoUNet
    C, I = cfg["C"], cfg["I"]
    net = AtomIoUNet(input_dim=(32, 64), pred_input_dim=(C, C), pred_inter_dim=(I, I)).eval()
    sd = net.state_dict()
    for k, v in params.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = T(v.copy())
    net.load_state_dict(sd, strict=True)
    return net


def gen_iou_refine_full():
    """IoU-guided refinement at the deployed DiMP-50 / PrDiMP-50 sizes (256-channel IoU features 36x36 / 18x18,
    256-wide LinearBlocks, 10 proposals): the reference's optimize_boxes_default (5 it), optimize_boxes_relative
    (10 it) and ATOM.optimize_boxes on CPU.  Weights / inputs are regenerated from the seeds by the tests."""
    import types
    from pytracking.tracker.dimp.dimp import DiMP
    from pytracking.tracker.atom.atom import ATOM
    from pytracking.utils import TrackerParams
    cfg = synth.IOU50
    pseed, iseed = 7301, 7302
    net = _iou_net(cfg, synth.iou_net_params(pseed, cfg))
    c3, c4, mod3, mod4, boxes = synth.iou_inputs(iseed, cfg)
    c3, c4, mod3, mod4 = T(c3), T(c4), T(mod3), T(mod4)
    out = dict(param_seed=pseed, input_seed=iseed)
    for tag, iters, step, decay, method in (("default", 5, 1.0, 1.0, DiMP.optimize_boxes_default),
                                            ("relative", 10, 2.5e-3, 1.0, DiMP.optimize_boxes_relative)):
        params = TrackerParams()
        params.device = "cpu"
        params.box_refinement_iter, params.box_refinement_step_length, params.box_refinement_step_decay = iters, step, decay
        me = types.SimpleNamespace(params=params, net=types.SimpleNamespace(bb_regressor=net), iou_modulation=(mod3, mod4))
        b, iou = method(me, (c3, c4), T(boxes.copy()))
        out.update({f"{tag}_boxes": b.numpy(), f"{tag}_iou": iou.numpy(), f"{tag}_cfg": np.array([iters, step, decay])})
    params = TrackerParams()
    params.device = "cpu"
    params.box_refinement_iter, params.box_refinement_step_length, params.box_refinement_step_decay = 5, 1.0, 1.0
    params.box_refinement_space = "default"
    me = types.SimpleNamespace(params=params, iou_predictor=net, target_feat=(mod3, mod4))
    b, iou = ATOM.optimize_boxes(me, (c3, c4), T(boxes.copy()))
    out.update(atom_boxes=b.numpy(), atom_iou=iou.numpy(), atom_cfg=np.array([5, 1.0, 1.0]))
    save("iou_refine_full", **out)


def gen_atom_gn_full():
    """ATOM first frame at the deployed schedule (parameter/atom/default.py:27-28: init_CG_iter 60 / init_GN_iter 6
    -> six Gauss-Newton iterations of ten CG iterations; Polak-Ribiere, default.py:30) on 30 samples of 256x18x18,
    64 compressed channels, 4x4 filter.  Reference: GaussNewtonCG on FactorizedConvProblem, CPU autograd."""
    from pytracking import TensorList
    from pytracking.libs import optimization
    from pytracking.tracker.atom.optim import FactorizedConvProblem
    from ltr.models.layers import activation
    cfg = synth.ATOM18
    seed = 9030
    f0, P0, samples, y, sw = synth.atom_gn_problem(seed)
    act = activation.MLU(cfg["act_min_val"])
    out = dict(seed=seed, filter_reg=cfg["filter_reg"], projection_reg=1e-4, act_min_val=cfg["act_min_val"])
    for tag, fr in (("pr", False), ("fr", True)):
        prob = FactorizedConvProblem(TensorList([T(samples)]), TensorList([T(y)[:, None]]), TensorList([cfg["filter_reg"]]),
                                     TensorList([1e-4]), None, TensorList([T(sw)]), lambda x: x, act)
        filt = T(f0.copy())[None].clone()
        proj = T(P0.copy())[:, :, None, None].clone()
        var = TensorList([filt]).concat(TensorList([proj]))
        opt = optimization.GaussNewtonCG(prob, var, fletcher_reeves=fr)
        opt.run([10] * 6)
        out[f"f_out_{tag}"] = var[0].detach()[0].numpy()
        out[f"P_out_{tag}"] = var[1].detach()[:, :, 0, 0].numpy()
    save("atom_gn_first_frame", **out)


def gen_branches():
    """Reference vectors for branches no earlier golden touched: score_act='bentpar', mask_act='linear',
    PrDiMP gauss_sigma=0, two sequences through the optimiser modules, CG direction forgetting across run() calls."""
    cfg = synth.DIMP50
    small = dict(C=16, H=10, W=10)
    # DiMP with BentIdentPar score activation and a linear target mask (activation.py:47-66, optimizer.py:57-66,75-77)
    for tag, over in (("bentpar", dict(score_act="bentpar", act_param=2.0)), ("linmask", dict(mask_act="linear")),
                      ("bentpar_linmask", dict(score_act="bentpar", act_param=0.7, mask_act="linear"))):
        c = dict(cfg, **over)
        w0, feat, bb, sw = synth.dimp_problem(301, 4, cfg, small=small)
        m = roptim.DiMPSteepestDescentGN(
            num_iter=3, feat_stride=c["feat_stride"], init_step_length=c["init_step_length"],
            init_filter_reg=c["init_filter_reg"], init_gauss_sigma=c["init_gauss_sigma"], num_dist_bins=c["num_dist_bins"],
            bin_displacement=c["bin_displacement"], mask_init_factor=c["mask_init_factor"], score_act=c["score_act"],
            act_param=c.get("act_param"), mask_act=c["mask_act"], min_filter_reg=c["min_filter_reg"],
            alpha_eps=c["alpha_eps"]).eval()
        its, losses, scores = _run_opt(m, w0, feat, bb, sw, 3)
        save(f"dimp_sd_{tag}", w0=w0, feat=feat, bb=bb, sw=sw, iterates=its, losses=losses, scores=scores, num_iter=3,
             score_act=c["score_act"], act_param=c.get("act_param") or 0.0, mask_act=c["mask_act"])
    # the same branches on the XCD-aligned path geometry (C=128, 18x18): seed + outputs only
    for tag, over in (("bentpar_c128", dict(score_act="bentpar", act_param=2.0, mask_act="linear")),):
        c = dict(cfg, **over)
        w0, feat, bb, sw = synth.dimp_problem(302, 6, cfg, small=dict(C=128, H=18, W=18))
        m = roptim.DiMPSteepestDescentGN(
            num_iter=3, feat_stride=c["feat_stride"], init_step_length=c["init_step_length"],
            init_filter_reg=c["init_filter_reg"], init_gauss_sigma=c["init_gauss_sigma"], num_dist_bins=c["num_dist_bins"],
            bin_displacement=c["bin_displacement"], mask_init_factor=c["mask_init_factor"], score_act=c["score_act"],
            act_param=c.get("act_param"), mask_act=c["mask_act"], min_filter_reg=c["min_filter_reg"],
            alpha_eps=c["alpha_eps"]).eval()
        its, losses, scores = _run_opt(m, w0, feat, bb, sw, 3)
        save(f"dimp_sd_{tag}", seed=302, n=6, iterates=its, losses=losses, num_iter=3, act_param=2.0)
    # PrDiMP with gauss_sigma = 0 (one-hot label at the nearest cell, optimizer.py:334-341)
    pc = dict(synth.PRDIMP50, gauss_sigma=0.0)
    w0, feat, bb, sw = synth.dimp_problem(303, 4, synth.PRDIMP50, small=small)
    its, losses, scores = _run_opt(_prdimp_module(pc), w0 * 0, feat, bb, sw, 3)
    save("prdimp_sd_sigma0", w0=w0 * 0, feat=feat, bb=bb, sw=sw, iterates=its, losses=losses, scores=scores, num_iter=3)
    w0, feat, bb, sw = synth.dimp_problem(304, 5, synth.PRDIMP50, small=dict(C=128, H=18, W=18))
    its, losses, scores = _run_opt(_prdimp_module(pc), w0 * 0, feat, bb, sw, 3)
    save("prdimp_sd_sigma0_c128", seed=304, n=5, iterates=its, losses=losses, num_iter=3)
    # two sequences in one call: feat (n,S,C,H,W), weights (S,C,K,K), bb (n,S,4), sample_weight (n,S)
    rng = np.random.default_rng(305)
    S, n = 2, 4
    probs = [synth.dimp_problem(306 + s, n, cfg, small=small) for s in range(S)]
    w0 = np.stack([p[0] for p in probs])
    feat = np.stack([p[1] for p in probs], axis=1)
    bb = np.stack([p[2] for p in probs], axis=1)
    sw = np.stack([rng.uniform(0.1, 1.0, n).astype(np.float32) for _ in range(S)], axis=1)
    for tag, mod in (("dimp", _dimp_module(cfg)), ("prdimp", _prdimp_module(synth.PRDIMP50))):
        with torch.no_grad():
            w, its, losses = mod(T(w0), T(feat), T(bb), sample_weight=T(sw), num_iter=3, compute_losses=True)
        save(f"{tag}_sd_two_sequences", w0=w0, feat=feat, bb=bb, sw=sw,
             iterates=torch.stack(its).numpy(), losses=torch.stack([l.reshape(-1) for l in losses]).numpy(), num_iter=3)
    # ATOM CG with direction forgetting: state (p, rho, r_prev) carried across three run() calls
    # (optimization.py:82-85; atom.py:85-89 gives (1 - lr)^CG_forgetting_rate)
    from pytracking import TensorList
    from pytracking.libs import optimization
    from pytracking.tracker.atom.optim import ConvProblem
    from ltr.models.layers import activation
    acfg = synth.ATOM18
    for tag, fr, forget in (("pr", False, 0.75), ("fr", True, 0.5)):
        x0, samples, y, sw1 = synth.atom_problem(311, 6, acfg, small=dict(C=8, H=10, W=10))
        prob = ConvProblem(TensorList([T(samples)]), TensorList([T(y)[:, None]]), TensorList([acfg["filter_reg"]]),
                           TensorList([T(sw1)]), activation.MLU(acfg["act_min_val"]))
        x = TensorList([T(x0.copy())[None].clone()])
        opt = optimization.ConjugateGradient(prob, x, fletcher_reeves=fr, direction_forget_factor=forget)
        outs = []
        for iters in (3, 2, 3):
            opt.run(iters)
            outs.append(x[0].detach()[0].numpy().copy())
        save(f"atom_cg_forget_{tag}", x0=x0, samples=samples, y=y, sw=sw1, x_out=np.stack(outs), iters=np.array([3, 2, 3]),
             fletcher_reeves=int(fr), forget=forget)


def gen_atom_cg_c64():
    """ATOM's compressed-channel shape (C = 64, 4x4 filter) at a small map: the shape class the fused CG path of
    csrc/atom_cg.hip serves.  Polak-Ribiere and Fletcher-Reeves, with and without direction forgetting, the CG state carried
    over three run() calls (optimization.py:72-163, 227-289).  Inputs are regenerated from the seed by the tests."""
    from pytracking import TensorList
    from pytracking.libs import optimization
    from pytracking.tracker.atom.optim import ConvProblem
    from ltr.models.layers import activation
    acfg = synth.ATOM18
    small = dict(C=64, H=12, W=12)
    out = dict(seed=321, n=12, H=12, W=12, iters=np.array([3, 2, 3]))
    for tag, fr, forget in (("pr", False, 0.75), ("fr", True, 0.5), ("pr0", False, 0.0), ("fr0", True, 0.0)):
        x0, samples, y, sw1 = synth.atom_problem(321, 12, acfg, small=small)
        prob = ConvProblem(TensorList([T(samples)]), TensorList([T(y)[:, None]]), TensorList([acfg["filter_reg"]]),
                           TensorList([T(sw1)]), activation.MLU(acfg["act_min_val"]))
        x = TensorList([T(x0.copy())[None].clone()])
        opt = optimization.ConjugateGradient(prob, x, fletcher_reeves=fr, direction_forget_factor=forget)
        outs = []
        for iters in (3, 2, 3):
            opt.run(iters)
            outs.append(x[0].detach()[0].numpy().copy())
        out[f"{tag}_x_out"] = np.stack(outs)
        out[f"{tag}_fr"] = int(fr)
        out[f"{tag}_forget"] = forget
    save("atom_cg_c64", **out)


def gen_patches():
    """`sample_patch` / `sample_patch_multiscale` of the unmodified reference (preprocessing.py:33-148) on synthetic images:
    every border mode, pre-downsampling factors 1..3, crops hanging over each border, up- and down-sampling."""
    from pytracking.features.preprocessing import sample_patch, sample_patch_multiscale
    rng = np.random.default_rng(151)
    im = T(rng.integers(0, 256, size=(1, 3, 70, 90)).astype(np.float32))
    out = {"im": im.numpy()}
    cases = []
    k = 0
    for mode, msc in (("replicate", None), ("inside", None), ("inside_major", 1.5), ("inside", 1.2)):
        for pos, ssz, osz in (((35.3, 44.8), (40.0, 40.0), (32, 32)), ((3.0, 5.0), (60.0, 50.0), (24, 24)),
                              ((66.7, 88.2), (100.0, 120.0), (32, 32)), ((30.0, 40.0), (14.0, 14.0), (32, 32)),
                              ((20.5, 70.0), (200.0, 260.0), (24, 32)), ((34.0, 45.0), (97.0, 131.0), (32, 32))):
            p, c = sample_patch(im, torch.Tensor(pos), torch.Tensor(ssz), torch.Tensor(osz), mode=mode, max_scale_change=msc)
            out.update({f"c{k}_pos": np.array(pos, np.float32), f"c{k}_ssz": np.array(ssz, np.float32),
                        f"c{k}_osz": np.array(osz, np.int64), f"c{k}_mode": mode, f"c{k}_msc": -1.0 if msc is None else msc,
                        f"c{k}_patch": p.numpy(), f"c{k}_coord": c.numpy()})
            k += 1
    out["n"] = k
    ps, cs = sample_patch_multiscale(im, torch.Tensor([33.0, 47.0]), torch.Tensor([0.8, 1.0, 1.9]), torch.Tensor([32.0, 32.0]))
    out.update(ms_pos=np.array([33.0, 47.0], np.float32), ms_scales=np.array([0.8, 1.0, 1.9], np.float32), ms_patches=ps.numpy(),
               ms_coords=cs.numpy())
    save("sample_patch", **out)


def gen_augment():
    """`sample_patch_transformed` of the unmodified reference (preprocessing.py:13-30) with transform lists built from the
    reference's own classes (augmentation.py): Identity, Translation, FlipHorizontal, FlipVertical, Blur, Scale -- with and without
    an output size, shifts that crop and shifts that pad.  `Rotate` needs cv2 (not installed): no golden, declared unpinned.
    Case `s`: small, full outputs.  Case `d`: the DiMP-50 first-frame list (parameter/dimp/dimp50.py:38-44 without 'rotate';
    288 output, expansion factor 2, random shifts drawn as dimp.py:363-366 does) -- outputs stored on a stride-5 pixel grid."""
    from pytracking.features.preprocessing import sample_patch_transformed
    from pytracking.features import augmentation as A
    rng = np.random.default_rng(171)

    def describe(Tr):
        d = {"cls": type(Tr).__name__, "output_sz": -1 if Tr.output_sz is None else np.array(Tr.output_sz, np.int64),
             "shift": np.array(Tr.shift, np.int64)}
        if isinstance(Tr, A.Blur):
            d["f0"], d["f1"] = Tr.filter[0].reshape(-1).numpy(), Tr.filter[1].reshape(-1).numpy()
        if isinstance(Tr, A.Scale):
            d["scale_factor"] = float(Tr.scale_factor)
        return d

    out = {}

    def run(tag, im, pos, scale, image_sz, transforms, stride):
        res = sample_patch_transformed(im, torch.Tensor(pos), scale, torch.Tensor(image_sz), transforms)
        out.update({f"{tag}_im": im.numpy().astype(np.uint8), f"{tag}_pos": np.array(pos, np.float32), f"{tag}_scale": np.float32(scale),
                    f"{tag}_image_sz": np.array(image_sz, np.float32), f"{tag}_n": len(transforms), f"{tag}_stride": stride,
                    f"{tag}_out": res.numpy()[:, :, ::stride, ::stride], f"{tag}_shape": np.array(res.shape, np.int64),
                    f"{tag}_sums": res.double().sum(dim=(1, 2, 3)).numpy()})
        for k, Tr in enumerate(transforms):
            for key, v in describe(Tr).items():
                out[f"{tag}_t{k}_{key}"] = v

    im = T(rng.integers(0, 256, size=(1, 3, 70, 90)).astype(np.float32))
    osz = [32, 32]
    trs = [A.Identity(osz, [0, 0]), A.Identity(osz, [3, -2]), A.Translation([5, -7], osz, [1, 1]), A.Translation([-20, 24], osz, [0, 0]),
           A.FlipHorizontal(osz, [2, 3]), A.FlipVertical(osz, [-4, 0]), A.Blur((3, 1), osz, [0, 5]), A.Blur((1, 3), osz, None),
           A.Blur(2, osz, [-6, -6]), A.Blur((0.2, 1.5), osz, [1, 0]), A.Scale(0.8, osz, [2, -3]), A.Scale(1.3, osz, None),
           A.Scale(2.5, osz, [4, 4])]
    run("s", im, (35.3, 44.8), 1.1, (64.0, 64.0), trs, 1)
    trs = [A.Identity(None, None), A.Translation([9, -4]), A.FlipHorizontal(None, [3, 3]), A.Blur((1, 2), None, [-5, 2])]
    run("n", im, (30.0, 50.0), 0.9, (48.0, 40.0), trs, 1)

    # DiMP-50 first frame (dimp.py:329-395): image_sample_size 288, augmentation_expansion_factor 2, random_shift_factor 1/3
    im = T(rng.integers(0, 256, size=(1, 3, 240, 320)).astype(np.float32))
    img_sample_sz = torch.Tensor([288.0, 288.0])
    aug_expansion_sz = (img_sample_sz * 2).long()
    aug_expansion_sz += (aug_expansion_sz - img_sample_sz.long()) % 2
    aug_expansion_sz = aug_expansion_sz.float()
    aug_output_sz = img_sample_sz.long().tolist()
    torch.manual_seed(5)
    global_shift = torch.zeros(2)
    rand_shift = lambda: ((torch.rand(2) - 0.5) * img_sample_sz * (1 / 3) + global_shift).long().tolist()
    get_absolute = lambda shift: (torch.Tensor(shift) * img_sample_sz / 2).long().tolist()
    trs = [A.Identity(aug_output_sz, global_shift.long().tolist())]
    trs += [A.Translation(get_absolute(sh), aug_output_sz, global_shift.long().tolist())
            for sh in [(0.6, 0.6), (-0.6, 0.6), (0.6, -0.6), (-0.6, -0.6)]]
    trs.append(A.FlipHorizontal(aug_output_sz, rand_shift()))
    trs += [A.Blur(sig, aug_output_sz, rand_shift()) for sig in [(3, 1), (1, 3), (2, 2)]]
    run("d", im, (118.0, 163.0), 0.7, aug_expansion_sz.tolist(), trs, 5)
    save("augment", **out)


def _lwl_compact(d):
    """The decoder's 480 x 832 score maps are stock-PyTorch outputs and 1.6 MB each: stored as float16 (they are compared at 2e-3);
    everything on the hot path (filters, mask encodings) stays float32."""
    return {k: (np.asarray(v).astype(np.float16) if k.endswith("_scores") else v) for k, v in d.items()}


def gen_trackers():
    """Trajectory-level vectors: the unmodified reference DiMP tracker (initialize + 10 x track) on a stubbed backbone,
    every boundary call recorded (oracle/tracker_harness.py)."""
    from oracle import tracker_harness as TH
    outs, rec, _ = TH.run_dimp(**TH.DIMP_RUN)
    save("tracker_dimp50", **rec.to_npz_dict())
    outs, rec, _ = TH.run_tomp(**TH.TOMP_RUN)
    save("tracker_tomp50", **rec.to_npz_dict())
    outs, rec, _ = TH.run_atom(**TH.ATOM_RUN)
    save("tracker_atom18", **rec.to_npz_dict())
    outs, rec, _ = TH.run_dimp(**TH.PRDIMP_RUN)
    save("tracker_prdimp50", **rec.to_npz_dict())
    outs, rec, _ = TH.run_lwl(**TH.LWL_RUN)
    save("tracker_lwl", **_lwl_compact(rec.to_npz_dict()))


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["filter", "dimp", "l2", "prdimp", "atom", "prroi", "lwl", "atomgn", "tomp", "head", "localize", "iou",
                             "branches", "ioufull", "atomgnfull", "lwlfull", "trackers", "patches", "long", "augment", "atomc64"]
    if "tomp" in which:
        gen_tomp()
    if "head" in which:
        gen_clf_head()
    if "localize" in which:
        gen_localize()
    if "iou" in which:
        gen_iou_refine()
    if "atomgn" in which:
        gen_atom_gn()
    if "lwl" in which:
        gen_lwl()
    if "filter" in which:
        gen_filter_ops()
    if "dimp" in which:
        gen_dimp()
    if "l2" in which:
        gen_dimp_l2()
    if "prdimp" in which:
        gen_prdimp()
    if "atom" in which:
        gen_atom_cg()
    if "prroi" in which:
        gen_prroi_consumers()
    if "branches" in which:
        gen_branches()
    if "ioufull" in which:
        gen_iou_refine_full()
    if "atomgnfull" in which:
        gen_atom_gn_full()
    if "lwlfull" in which:
        gen_lwl_full()
    if "trackers" in which:
        gen_trackers()
    if "patches" in which:
        gen_patches()
    if "long" in which:
        gen_long_runs()
    if "augment" in which:
        gen_augment()
    if "atomc64" in which:
        gen_atom_cg_c64()
