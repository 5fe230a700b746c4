"""The reference-side binding (pytracking_amd/install.py) against the real reference tree.  Runs only where
/root/reference exists (the build container); the GPU box has no reference and skips."""
import pytest
import torch

from oracle import ref_harness

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason="reference tree not mounted")


def test_install_rebinds_boundary_symbols_and_restores():
    ref_harness.install()
    import sys
    saved_prroi = sys.modules.get("ltr.external.PreciseRoIPooling.pytorch.prroi_pool")
    import ltr.models.layers.filter as fl
    import ltr.models.target_classifier.optimizer as opt
    orig_apply, orig_cls = fl.apply_filter, opt.DiMPSteepestDescentGN
    from pytracking_amd import install as amd, optimizer as ours, prroi_pool
    amd.install()
    try:
        assert issubclass(opt.DiMPSteepestDescentGN, ours.DiMPSteepestDescentGN)
        assert issubclass(opt.PrDiMPSteepestDescentNewton, ours.PrDiMPSteepestDescentNewton)
        assert opt.DiMPSteepestDescentGN.__name__ == "DiMPSteepestDescentGN"
        assert fl.apply_filter is not orig_apply and fl.apply_filter.__wrapped__ is orig_apply
        from ltr.external.PreciseRoIPooling.pytorch.prroi_pool import PrRoIPool2D
        assert issubclass(PrRoIPool2D, prroi_pool.PrRoIPool2D)
        # CPU tensors are outside the hot path: the dispatcher hands them to the reference's own function
        feat, filt = torch.randn(2, 1, 8, 6, 6), torch.randn(1, 8, 4, 4)
        torch.testing.assert_close(fl.apply_filter(feat, filt), orig_apply(feat, filt))
        # a network built now instantiates the mirrored optimiser and loads a reference state_dict
        ref_mod = orig_cls(num_iter=5, feat_stride=16, init_step_length=0.9, init_filter_reg=0.1, init_gauss_sigma=0.9,
                           num_dist_bins=100, bin_displacement=0.1, mask_init_factor=3.0)
        mine = opt.DiMPSteepestDescentGN(num_iter=5, feat_stride=16, init_step_length=1.0, init_filter_reg=0.3,
                                         init_gauss_sigma=0.5, num_dist_bins=100, bin_displacement=0.1,
                                         mask_init_factor=4.0)
        mine.load_state_dict(ref_mod.state_dict(), strict=True)
        torch.testing.assert_close(mine.label_map_predictor.weight, ref_mod.label_map_predictor.weight)
        import pytracking.libs.optimization as po
        assert po.ConjugateGradient is not amd._state["originals"]["cg"]
    finally:
        amd.uninstall()
        if saved_prroi is not None:
            sys.modules["ltr.external.PreciseRoIPooling.pytorch.prroi_pool"] = saved_prroi
    assert fl.apply_filter is orig_apply and opt.DiMPSteepestDescentGN is orig_cls


def test_tomp_mirror_state_dict_matches_reference_and_install_dispatch():
    """The mirrored ToMP modules expose exactly the reference's state_dict (names and shapes), so a reference checkpoint
    loads strict; after install() the tompnet constructors build the mirrors for the covered configuration and the
    reference classes otherwise."""
    ref_harness.install()
    import ltr.models.transformer.transformer as rt
    import ltr.models.transformer.filter_predictor as rf
    import ltr.models.transformer.heads as rh
    from pytracking_amd import install as amd, transformer as TM
    kw = dict(d_model=128, nhead=4, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=256)
    ref_pred = rf.FilterPredictor(rt.Transformer(**kw), feature_sz=6)
    mine = TM.FilterPredictor(TM.Transformer(**kw), feature_sz=6)
    for a, b in ((ref_pred, mine), (rh.LinearFilterClassifier(128), TM.LinearFilterClassifier(128)),
                 (rh.DenseBoxRegressor(128), TM.DenseBoxRegressor(128))):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        assert all(sa[k].shape == sb[k].shape for k in sa)
        b.load_state_dict(sa, strict=True)
    orig_tr = rt.Transformer
    amd.install()
    try:
        t = rt.Transformer(**kw)
        assert isinstance(t, TM.Transformer)
        assert isinstance(rf.FilterPredictor(t, feature_sz=6), TM.FilterPredictor)
        assert rh.LinearFilterClassifier is TM.LinearFilterClassifier
        assert isinstance(rh.DenseBoxRegressor(128), TM.DenseBoxRegressor)
        pre = rt.Transformer(normalize_before=True, **kw)                 # outside the hot path: the reference's class
        assert isinstance(pre, orig_tr) and not isinstance(pre, TM.Transformer)
        assert not isinstance(rf.FilterPredictor(pre, feature_sz=6), TM.FilterPredictor)
    finally:
        amd.uninstall()
    assert rt.Transformer is orig_tr


def test_clf_head_install_dispatch():
    """After install() `residual_bottleneck` builds the fused head around the reference's own layers for the trackers'
    configuration (same state_dict, CPU call = the reference's nn.Sequential), the reference module otherwise."""
    ref_harness.install()
    import ltr.models.target_classifier.features as rfeat
    from pytracking_amd import install as amd, features as FM
    ref_fn = rfeat.residual_bottleneck
    ref = ref_fn(feature_dim=16, num_blocks=0, l2norm=True, final_conv=True, norm_scale=0.5, out_dim=8)
    amd.install()
    try:
        head = rfeat.residual_bottleneck(feature_dim=16, num_blocks=0, l2norm=True, final_conv=True, norm_scale=0.5, out_dim=8)
        assert isinstance(head, FM.ClfHead)
        assert list(head.state_dict().keys()) == list(ref.state_dict().keys()) == ["0.weight"]
        head.load_state_dict(ref.state_dict(), strict=True)
        x = torch.randn(2, 64, 5, 5)
        with torch.no_grad():
            torch.testing.assert_close(head.eval()(x), ref.eval()(x))                 # CPU tensor -> the reference's layers
        other = rfeat.residual_bottleneck(feature_dim=16, num_blocks=0, l2norm=False, final_conv=True, out_dim=8)
        assert not isinstance(other, FM.ClfHead)
    finally:
        amd.uninstall()
    assert rfeat.residual_bottleneck is ref_fn
    mine = FM.residual_bottleneck(feature_dim=16, num_blocks=0, l2norm=True, final_conv=True, out_dim=8)
    assert list(mine.state_dict().keys()) == ["0.weight"]
    with pytest.raises(NotImplementedError):
        FM.residual_bottleneck(feature_dim=16, num_blocks=1, final_conv=True)


def test_localization_install_dispatch():
    """install() rebinds dcf.max2d and the trackers' localize_advanced; CPU tensors still take the reference's code."""
    ref_harness.install()
    import pytracking.libs.dcf as dcf
    from pytracking.tracker.dimp.dimp import DiMP
    from pytracking.tracker.tomp.tomp import ToMP
    from pytracking_amd import install as amd
    ref_max2d, ref_dimp, ref_tomp = dcf.max2d, DiMP.localize_advanced, ToMP.localize_advanced
    amd.install()
    try:
        assert dcf.max2d.__wrapped__ is ref_max2d and DiMP.localize_advanced.__wrapped__ is ref_dimp
        assert ToMP.localize_advanced.__wrapped__ is ref_tomp
        a = torch.randn(2, 7, 9)
        v, i = dcf.max2d(a)
        rv, ri = ref_max2d(a)
        assert torch.equal(v, rv) and torch.equal(i, ri)
    finally:
        amd.uninstall()
    assert dcf.max2d is ref_max2d and DiMP.localize_advanced is ref_dimp and ToMP.localize_advanced is ref_tomp


def test_iou_refine_install_dispatch():
    ref_harness.install()
    from pytracking.tracker.dimp.dimp import DiMP
    from pytracking.tracker.atom.atom import ATOM
    from pytracking_amd import install as amd
    ref_d, ref_r, ref_a = DiMP.optimize_boxes_default, DiMP.optimize_boxes_relative, ATOM.optimize_boxes
    amd.install()
    try:
        assert DiMP.optimize_boxes_default.__wrapped__ is ref_d and DiMP.optimize_boxes_relative.__wrapped__ is ref_r
        assert ATOM.optimize_boxes.__wrapped__ is ref_a
    finally:
        amd.uninstall()
    assert DiMP.optimize_boxes_default is ref_d and DiMP.optimize_boxes_relative is ref_r and ATOM.optimize_boxes is ref_a


def test_installed_optimizers_fall_back_to_the_reference_forward_off_the_hot_path():
    """install(strict=False): CPU tensors are outside the hot path, so the rebound optimiser classes must run the
    reference's own forward on their (identically named) parameters -- checked against the reference-generated goldens;
    the per-iterate losses keep the reference's shape (1,) so that the trackers' `torch.cat(losses)` (dimp.py:583,641)
    works.  Also covers the LWL learner and ATOM's non-square-kernel gate."""
    import numpy as np
    from conftest import load_golden
    from pytracking_amd import synth
    ref_harness.install()
    import ltr.models.target_classifier.optimizer as opt
    import ltr.models.meta.steepestdescent as smod
    import ltr.models.lwl.loss_residual_modules as rmod
    import pytracking.libs.optimization as po
    from pytracking import TensorList
    from pytracking_amd import install as amd, steepestdescent as SD
    T = torch.from_numpy
    amd.install()
    try:
        g = load_golden("dimp_sd_small_w")
        c = synth.DIMP50
        mod = opt.DiMPSteepestDescentGN(
            num_iter=3, feat_stride=c["feat_stride"], init_step_length=c["init_step_length"],
            init_filter_reg=c["init_filter_reg"], init_gauss_sigma=c["init_gauss_sigma"], num_dist_bins=c["num_dist_bins"],
            bin_displacement=c["bin_displacement"], mask_init_factor=c["mask_init_factor"], score_act=c["score_act"],
            mask_act=c["mask_act"], min_filter_reg=c["min_filter_reg"], alpha_eps=c["alpha_eps"]).eval()
        with torch.no_grad():
            w, its, losses = mod(T(g["w0"])[None], T(g["feat"]), T(g["bb"]), sample_weight=T(g["sw"]), num_iter=3)
        np.testing.assert_allclose(torch.stack([i[0] for i in its]).numpy(), g["iterates"], atol=1e-6)
        assert losses[0].shape == (1,)
        np.testing.assert_allclose(torch.cat(losses).numpy(), g["losses"], rtol=1e-5)
        g = load_golden("prdimp_sd_small")
        c = synth.PRDIMP50
        mod = opt.PrDiMPSteepestDescentNewton(
            num_iter=3, feat_stride=c["feat_stride"], init_step_length=c["init_step_length"],
            init_filter_reg=c["init_filter_reg"], gauss_sigma=c["gauss_sigma"], min_filter_reg=c["min_filter_reg"],
            alpha_eps=c["alpha_eps"], normalize_label=c["normalize_label"]).eval()
        with torch.no_grad():
            w, its, losses = mod(T(g["w0"])[None], T(g["feat"]), T(g["bb"]), sample_weight=T(g["sw"]), num_iter=3)
        np.testing.assert_allclose(torch.stack([i[0] for i in its]).numpy(), g["iterates"], atol=1e-6)
        # LWL: the rebound residual module / optimiser on CPU tensors = the reference's autograd formulation
        g = load_golden("lwl_gn_small_full")
        res = rmod.LWTLResidual(init_filter_reg=float(g["filter_reg"]))
        assert isinstance(res, SD.LWTLResidual)
        o = smod.GNSteepestDescent(residual_module=res, num_iter=int(g["num_iter"]), compute_losses=True,
                                   steplength_reg=float(g["steplength_reg"]), residual_batch_dim=1)
        assert isinstance(o, SD.GNSteepestDescent)
        w, its, losses = o(TensorList([T(g["w0"])[None]]), feat=T(g["feat"])[:, None], label=T(g["label"])[:, None],
                           sample_weight=T(g["sw"])[:, None])
        np.testing.assert_allclose(w[0].detach()[0].numpy(), g["iterates"][-1], atol=1e-6)
        dil = rmod.LWTLResidual(init_filter_reg=0.1, filter_dilation_factors=[1, 2])      # dilated: the reference class
        assert not isinstance(dil, SD.LWTLResidual)
        assert not isinstance(smod.GNSteepestDescent(residual_module=dil, residual_batch_dim=1), SD.GNSteepestDescent)
        # ATOM: a non-square kernel is not covered -> the reference ConjugateGradient object
        from pytracking.tracker.atom.optim import ConvProblem
        from ltr.models.layers import activation
        prob = ConvProblem(TensorList([torch.zeros(2, 4, 6, 6)]), TensorList([torch.zeros(2, 1, 6, 6)]), TensorList([0.1]),
                           TensorList([torch.ones(2)]), activation.MLU(0.05))
        cg = po.ConjugateGradient(prob, TensorList([torch.zeros(1, 4, 2, 4)]), fletcher_reeves=False)
        from pytracking_amd import optimization as OM
        assert not isinstance(cg, OM.ConjugateGradient) and isinstance(cg, amd._state["originals"]["cg"])
        # TensorList with more than one feature block (multi-resolution ATOM): the fused path takes single-block
        # problems only, so such a variable gets the reference class
        prob2 = ConvProblem(TensorList([torch.zeros(2, 4, 6, 6)] * 2), TensorList([torch.zeros(2, 1, 6, 6)] * 2),
                            TensorList([0.1, 0.1]), TensorList([torch.ones(2)] * 2), activation.MLU(0.05))
        cg2 = po.ConjugateGradient(prob2, TensorList([torch.zeros(1, 4, 4, 4), torch.zeros(1, 4, 4, 4)]), fletcher_reeves=False)
        assert not isinstance(cg2, OM.ConjugateGradient)
    finally:
        amd.uninstall()


def test_operation_conv2d_rebinding():
    """`pytracking.libs.operation.conv2d` stays TensorList-lifted and numerically the reference's for everything off the
    hot path (CPU tensors here); the device route is checked in tests/test_gpu_parity.py."""
    ref_harness.install()
    import pytracking.libs.operation as op
    from pytracking import TensorList
    from pytracking_amd import install as amd
    ref = op.conv2d
    x, w = torch.randn(3, 8, 9, 9), torch.randn(1, 8, 4, 4)
    want_same, want_valid = ref(x, w, mode='same'), ref(x, w)
    amd.install()
    try:
        assert op.conv2d is not ref
        assert torch.equal(op.conv2d(x, w, mode='same'), want_same) and torch.equal(op.conv2d(x, w), want_valid)
        out = op.conv2d(TensorList([x, x]), TensorList([w, w]), mode='same')
        assert isinstance(out, TensorList) and torch.equal(out[1], want_same)
        assert op.conv2d(x, None) is x
    finally:
        amd.uninstall()
    assert op.conv2d is ref


def test_preprocessing_install_dispatch():
    """CPU images keep the reference's `sample_patch`; the names imported by the tracker modules are rebound and restored."""
    ref_harness.install()
    import pytracking.features.preprocessing as pp
    import pytracking.tracker.dimp.dimp as dimp_mod
    from pytracking_amd import install as amd
    from pytracking.features import augmentation as A
    ref_sp, ref_ms, ref_tr = pp.sample_patch, pp.sample_patch_multiscale, pp.sample_patch_transformed
    im = torch.rand(1, 3, 40, 50) * 255
    want, wc = ref_ms(im, torch.Tensor([20.0, 25.0]), torch.Tensor([1.0, 1.5]), torch.Tensor([16.0, 16.0]))
    trs = [A.Identity([16, 16], [0, 0]), A.Translation([3, -2], [16, 16]), A.FlipHorizontal([16, 16], [1, 1]), A.Blur((2, 1), [16, 16])]
    want_t = ref_tr(im, torch.Tensor([20.0, 25.0]), 1.2, torch.Tensor([32.0, 32.0]), trs)
    amd.install()
    try:
        assert pp.sample_patch.__wrapped__ is ref_sp and dimp_mod.sample_patch_multiscale is pp.sample_patch_multiscale
        assert dimp_mod.sample_patch_transformed is pp.sample_patch_transformed and pp.sample_patch_transformed.__wrapped__ is ref_tr
        got, gc = dimp_mod.sample_patch_multiscale(im, torch.Tensor([20.0, 25.0]), torch.Tensor([1.0, 1.5]), torch.Tensor([16.0, 16.0]))
        assert torch.equal(got, want) and torch.equal(gc, wc)
        assert torch.equal(dimp_mod.sample_patch_transformed(im, torch.Tensor([20.0, 25.0]), 1.2, torch.Tensor([32.0, 32.0]), trs), want_t)
        # the host descriptors read from the REAL reference objects (names, filter layout, shifts as the classes store them)
        from pytracking_amd import preprocessing as PP
        descs, taps, hw = PP.transform_descriptors(trs + [A.Scale(0.8, [16, 16], [2, 0]), A.Rotate(10, [16, 16])], (32, 32))
        assert hw == (16, 16) and [d.kind for d in descs] == [0, 0, 1, 3, 4, 5]
        assert (descs[1].pad_top, descs[1].pad_left) == (-8 + 3, -8 - 2) and (descs[4].th, descs[4].pad_top) == (40, -12 + 2)
        assert len(taps) == (2 * 4 + 1) + (2 * 2 + 1) and abs(sum(taps[:9]) - 1) < 1e-6
    finally:
        amd.uninstall()
    assert pp.sample_patch is ref_sp and dimp_mod.sample_patch_multiscale is ref_ms and pp.sample_patch_transformed is ref_tr
    assert dimp_mod.sample_patch_transformed is ref_tr
