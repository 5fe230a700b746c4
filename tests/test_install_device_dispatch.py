"""The device branch of every FUNCTION / CLASS `install()` rebinds that no whole-tracker run reaches (the trackers crop on the host, never
call `filter_gradient`, and none of the shipped parameter files builds `DiMPL2SteepestDescentGN`): each rebound symbol is called with device
tensors and compared with the reference's ORIGINAL object on the same inputs on the CPU; `install.stats` must show the gfx950 branch.
Needs the reference (bundle oracle/_ref on the GPU box)."""
import numpy as np
import pytest
import torch

from oracle import ref_harness

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_harness.available(), reason="no reference tree / bundle (oracle/_ref)")]


@pytest.fixture()
def installed():
    ref_harness.install()
    from pytracking_amd import install as amd
    import ltr.models.layers.filter as fl
    import ltr.models.target_classifier.optimizer as opt
    import pytracking.features.preprocessing as pp
    import pytracking.libs.dcf as dcf
    originals = dict(apply_filter=fl.apply_filter, apply_feat_transpose=fl.apply_feat_transpose, filter_gradient=fl.filter_gradient,
                     L2=opt.DiMPL2SteepestDescentGN, sample_patch=pp.sample_patch, sample_patch_multiscale=pp.sample_patch_multiscale,
                     max2d=dcf.max2d)
    amd.install()
    amd.stats.clear()
    try:
        yield amd, fl, opt, pp, dcf, originals
    finally:
        amd.uninstall()


def _close(a, b, atol, scaled=False):
    """scaled: N(0,1) operands without the networks' normalisation give sums of magnitude ~100; the bound is then relative to the largest
    reference value (float32 sums of ~1600 products in two different orders)."""
    b64 = b.detach().cpu().numpy().astype(np.float64)
    tol = atol * max(1.0, float(np.abs(b64).max())) if scaled else atol
    np.testing.assert_allclose(a.detach().cpu().numpy().astype(np.float64), b64, atol=tol, rtol=0)


def test_filter_layer_functions_on_device(installed):
    amd, fl, _, _, _, o = installed
    g = torch.Generator().manual_seed(1)
    dev = "cuda"
    # single filter, 4x4 (DiMP) and 3x3 x 16 filters (LWL)
    feat = torch.randn(5, 1, 64, 18, 18, generator=g)
    filt = torch.randn(1, 64, 4, 4, generator=g) * 0.1
    s_ref = o["apply_filter"](feat, filt)
    with torch.no_grad():
        s = fl.apply_filter(feat.to(dev), filt.to(dev))
    _close(s, s_ref, 2e-6, scaled=True)
    inp = torch.randn(*s_ref.shape, generator=g)
    for training in (False, True):
        gr_ref = o["apply_feat_transpose"](feat, inp, (4, 4), training=training)
        with torch.no_grad():
            gr = fl.apply_feat_transpose(feat.to(dev), inp.to(dev), (4, 4), training=training)
        _close(gr, gr_ref, 2e-6, scaled=True)
    fg_ref = o["filter_gradient"](feat, filt, label=inp, training=False)
    with torch.no_grad():
        fg = fl.filter_gradient(feat.to(dev), filt.to(dev), label=inp.to(dev), training=False)
    _close(fg, fg_ref, 2e-6, scaled=True)
    mf_feat = torch.randn(3, 1, 32, 10, 12, generator=g)
    mf = torch.randn(1, 16, 32, 3, 3, generator=g) * 0.1
    m_ref = o["apply_filter"](mf_feat, mf)
    with torch.no_grad():
        m = fl.apply_filter(mf_feat.to(dev), mf.to(dev))
    _close(m, m_ref, 2e-6, scaled=True)
    mi = torch.randn(*m_ref.shape, generator=g)
    a_ref = o["apply_feat_transpose"](mf_feat, mi, (3, 3), training=True)
    with torch.no_grad():
        a = fl.apply_feat_transpose(mf_feat.to(dev), mi.to(dev), (3, 3), training=True)
    _close(a, a_ref, 2e-6, scaled=True)
    assert amd.stats["apply_filter.fast"] == 2 and amd.stats["apply_feat_transpose.fast"] == 3 and amd.stats["filter_gradient.fast"] == 1
    # the only calls handed back so far are the two the reference's ORIGINAL filter_gradient made itself on its CPU tensors (it reaches
    # apply_filter / apply_feat_transpose through the rebound module attributes)
    assert {k: v for k, v in amd.stats.items() if k.endswith(".reference")} == {"apply_filter.reference": 1, "apply_feat_transpose.reference": 1}
    # outside the hot path on the device (5x5 filter: 25 taps): handed to the reference's own function, and counted as such
    f5 = torch.randn(1, 64, 5, 5, generator=g) * 0.1
    with torch.no_grad():
        s5 = fl.apply_filter(feat.to(dev), f5.to(dev))
    _close(s5, o["apply_filter"](feat, f5), 5e-6, scaled=True)
    assert amd.stats["apply_filter.reference"] == 2                  # + the 5x5 call on the device


def test_dimp_l2_optimizer_class_on_device(installed):
    amd, _, opt, _, _, o = installed
    from pytracking_amd import synth
    cfg = synth.DIMP50
    w0, feat, bb, sw = synth.dimp_problem(31, 5, cfg, small=dict(C=16, H=10, W=10))
    kw = dict(num_iter=3, feat_stride=16, init_step_length=1.0, gauss_sigma=0.9, hinge_threshold=0.05, init_filter_reg=0.1,
              min_filter_reg=1e-3, alpha_eps=0.0)
    ref = o["L2"](**kw).eval()
    mine = opt.DiMPL2SteepestDescentGN(**kw).eval()
    mine.load_state_dict(ref.state_dict(), strict=True)
    T = torch.from_numpy
    with torch.no_grad():
        w_ref, its_ref, l_ref = ref(T(w0)[None], T(feat), T(bb), sample_weight=T(sw), num_iter=3, compute_losses=True)
        mine = mine.to("cuda")
        w, its, l = mine(T(w0)[None].cuda(), T(feat).cuda(), T(bb).cuda(), sample_weight=T(sw).cuda(), num_iter=3, compute_losses=True)
    _close(w, w_ref, 2e-5)
    _close(torch.stack(its), torch.stack(its_ref), 2e-5)
    assert amd.stats["DiMPL2SteepestDescentGN.fast"] == 1


def test_patch_sampling_and_max2d_on_device(installed):
    amd, _, _, pp, dcf, o = installed
    g = torch.Generator().manual_seed(3)
    im = (torch.rand(1, 3, 240, 320, generator=g) * 255).float()
    pos, sz, out = torch.Tensor([120.0, 150.0]), torch.Tensor([200.0, 200.0]), torch.Tensor([128.0, 128.0])
    p_ref, c_ref = o["sample_patch"](im, pos, sz, out)
    p, c = pp.sample_patch(im.cuda(), pos, sz, out)
    assert torch.equal(c.cpu().float(), c_ref.float())
    _close(p, p_ref, 1e-3)                                           # pixels 0..255: the bilinear weight carries the ulp of the source index
    scales = torch.Tensor([0.9, 1.0, 1.1])
    m_ref, mc_ref = o["sample_patch_multiscale"](im, pos, scales, out)
    m, mc = pp.sample_patch_multiscale(im.cuda(), pos, scales, out)
    assert torch.equal(mc.cpu().float(), mc_ref.float())
    _close(m, m_ref, 1e-3)
    a = torch.randn(4, 19, 19, generator=g)
    v_ref, i_ref = o["max2d"](a)
    v, i = dcf.max2d(a.cuda())
    assert torch.equal(v.cpu(), v_ref) and torch.equal(i.cpu(), i_ref)
    assert amd.stats["sample_patch.fast"] >= 1 and amd.stats["sample_patch_multiscale.fast"] == 1 and amd.stats["max2d.fast"] == 1
