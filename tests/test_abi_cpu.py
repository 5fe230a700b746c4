"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol the
header declares (no compute calls -- there is no GPU here), argument checking returns the documented
status codes, and the host-side mirrors keep the reference's names and state_dict keys."""
import ctypes
import os
import re

import pytest
import torch

from pytracking_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    _lib.build_library()
    return _lib.lib()


def test_header_symbols_exported(L):
    hdr = open(os.path.join(ROOT, "include", "pt_hot.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"pt_status", "pt_sd_params", "pt_profile"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.pt_abi_version() == 1


def test_argument_checks_do_not_launch(L):
    p = _lib.SdParams()
    n = None
    one = ctypes.c_void_p(256)
    assert L.pt_apply_filter_f32(n, 0, n, n, 1, 1, 1, 1, 1, 1, 1, 1, n, 0, n) == -1          # PT_ERR_NULL
    assert L.pt_apply_filter_f32(one, 324, one, one, 0, 8, 18, 18, 4, 4, 19, 19, one, 0, n) == -2   # shape
    assert L.pt_apply_filter_f32(one, 324, one, one, 1, 8, 18, 18, 5, 5, 18, 18, one, 0, n) == -3   # 25 taps
    assert L.pt_apply_filter_f32(one, 8 * 324, one, one, 1, 8, 18, 18, 4, 4, 19, 19, one, 0, n) == -4  # workspace
    assert L.pt_sd_solve_f32(ctypes.byref(p), n, n, 0, n, n, 1, 1, 1, 1, 1, 1, n, n, n, 0, n) == -1
    assert L.pt_prroi_fwd_f32(one, one, one, 1, 1, 0, 4, 1, 2, 2, 1.0, n) == -2
    assert b"workspace" in L.pt_strerror(-4)
    assert L.pt_sd_ws_bytes(50, 512, 18, 18, 4) > 0
    assert L.pt_sd_ws_bytes(0, 512, 18, 18, 4) == 0
    assert L.pt_track_frame_ws_bytes(50, 512, 18, 18, 4) > L.pt_sd_ws_bytes(50, 512, 18, 18, 4)


def test_multifilter_and_lwl_argument_checks(L):
    n = None
    one = ctypes.c_void_p(256)
    # F > 16 filters, even K, W > 256 are outside the multi-filter kernels: no workspace, UNSUPPORTED
    assert L.pt_lwl_ws_bytes(4, 16, 512, 30, 52, 3) > 0
    assert L.pt_lwl_ws_bytes(4, 17, 512, 30, 52, 3) == 0
    assert L.pt_lwl_ws_bytes(4, 16, 512, 30, 52, 4) == 0
    assert L.pt_feat_transpose_mf_ws_bytes(4, 16, 512, 30, 300, 3) == 0
    assert L.pt_apply_filter_mf_f32(n, 0, n, n, 1, 1, 16, 4, 4, 3, n, 0, n) == -1                 # PT_ERR_NULL
    assert L.pt_apply_filter_mf_f32(one, 16 * 16, one, one, 1, 17, 16, 4, 4, 3, one, 1 << 20, n) == -3   # 17 filters
    assert L.pt_apply_filter_mf_f32(one, 8, one, one, 1, 4, 16, 4, 4, 3, one, 1 << 20, n) == -2   # sample stride < C*H*W
    assert L.pt_apply_filter_mf_f32(one, 256, one, one, 1, 4, 16, 4, 4, 3, one, 0, n) == -4        # workspace
    assert L.pt_feat_transpose_mf_f32(one, 256, one, one, 1, 4, 16, 4, 4, 3, one, 0, n) == -4     # workspace
    assert L.pt_lwl_gn_solve_f32(one, one, 256, one, n, 2, 0.1, 0.0, 1, 4, 16, 4, 4, 3, 1, one, n, one, 1 << 30, n) == -1
    assert L.pt_lwl_gn_solve_f32(one, one, 256, one, n, 0, 0.1, 0.0, 1, 4, 16, 4, 4, 3, 65, one, n, one, 1 << 30, n) == -3


def test_atom_joint_gn_argument_checks(L):
    n = None
    one = ctypes.c_void_p(256)
    it = (ctypes.c_int * 2)(3, 3)
    assert L.pt_atom_gn_ws_bytes(30, 256, 64, 18, 18, 4) > 0          # ATOM first frame (atom.py:156-176)
    assert L.pt_atom_gn_ws_bytes(30, 256, 64, 18, 18, 5) == 0         # 25 taps
    assert L.pt_atom_gn_ws_bytes(0, 256, 64, 18, 18, 4) == 0
    assert L.pt_atom_gn_f32(n, one, one, 0, one, one, 0.1, 1e-4, 0.05, 1, 8, 4, 6, 6, 4, it, 2, 1, one, 0, n) == -1
    assert L.pt_atom_gn_f32(one, one, one, 8 * 36, one, one, 0.1, 1e-4, 0.05, 1, 8, 4, 6, 6, 4, None, 2, 1, one, 0, n) == -1
    assert L.pt_atom_gn_f32(one, one, one, 8, one, one, 0.1, 1e-4, 0.05, 1, 8, 4, 6, 6, 4, it, 2, 1, one, 1 << 30, n) == -2
    assert L.pt_atom_gn_f32(one, one, one, 8 * 36, one, one, 0.1, 1e-4, 0.05, 1, 8, 4, 6, 6, 5, it, 2, 1, one, 1 << 30, n) == -3
    assert L.pt_atom_gn_f32(one, one, one, 8 * 36, one, one, 0.1, 1e-4, 0.05, 1, 8, 4, 6, 6, 4, it, 2, 1, one, 0, n) == -4


def test_tomp_argument_checks(L):
    n = None
    one = ctypes.c_void_p(256)
    d = _lib.TompDims(256, 8, 2048, 6, 6, 18, 18, 18)
    assert L.pt_tomp_param_floats(ctypes.byref(d)) == 6 * (4 * 256 * 256 + 4 * 256 + 2 * 256 * 2048 + 2048 + 256 + 4 * 256) \
        + 6 * (8 * 256 * 256 + 8 * 256 + 2 * 256 * 2048 + 2048 + 256 + 6 * 256) + 2 * 256 \
        + (64 * 4 + 64 + 4 * 64) + (256 * 64 + 256 + 4 * 256) + (256 * 256 + 256) + 2 * 256
    assert L.pt_tomp_predict_ws_bytes(ctypes.byref(d), 2, 1, 1) > 0
    assert L.pt_tomp_predict_ws_bytes(ctypes.byref(d), 2, 2, 1) == 0            # the parallel entry point is single-sequence
    assert L.pt_tomp_predict_ws_bytes(ctypes.byref(d), 2, 9, 0) == 0            # more than 8 batch rows
    bad = _lib.TompDims(256, 5, 2048, 6, 6, 18, 18, 18)
    assert L.pt_tomp_param_floats(ctypes.byref(bad)) == 0
    a = [one] * 6
    assert L.pt_tomp_prepared_floats(ctypes.byref(d)) >= 6 * (256 * 256 + 2 * 8 * 256 * 256)
    assert L.pt_tomp_prepared_floats(ctypes.byref(bad)) == 0
    assert L.pt_tomp_prepare_f32(ctypes.byref(d), n, one, n) == -1
    assert L.pt_tomp_prepare_f32(ctypes.byref(bad), one, one, n) == -2
    assert L.pt_tomp_predict_f32(ctypes.byref(d), n, *a, 2, 1, 1, 1, one, one, one, 1 << 30, n) == -1
    assert L.pt_tomp_predict_f32(ctypes.byref(bad), one, *a, 2, 1, 1, 1, one, one, one, 1 << 30, n) == -2
    assert L.pt_tomp_predict_f32(ctypes.byref(d), one, *a, 2, 1, 1, 3, one, one, one, 1 << 30, n) == -2   # num_gth > n_train
    assert L.pt_tomp_predict_f32(ctypes.byref(d), one, *a, 2, 2, 1, 1, one, one, one, 1 << 30, n) == -3
    assert L.pt_tomp_predict_f32(ctypes.byref(d), one, *a, 2, 1, 1, 1, one, one, one, 0, n) == -4
    assert L.pt_tomp_bbreg_param_floats(256) == 256 * 256 + 256 + 4 * (9 * 256 * 256 + 3 * 256) + 4 * 9 * 256 + 4
    assert L.pt_tomp_bbreg_ws_bytes(1, 256, 18, 18) > 0 and L.pt_tomp_bbreg_ws_bytes(1, 100, 18, 18) == 0
    assert L.pt_tomp_bbreg_f32(one, one, one, one, 1, 256, 18, 18, one, 0, n) == -4
    assert L.pt_tomp_bbreg_f32(one, one, one, one, 1, 100, 18, 18, one, 1 << 30, n) == -3
    assert L.pt_tomp_linear_f32(one, one, n, one, 1, 256, 256, 0, n) == -1
    assert L.pt_tomp_posenc_f32(one, 18, 18, 255, 18, n) == -2


def test_clf_head_argument_checks(L):
    n = None
    one = ctypes.c_void_p(256)
    assert L.pt_clf_head_ws_bytes(1, 1024, 512, 18, 18) > 0
    assert L.pt_clf_head_ws_bytes(1, 1000, 512, 18, 18) == 0
    assert L.pt_clf_head_f32(n, one, one, 1, 1024, 512, 18, 18, 1.0, 1e-5, one, 1 << 30, n) == -1
    assert L.pt_clf_head_f32(one, one, one, 0, 1024, 512, 18, 18, 1.0, 1e-5, one, 1 << 30, n) == -2
    assert L.pt_clf_head_f32(one, one, one, 1, 1000, 512, 18, 18, 1.0, 1e-5, one, 1 << 30, n) == -3
    assert L.pt_clf_head_f32(one, one, one, 1, 1024, 512, 18, 18, 1.0, 1e-5, one, 0, n) == -4


def test_localize_argument_checks(L):
    n = None
    one = ctypes.c_void_p(256)
    f2 = (ctypes.c_float * 2)(3.0, 3.0)
    assert L.pt_max2d_f32(n, one, one, 1, 19, 19, n) == -1
    assert L.pt_max2d_f32(one, one, one, 0, 19, 19, n) == -2
    assert L.pt_localize_f32(one, n, None, f2, one, 1, 19, 19, n) == -1
    assert L.pt_localize_f32(one, n, f2, f2, one, 0, 19, 19, n) == -2
    assert L.pt_localize_f32(one, n, f2, f2, one, 9, 19, 19, n) == -3


def test_iou_refine_argument_checks(L):
    n = None
    one = ctypes.c_void_p(256)
    d = _lib.IouDims(256, 256, 256, 256, 36, 36, 18, 18)               # AtomIoUNet defaults (atom_iou_net.py:23)
    assert L.pt_iou_param_floats(ctypes.byref(d)) == 256 * 6400 + 5 * 256 + 256 * 2304 + 5 * 256 + 512 + 1
    assert L.pt_iou_prepared_floats(ctypes.byref(d)) >= 256 * (6400 + 2304)
    assert L.pt_iou_refine_ws_bytes(ctypes.byref(d), 10) > 0
    bad = _lib.IouDims(250, 256, 256, 256, 36, 36, 18, 18)             # 250 * 25 is not a multiple of 32
    assert L.pt_iou_param_floats(ctypes.byref(bad)) == 0 and L.pt_iou_refine_ws_bytes(ctypes.byref(bad), 10) == 0
    f4 = (ctypes.c_float * 4)(1, 1, 1, 1)
    a = [one] * 9
    assert L.pt_iou_refine_f32(ctypes.byref(d), n, *a[1:], 10, 5, f4, 1.0, 0, 0, one, 1 << 30, n) == -1
    assert L.pt_iou_refine_f32(ctypes.byref(d), *a, 0, 5, f4, 1.0, 0, 0, one, 1 << 30, n) == -2
    assert L.pt_iou_refine_f32(ctypes.byref(bad), *a, 10, 5, f4, 1.0, 0, 0, one, 1 << 30, n) == -3
    assert L.pt_iou_refine_f32(ctypes.byref(d), *a, 10, 5, f4, 1.0, 0, 0, one, 0, n) == -4
    assert L.pt_iou_prepare_f32(ctypes.byref(d), one, n, n) == -1
    # the per-frame variant: host proposals / pinned host results, <= 16 proposals (checked before any device call)
    a8 = [one] * 8
    assert L.pt_iou_refine_sync_f32(ctypes.byref(d), *a8[:7], n, 10, 5, f4, 1.0, 0, 0, one, 1 << 30, n) == -1
    assert L.pt_iou_refine_sync_f32(ctypes.byref(d), *a8[:6], n, one, 10, 5, f4, 1.0, 0, 0, one, 1 << 30, n) == -1
    assert L.pt_iou_refine_sync_f32(ctypes.byref(d), *a8, 17, 5, f4, 1.0, 0, 0, one, 1 << 30, n) == -3


def test_tomp_mirror_contract():
    """Constructor signatures and refusals of the ToMP mirror (no device work)."""
    from pytracking_amd import transformer as TM
    tr = TM.Transformer(d_model=128, nhead=4, num_encoder_layers=1, num_decoder_layers=1, dim_feedforward=256)
    pred = TM.FilterPredictor(tr, feature_sz=6)
    keys = list(pred.state_dict().keys())
    assert "transformer.encoder.layers.0.self_attn.in_proj_weight" in keys
    assert "transformer.decoder.layers.0.multihead_attn.out_proj.bias" in keys
    assert "transformer.decoder.norm.weight" in keys and "box_encoding.4.running_var" in keys
    assert "query_embed_fg_decoder.weight" in keys and "query_embed_test.weight" in keys
    with pytest.raises(NotImplementedError):
        TM.Transformer(normalize_before=True)
    with pytest.raises(NotImplementedError):
        TM.Transformer(activation="gelu")
    x = torch.zeros(2, 1, 128, 6, 6)
    with pytest.raises(NotImplementedError):                # training mode
        pred.predict_filter(x, x[:1], torch.zeros(2, 1, 6, 6), torch.zeros(2, 1, 4, 6, 6))
    with pytest.raises(RuntimeError):                        # CPU tensors
        pred.eval().predict_filter(x, x[:1], torch.zeros(2, 1, 6, 6), torch.zeros(2, 1, 4, 6, 6))


def test_activation_recognition():
    """The reference tracker hands its activations over as lambdas (atom.py:444-466); the mirror recognises the two
    the fused solvers implement and nothing else."""
    import torch.nn.functional as F
    from pytracking_amd.optimization import MLU, activation_kind, ConjugateGradient, ConvProblem, GaussNewtonCG, \
        FactorizedConvProblem
    assert activation_kind(lambda x: x) == ("identity", None)
    assert activation_kind(None) == ("identity", None)
    assert activation_kind(lambda x: F.elu(F.leaky_relu(x, 1 / 0.05), 0.05)) == ("mlu", 0.05)
    assert activation_kind(MLU(0.05)) == ("mlu", 0.05)
    assert activation_kind(torch.nn.ReLU(inplace=True)) == (None, None)
    assert activation_kind(lambda x: x * 2) == (None, None)
    with pytest.raises(NotImplementedError):
        ConjugateGradient(ConvProblem([], [], [], [], torch.nn.ReLU()), [])
    with pytest.raises(NotImplementedError):
        GaussNewtonCG(FactorizedConvProblem([], [], [], [], None, [], torch.nn.ReLU(), MLU(0.05)), [])
    with pytest.raises(NotImplementedError):
        GaussNewtonCG(FactorizedConvProblem([], [], [], [], None, [], None, MLU(0.05)), [], cg_eps=1e-3)


def test_lwl_mirror_contract():
    from pytracking_amd import steepestdescent as sd
    res = sd.LWTLResidual(init_filter_reg=0.05)
    assert set(res.state_dict().keys()) == {"filter_reg"}                 # loss_residual_modules.py:13
    opt = sd.GNSteepestDescent(res, num_iter=3, residual_batch_dim=1)
    assert opt.num_iter == 3 and opt.steplength_reg == 0.0
    with pytest.raises(RuntimeError, match="MI355X"):                     # no CPU fallback on the product path
        opt(torch.zeros(1, 2, 16, 3, 3), feat=torch.zeros(1, 1, 16, 4, 4), label=torch.zeros(1, 1, 2, 4, 4))
    with pytest.raises(NotImplementedError):                              # default residual_batch_dim=0 is not LWL's
        sd.GNSteepestDescent(res, num_iter=1)(torch.zeros(1, 2, 16, 3, 3), feat=torch.zeros(1, 1, 16, 4, 4),
                                              label=torch.zeros(1, 1, 2, 4, 4))


def test_module_mirror_state_dict_keys():
    from pytracking_amd import optimizer
    m = optimizer.DiMPSteepestDescentGN(num_iter=5, init_step_length=0.9, init_filter_reg=0.1, init_gauss_sigma=0.9,
                                        num_dist_bins=100, bin_displacement=0.1, mask_init_factor=3.0)
    # keys a reference checkpoint carries for this module (optimizer.py:39-72)
    assert set(m.state_dict().keys()) == {"log_step_length", "filter_reg", "label_map_predictor.weight",
                                          "target_mask_predictor.0.weight", "spatial_weight_predictor.weight"}
    assert m.label_map_predictor.weight.shape == (1, 100, 1, 1)
    from pytracking_amd import synth
    torch.testing.assert_close(m.label_map_predictor.weight.reshape(-1),
                               torch.from_numpy(synth.gauss_lut(100, 0.1, 0.9)), atol=1e-6, rtol=0)
    torch.testing.assert_close(m.target_mask_predictor[0].weight.reshape(-1),
                               torch.from_numpy(synth.mask_lut(100, 0.1, 3.0)), atol=1e-6, rtol=0)
    p = optimizer.PrDiMPSteepestDescentNewton()
    assert set(p.state_dict().keys()) == {"log_step_length", "filter_reg"}


def test_product_path_refuses_cpu_tensors():
    from pytracking_amd import filter as F
    with pytest.raises(RuntimeError, match="MI355X"):
        F.apply_filter(torch.zeros(1, 4, 6, 6), torch.zeros(1, 4, 4, 4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pytracking_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_iou_head_tensor_gathering_matches_attribute_access():
    """`iou_refine._iou_tensors` reads the IoU head's parameters through the module registries (per-frame path); it must hand out the
    very tensor objects attribute access does, in the packed order, and fall back for objects that are not nn.Modules."""
    import types
    import torch
    from pytracking_amd import iou_refine as IR

    def block(k):
        m = torch.nn.Module()
        m.linear, m.bn, m.relu = torch.nn.Linear(8 * k * k, 32), torch.nn.BatchNorm2d(32), torch.nn.ReLU()
        return m
    net = torch.nn.Module()
    net.fc3_rt, net.fc4_rt, net.iou_predictor = block(5), block(3), torch.nn.Linear(64, 1)
    want = []
    for blk in (net.fc3_rt, net.fc4_rt):
        want += [blk.linear.weight, blk.linear.bias, blk.bn.weight, blk.bn.bias, blk.bn.running_mean, blk.bn.running_var]
    want += [net.iou_predictor.weight, net.iou_predictor.bias]
    got = IR._iou_tensors(net)
    assert len(got) == 14 and all(a is b for a, b in zip(got, want))
    loose = types.SimpleNamespace(fc3_rt=net.fc3_rt, fc4_rt=net.fc4_rt, iou_predictor=net.iou_predictor)   # no registries: attribute path
    assert all(a is b for a, b in zip(IR._iou_tensors(loose), want))


def test_round5_entry_points_argument_checks(L):
    """pt_track_frame_full_* / pt_sd_solve_batch_f32 / pt_stream_probe_f32 refuse bad arguments before anything is queued (no GPU here)."""
    n = None
    one = ctypes.c_void_p(256)
    # full frame: null descriptor / sub-structs, proposal count, workspace
    assert L.pt_track_frame_full_ws_bytes(None) == 0
    f = _lib.FrameFull()
    assert L.pt_track_frame_full_ws_bytes(ctypes.byref(f)) == 0                            # sd / loc / glue / iou_dims missing
    assert L.pt_track_frame_full_launch_f32(ctypes.byref(f), one, one, 1 << 30, n) == _lib.PT_ERR_NULL
    sd, loc, glue, dims = _lib.SdParams(), _lib.LocalizeState(), _lib.FrameGlue(), _lib.IouDims(256, 256, 256, 256, 36, 36, 18, 18)
    f.sd, f.loc, f.glue, f.iou_dims = ctypes.pointer(sd), ctypes.pointer(loc), ctypes.pointer(glue), ctypes.pointer(dims)
    f.n, f.Cin, f.C, f.H, f.W, f.K, f.num_iter = 50, 1024, 512, 18, 18, 4, 5
    glue.num_random = 9
    need = L.pt_track_frame_full_ws_bytes(ctypes.byref(f))
    assert need > L.pt_track_frame_head_ws_bytes(50, 1024, 512, 18, 18, 4) + L.pt_iou_refine_ws_bytes(ctypes.byref(dims), 10) - 1
    glue.num_random = 16                                                                    # P = 17 > 16 proposals
    assert L.pt_track_frame_full_ws_bytes(ctypes.byref(f)) == 0
    assert L.pt_track_frame_full_launch_f32(ctypes.byref(f), one, one, 1 << 30, n) == _lib.PT_ERR_UNSUPPORTED
    glue.num_random = 9
    f.scores_out = 256
    assert L.pt_track_frame_full_launch_f32(ctypes.byref(f), one, one, 0, n) == _lib.PT_ERR_WORKSPACE
    assert L.pt_track_frame_full_launch_f32(ctypes.byref(f), n, one, 1 << 30, n) == _lib.PT_ERR_NULL
    f.K = 5                                                                                 # 25 taps: the frame's solver refuses
    assert L.pt_track_frame_full_ws_bytes(ctypes.byref(f)) > 0
    f.K = 4
    # batched solve
    arr = (ctypes.c_void_p * 2)(256, 256)
    assert L.pt_sd_solve_batch_f32(ctypes.byref(sd), 2, None, arr, 1, arr, None, 4, 16, 6, 6, 4, 1, arr, None, arr, 1 << 20, n, None, 0) == _lib.PT_ERR_NULL
    assert L.pt_sd_solve_batch_f32(ctypes.byref(sd), 0, arr, arr, 1, arr, None, 4, 16, 6, 6, 4, 1, arr, None, arr, 1 << 20, n, None, 0) == _lib.PT_ERR_SHAPE
    assert L.pt_sd_solve_batch_f32(ctypes.byref(sd), 2, arr, arr, 1, arr, None, 4, 16, 6, 6, 4, 1, arr, None, arr, 1 << 20, n, None, 2) == _lib.PT_ERR_SHAPE
    assert L.pt_sd_solve_batch_f32(ctypes.byref(sd), 2, arr, arr, 1, arr, None, 4, 16, 6, 6, 4, 1, arr, None, arr, 1 << 20, n, arr, 16) == _lib.PT_ERR_UNSUPPORTED
    # probe
    assert L.pt_stream_probe_f32(n, 1024, one, 1, n) == _lib.PT_ERR_NULL
    assert L.pt_stream_probe_f32(one, 2, one, 1, n) == _lib.PT_ERR_SHAPE
    assert L.pt_stream_probe_f32(ctypes.c_void_p(260), 1024, one, 1, n) == _lib.PT_ERR_SHAPE   # not 16-byte aligned
    L.pt_host_buffer_forget(one)                                                                 # unknown pointer: no-op


def test_round6_entry_points_argument_checks(L):
    """Frame chains (pt_track_frame_chain_f32 / pt_track_frame_flush_f32), the two-stream acknowledgement of the one-call frame and the
    up-front refinement checks refuse bad arguments before anything is queued (no GPU here)."""
    n = None
    one = ctypes.c_void_p(256)
    sd = _lib.SdParams()
    pend = _lib.FramePending()
    # flush: nulls, nothing pending, bad shapes, workspace, iteration counts that can never be pending
    assert L.pt_track_frame_flush_f32(None, one, 50, 512, 18, 18, 4, one, 1 << 30, n) == _lib.PT_ERR_NULL
    assert L.pt_track_frame_flush_f32(ctypes.byref(pend), None, 50, 512, 18, 18, 4, one, 1 << 30, n) == _lib.PT_ERR_NULL
    assert L.pt_track_frame_flush_f32(ctypes.byref(pend), one, 50, 512, 18, 18, 4, one, 1 << 30, n) == 0            # iters == 0: no-op
    pend.iters = 5
    assert L.pt_track_frame_flush_f32(ctypes.byref(pend), one, 0, 512, 18, 18, 4, one, 1 << 30, n) == _lib.PT_ERR_SHAPE
    assert L.pt_track_frame_flush_f32(ctypes.byref(pend), one, 50, 512, 18, 18, 4, one, 16, n) == _lib.PT_ERR_WORKSPACE
    pend.iters = 1
    assert L.pt_track_frame_flush_f32(ctypes.byref(pend), one, 50, 512, 18, 18, 4, one, 1 << 30, n) == _lib.PT_ERR_SHAPE
    pend.iters = 65
    assert L.pt_track_frame_flush_f32(ctypes.byref(pend), one, 50, 512, 18, 18, 4, one, 1 << 30, n) == _lib.PT_ERR_SHAPE
    # chain: the pending block is mandatory, its count cannot be negative; the frame's own checks come next
    args = [ctypes.byref(sd), one, one, one, None, one, 0, 50, 512, 18, 18, 4, 5, one, one, one, 1 << 30]
    assert L.pt_track_frame_chain_f32(*args, None, 1, n) == _lib.PT_ERR_NULL
    pend.iters = -1
    assert L.pt_track_frame_chain_f32(*args, ctypes.byref(pend), 1, n) == _lib.PT_ERR_SHAPE
    pend.iters = 0
    bad = list(args); bad[6] = 50                                                                              # slot == n
    assert L.pt_track_frame_chain_f32(*bad, ctypes.byref(pend), 1, n) == _lib.PT_ERR_SHAPE
    bad = list(args); bad[16] = 16                                                                             # workspace too small
    assert L.pt_track_frame_chain_f32(*bad, ctypes.byref(pend), 1, n) == _lib.PT_ERR_WORKSPACE
    # one-call frame: two streams + an update need the caller's acknowledgement; refinement arguments are checked before the head is queued
    f = _lib.FrameFull()
    loc, glue, dims = _lib.LocalizeState(), _lib.FrameGlue(), _lib.IouDims(256, 256, 256, 256, 36, 36, 18, 18)
    f.sd, f.loc, f.glue, f.iou_dims = ctypes.pointer(sd), ctypes.pointer(loc), ctypes.pointer(glue), ctypes.pointer(dims)
    f.n, f.Cin, f.C, f.H, f.W, f.K, f.num_iter = 50, 1024, 512, 18, 18, 4, 5
    glue.num_random = 9
    f.scores_out = 256
    f.aux_stream, f.aux_reordered_update_ok = 512, 0
    assert L.pt_track_frame_full_launch_f32(ctypes.byref(f), one, one, 1 << 30, n) == _lib.PT_ERR_UNSUPPORTED
    assert ctypes.sizeof(_lib.FramePending) == 12 and _lib.FrameFull.aux_reordered_update_ok.offset == _lib.FrameFull.aux_stream.offset + 8


def test_round6_graph_mode_entry_points_argument_checks(L):
    """pt_frame_full.dyn: block size, host-side fill and the word wait refuse bad arguments without touching a device."""
    n = None
    one = ctypes.c_void_p(256)
    assert L.pt_track_frame_full_dyn_bytes() >= 512 and L.pt_track_frame_full_dyn_bytes() % 256 == 0
    f = _lib.FrameFull()
    assert L.pt_track_frame_full_dyn_fill_f32(ctypes.byref(f), 1.0, one, one, 1 << 30, one) == _lib.PT_ERR_NULL      # sd / loc / glue / iou_dims missing
    sd, loc, glue, dims = _lib.SdParams(), _lib.LocalizeState(), _lib.FrameGlue(), _lib.IouDims(256, 256, 256, 256, 36, 36, 18, 18)
    f.sd, f.loc, f.glue, f.iou_dims = ctypes.pointer(sd), ctypes.pointer(loc), ctypes.pointer(glue), ctypes.pointer(dims)
    f.n, f.Cin, f.C, f.H, f.W, f.K, f.num_iter = 50, 1024, 512, 18, 18, 4, 5
    glue.num_random = 9
    f.scores_out = 256
    assert L.pt_track_frame_full_dyn_fill_f32(ctypes.byref(f), 1.0, one, one, 1 << 30, n) == _lib.PT_ERR_NULL
    f.slot = 50
    assert L.pt_track_frame_full_dyn_fill_f32(ctypes.byref(f), 1.0, one, one, 1 << 30, one) == _lib.PT_ERR_SHAPE       # slot == n
    f.slot = 3
    assert L.pt_track_frame_full_dyn_fill_f32(ctypes.byref(f), 0.0, one, one, 1 << 30, one) == _lib.PT_ERR_SHAPE       # seq 0 = "no word"
    assert L.pt_track_frame_full_dyn_fill_f32(ctypes.byref(f), 1.0, one, one, 16, one) == _lib.PT_ERR_WORKSPACE
    f.dyn, f.aux_stream, f.aux_reordered_update_ok = 264, 512, 1                    # misaligned block / two streams: refused, nothing queued
    assert L.pt_host_wait_word_f32(n, 1.0, n) == _lib.PT_ERR_NULL
