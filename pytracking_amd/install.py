"""Install the MI355X hot path behind the reference's own Python call boundary.

    import pytracking_amd.install as amd
    amd.install()                     # before the tracker / network is constructed
    # ... `pytracking.tracker.dimp.DiMP`, `atom.ATOM` etc. now run unchanged on top of libpt_hot.so

What gets rebound (import path = contract; nothing in the reference tree is edited):

  ltr.models.layers.filter.apply_filter / apply_feat_transpose / filter_gradient      -> pytracking_amd.filter
  ltr.models.target_classifier.optimizer.{DiMPSteepestDescentGN, DiMPL2SteepestDescentGN,
                                           PrDiMPSteepestDescentNewton}               -> pytracking_amd.optimizer
  ltr.external.PreciseRoIPooling.pytorch.prroi_pool.PrRoIPool2D  (module is CREATED: the submodule is empty)
                                                                                      -> pytracking_amd.prroi_pool
  pytracking.libs.optimization.ConjugateGradient  (ConvProblem + MLU fast path)       -> pytracking_amd.optimization
  pytracking.libs.optimization.GaussNewtonCG      (FactorizedConvProblem fast path)   -> pytracking_amd.optimization
  pytracking.libs.operation.conv2d                (mode='same', one output channel: ATOM's per-frame classification)
                                                                                      -> pytracking_amd.filter.corr_raw
  pytracking.features.preprocessing.sample_patch / sample_patch_multiscale (device images)
                                                                                      -> pytracking_amd.preprocessing
  ltr.models.lwl.loss_residual_modules.LWTLResidual, ltr.models.meta.steepestdescent.GNSteepestDescent
                                                  (LWL few-shot learner)              -> pytracking_amd.steepestdescent
  ltr.models.transformer.transformer.Transformer, ltr.models.transformer.filter_predictor.FilterPredictor,
  ltr.models.transformer.heads.{LinearFilterClassifier, DenseBoxRegressor}
                                                  (ToMP model predictor, inference)   -> pytracking_amd.transformer
  ltr.models.target_classifier.features.residual_bottleneck  (final 3x3 conv + InstanceL2Norm, inference)
                                                                                      -> pytracking_amd.features
  pytracking.libs.dcf.max2d, pytracking.tracker.dimp.dimp.DiMP.localize_advanced,
  pytracking.tracker.tomp.tomp.ToMP.localize_advanced  (score-map localisation)       -> pytracking_amd.localization
  pytracking.tracker.dimp.dimp.DiMP.optimize_boxes_default / optimize_boxes_relative,
  pytracking.tracker.atom.atom.ATOM.optimize_boxes                                    (IoU-guided box refinement)
                                                                                      -> pytracking_amd.iou_refine

Dispatch rule of the rebound *functions*: device fp32 tensors of a shape the gfx950 kernels cover go to the C ABI;
everything else (CPU tensors, dilations, grouped filters, K*K > 16, more than 16 filters) is outside the hot path and
is handed to the reference's ORIGINAL function object -- its own stock-PyTorch code, not a re-implementation --
unless `strict=True`, in which case it raises.  The rebound *classes* have no such escape: they run the fused
solver or raise.
"""
import collections
import importlib
import sys
import types

import torch

from . import filter as _filter
from . import optimization as _optimization
from . import optimizer as _optimizer
from . import prroi_pool as _prroi
from . import steepestdescent as _sd

_state = {"installed": False, "originals": {}}

# which branch every rebound symbol took, per call: "<symbol>.fast" (gfx950 path) / "<symbol>.reference" (handed to the
# reference's own code).  tests/test_trackers_on_device.py asserts on it that a tracker running on the GPU really went
# through the HIP library; reset with `stats.clear()`.
stats = collections.Counter()


def _hit(name, fast):
    stats[name + (".fast" if fast else ".reference")] += 1
    return fast


def _covered(feat, filt, dilation_factors=None):
    if not (isinstance(feat, torch.Tensor) and feat.is_cuda and feat.dtype == torch.float32 and filt.is_cuda
            and dilation_factors is None):
        return False
    if torch.is_grad_enabled() and feat.requires_grad:   # feature gradients (offline training): the reference's convs
        return False
    if filt.dim() == 5:                                  # multi-filter (LWL): <= 16 filters, 1x1 / 3x3, groups == 1
        return (filt.shape[1] <= 16 and filt.shape[-1] == filt.shape[-2] and filt.shape[-1] in (1, 3)
                and filt.shape[-3] == feat.shape[-3] and feat.shape[-1] <= 256)
    return filt.dim() == 4 and filt.shape[-1] * filt.shape[-2] <= 16


def _make_dispatchers(orig_mod, strict):
    o_apply, o_adj, o_grad = orig_mod.apply_filter, orig_mod.apply_feat_transpose, orig_mod.filter_gradient

    def apply_filter(feat, filter, dilation_factors=None):
        if _hit("apply_filter", _covered(feat, filter, dilation_factors)):
            return _filter.apply_filter(feat, filter)
        if strict:
            raise NotImplementedError("apply_filter: configuration outside the gfx950 hot path")
        return o_apply(feat, filter, dilation_factors)

    def apply_feat_transpose(feat, input, filter_ksz, training=True, groups=1):
        ksz = (filter_ksz, filter_ksz) if isinstance(filter_ksz, int) else tuple(filter_ksz)
        mf = input.dim() == 5 and input.shape[2] <= 16 and ksz[0] == ksz[1] and ksz[0] in (1, 3) and feat.shape[-1] <= 256
        if _hit("apply_feat_transpose", feat.is_cuda and feat.dtype == torch.float32 and groups == 1
                and (mf or (input.dim() == 4 and ksz[0] * ksz[1] <= 16))
                and not (torch.is_grad_enabled() and feat.requires_grad)):
            return _filter.apply_feat_transpose(feat, input, ksz, training=training, groups=groups)
        if strict:
            raise NotImplementedError("apply_feat_transpose: configuration outside the gfx950 hot path")
        return o_adj(feat, input, filter_ksz, training=training, groups=groups)

    def filter_gradient(feat, filter, label=None, training=True):
        if _hit("filter_gradient", _covered(feat, filter)):
            return _filter.filter_gradient(feat, filter, label=label, training=training)
        if strict:
            raise NotImplementedError("filter_gradient: configuration outside the gfx950 hot path")
        return o_grad(feat, filter, label=label, training=training)

    for fn, o in ((apply_filter, o_apply), (apply_feat_transpose, o_adj), (filter_gradient, o_grad)):
        fn.__doc__ = o.__doc__
        fn.__wrapped__ = o
    return apply_filter, apply_feat_transpose, filter_gradient


def _optimizer_class(fused_cls, ref_cls, strict):
    """The class bound under the reference's name: the gfx950 module (same parameters, same state_dict keys) whose
    forward runs the fused solver for device tensors under no_grad and hands every other call (CPU tensors, offline
    training with requires_grad inputs, K*K > 16) to the reference class's own forward on the same parameters."""
    import ltr.models.layers.activation as ract
    import ltr.models.layers.distance as rdist

    class Dispatching(fused_cls):
        __doc__ = ref_cls.__doc__

        def __init__(self, *args, **kw):
            super().__init__(*args, **kw)
            # parameter-free helper modules the reference forward reads (optimizer.py:44,74-82)
            if hasattr(self, "num_dist_bins"):
                self.distance_map = rdist.DistanceMap(self.num_dist_bins, self.bin_displacement)
                if self.score_act == 'bentpar':
                    self.score_activation = ract.BentIdentPar(self.act_param)
                    self.score_activation_deriv = ract.BentIdentParDeriv(self.act_param)
                else:
                    self.score_activation = ract.LeakyReluPar()
                    self.score_activation_deriv = ract.LeakyReluParDeriv()

        def forward(self, weights, feat, bb, sample_weight=None, num_iter=None, compute_losses=True):
            fused = (feat.is_cuda and weights.is_cuda and feat.dtype == torch.float32
                     and weights.shape[-1] == weights.shape[-2] and weights.shape[-1] ** 2 <= 16
                     and (sample_weight is None or isinstance(sample_weight, torch.Tensor))
                     and not (torch.is_grad_enabled() and (weights.requires_grad or feat.requires_grad)))
            if _hit(ref_cls.__name__, fused):
                return fused_cls.forward(self, weights, feat, bb, sample_weight=sample_weight, num_iter=num_iter,
                                         compute_losses=compute_losses)
            if strict:
                raise NotImplementedError(f"{ref_cls.__name__}: call outside the gfx950 hot path")
            return ref_cls.forward(self, weights, feat, bb, sample_weight=sample_weight, num_iter=num_iter,
                                   compute_losses=compute_losses)

    for name, member in vars(ref_cls).items():           # helper methods the reference forward calls (get_label_density ...)
        if callable(member) and not name.startswith("__") and not hasattr(fused_cls, name):
            setattr(Dispatching, name, member)
    Dispatching.__name__ = Dispatching.__qualname__ = ref_cls.__name__
    return Dispatching


def provide_prroi_module(orig=None):
    """Create `ltr.external.PreciseRoIPooling.pytorch.prroi_pool` (an empty git submodule in the reference) in
    sys.modules so that `from ltr.external.PreciseRoIPooling.pytorch.prroi_pool import PrRoIPool2D`
    (initializer.py:4, atom_iou_net.py:4) resolves to the HIP implementation.  If an implementation was already
    registered under that name (a CPU stand-in in a test harness), CPU tensors keep going to it; modules that imported
    the class by name before this call are re-pointed."""
    for pkg in ("ltr.external", "ltr.external.PreciseRoIPooling", "ltr.external.PreciseRoIPooling.pytorch"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    name = "ltr.external.PreciseRoIPooling.pytorch.prroi_pool"
    prev = sys.modules.get(name)
    prev_cls = getattr(prev, "PrRoIPool2D", None)

    class PrRoIPool2D(_prroi.PrRoIPool2D):
        __doc__ = _prroi.PrRoIPool2D.__doc__

        def forward(self, features, rois):
            if _hit("PrRoIPool2D", features.is_cuda or prev_cls is None):
                return _prroi.PrRoIPool2D.forward(self, features, rois)
            return prev_cls(self.pooled_height, self.pooled_width, self.spatial_scale)(features, rois)

    pm = types.ModuleType(name)
    pm.PrRoIPool2D = PrRoIPool2D
    pm.prroi_pool2d = _prroi.prroi_pool2d
    sys.modules[name] = pm
    parent = sys.modules["ltr.external.PreciseRoIPooling.pytorch"]
    prev_attr = getattr(parent, "prroi_pool", None)
    parent.prroi_pool = pm
    importers = []
    if prev_cls is not None:
        for modname in ("ltr.models.target_classifier.initializer", "ltr.models.bbreg.atom_iou_net"):
            mod = sys.modules.get(modname)
            if mod is not None and getattr(mod, "PrRoIPool2D", None) is prev_cls:
                mod.PrRoIPool2D = PrRoIPool2D
                importers.append((mod, prev_cls))
    if orig is not None:
        orig["prroi"] = (name, prev, prev_attr, importers)
    return pm


def _install_tomp(orig, strict):
    """ToMP model predictor (inference): the reference's tompnet constructors (ltr/models/tracking/tompnet.py:106-118)
    reach these classes through module attributes, so rebinding the attributes is enough."""
    from . import transformer as _tm
    try:
        tmod = importlib.import_module("ltr.models.transformer.transformer")
        pmod = importlib.import_module("ltr.models.transformer.filter_predictor")
        hmod = importlib.import_module("ltr.models.transformer.heads")
    except Exception:                   # torchvision (heads.py:3) missing: leave the transformer family alone
        return
    orig["tomp"] = (tmod.Transformer, pmod.FilterPredictor, hmod.LinearFilterClassifier, hmod.DenseBoxRegressor)
    ref_tr, ref_fp, ref_cls, ref_reg = orig["tomp"]

    def covered(d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048, dropout=0.1,
                activation="relu", normalize_before=False, return_intermediate_dec=False):
        hd = d_model // max(nhead, 1)
        return (activation == "relu" and not normalize_before and not return_intermediate_dec and d_model % 128 == 0
                and d_model <= 512 and d_model % nhead == 0 and hd in (16, 32, 64) and nhead <= 16
                and dim_feedforward % 64 == 0 and num_encoder_layers <= 16 and num_decoder_layers <= 16)

    class Transformer(ref_tr):
        """Covered configuration -> gfx950 parameter container; anything else -> the reference class."""

        def __new__(cls, *args, **kw):
            if covered(*args, **kw):
                return _tm.Transformer(*args, **kw)
            if strict:
                raise NotImplementedError("Transformer: configuration outside the gfx950 hot path")
            return ref_tr.__new__(cls)

    class FilterPredictor(ref_fp):
        def __new__(cls, transformer, *args, **kw):
            if isinstance(transformer, _tm.Transformer):
                return _tm.FilterPredictor(transformer, *args, **kw)
            if strict:
                raise NotImplementedError("FilterPredictor: transformer outside the gfx950 hot path")
            return ref_fp.__new__(cls)

    class DenseBoxRegressor(ref_reg):
        def __new__(cls, num_channels, *args, **kw):
            if num_channels % 64 == 0 and num_channels <= 512:
                return _tm.DenseBoxRegressor(num_channels, *args, **kw)
            if strict:
                raise NotImplementedError("DenseBoxRegressor: channel count outside the gfx950 hot path")
            return ref_reg.__new__(cls)

    tmod.Transformer = Transformer
    pmod.FilterPredictor = FilterPredictor
    hmod.LinearFilterClassifier = _tm.LinearFilterClassifier
    hmod.DenseBoxRegressor = DenseBoxRegressor


def _install_clf_head(orig, strict):
    """Classification-feature head: `clf_features.residual_bottleneck(...)` is reached through the module attribute by
    every network constructor (dimpnet.py:169, tompnet.py:99).  Covered argument sets build the fused module around the
    reference's own layer objects; on a CPU tensor / in training it runs those layers as `nn.Sequential` would."""
    from . import features as _fm
    try:
        fmod = importlib.import_module("ltr.models.target_classifier.features")
    except Exception:                   # torchvision (features.py:4) missing
        return
    ref_fn = fmod.residual_bottleneck
    orig["clf_head"] = ref_fn

    class ClfHead(_fm.ClfHead):
        def forward(self, x):
            fused = (x.is_cuda and x.dtype == torch.float32 and not self.training
                     and not (torch.is_grad_enabled() and (x.requires_grad or self[0].weight.requires_grad)))
            if _hit("residual_bottleneck", fused):
                return _fm.ClfHead.forward(self, x)
            if strict:
                raise NotImplementedError("clf feature head: call outside the gfx950 hot path")
            return torch.nn.Sequential.forward(self, x)

    def residual_bottleneck(*args, **kw):
        seq = ref_fn(*args, **kw)
        if _fm.covered(*args, **kw) and len(seq) == 2:
            return ClfHead(seq[0], seq[1])                # the reference's Conv2d and InstanceL2Norm objects
        if strict:
            raise NotImplementedError("residual_bottleneck: configuration outside the gfx950 hot path")
        return seq

    residual_bottleneck.__doc__ = ref_fn.__doc__
    residual_bottleneck.__wrapped__ = ref_fn
    fmod.residual_bottleneck = residual_bottleneck


def _install_localization(orig, strict):
    """`dcf.max2d` (function, reached as `dcf.max2d` by the trackers) and the `localize_advanced` methods of DiMP / ToMP:
    one launch + one 32-byte copy per frame instead of ~10 host synchronisations."""
    from . import localization as _loc
    try:
        dmod = importlib.import_module("pytracking.libs.dcf")
    except Exception:
        return
    ref_max2d = dmod.max2d
    orig["localization"] = {"max2d": ref_max2d}

    def max2d(a):
        if _hit("max2d", a.is_cuda and a.dtype == torch.float32):
            return _loc.max2d(a)
        if strict:
            raise NotImplementedError("max2d: tensor outside the gfx950 hot path")
        return ref_max2d(a)

    max2d.__doc__, max2d.__wrapped__ = ref_max2d.__doc__, ref_max2d
    dmod.max2d = max2d
    for modname, clsname, fn in (("pytracking.tracker.dimp.dimp", "DiMP", _loc.localize_advanced),
                                 ("pytracking.tracker.tomp.tomp", "ToMP", _loc.localize_advanced_tomp)):
        try:
            cls = getattr(importlib.import_module(modname), clsname)
        except Exception:               # tracker module not importable here (cv2 ...): nothing to patch
            continue
        ref_method = cls.localize_advanced
        orig["localization"][(modname, clsname)] = ref_method

        def method(self, scores, sample_pos, sample_scales, _fast=fn, _ref=ref_method, _name=clsname + ".localize_advanced"):
            if _hit(_name, scores.is_cuda and scores.dtype == torch.float32 and scores.dim() == 3 and scores.shape[0] <= 8):
                return _fast(self, scores, sample_pos, sample_scales)
            if strict:
                raise NotImplementedError("localize_advanced: scores outside the gfx950 hot path")
            return _ref(self, scores, sample_pos, sample_scales)

        method.__doc__, method.__wrapped__ = ref_method.__doc__, ref_method
        cls.localize_advanced = method


def _install_iou_refine(orig, strict):
    """`DiMP.optimize_boxes_default` / `optimize_boxes_relative` (dimp.py:725-788): all refinement iterations in one
    device-side sequence instead of one autograd graph per iteration."""
    from . import iou_refine as _ir
    try:
        cls = importlib.import_module("pytracking.tracker.dimp.dimp").DiMP
    except Exception:
        return
    orig["iou_refine"] = [(cls, "optimize_boxes_default", cls.optimize_boxes_default),
                          (cls, "optimize_boxes_relative", cls.optimize_boxes_relative)]
    targets = [(cls, "optimize_boxes_default", _ir.optimize_boxes_default, lambda t: t.net.bb_regressor),
               (cls, "optimize_boxes_relative", _ir.optimize_boxes_relative, lambda t: t.net.bb_regressor)]
    try:
        acls = importlib.import_module("pytracking.tracker.atom.atom").ATOM
        orig["iou_refine"].append((acls, "optimize_boxes", acls.optimize_boxes))
        targets.append((acls, "optimize_boxes", _ir.optimize_boxes_atom, lambda t: t.iou_predictor))
    except Exception:
        pass
    for cls, name, fast, get_net in targets:
        ref_method = getattr(cls, name)

        def method(self, iou_features, init_boxes, _fast=fast, _ref=ref_method, _net=get_net,
                   _name=cls.__name__ + "." + name):
            feats = list(iou_features)
            ok = (len(feats) == 2 and all(f.is_cuda and f.dtype == torch.float32 and f.shape[0] == 1 for f in feats)
                  and not _net(self).training)
            if ok:
                try:
                    out = _fast(self, feats, init_boxes)
                    _hit(_name, True)
                    return out
                except NotImplementedError:
                    if strict:
                        raise
            _hit(_name, False)
            if strict and not ok:
                raise NotImplementedError(f"{_ref.__name__}: call outside the gfx950 hot path")
            return _ref(self, iou_features, init_boxes)

        method.__doc__, method.__wrapped__ = ref_method.__doc__, ref_method
        setattr(cls, name, method)


def _install_operation(orig, strict):
    """`pytracking.libs.operation.conv2d` (operation.py:6-33; TensorList-lifted): ATOM classifies every frame with
    `operation.conv2d(sample_x, self.filter, mode='same')` (atom.py:300-302).  A single-output-channel 'same'
    correlation on device tensors goes to the gfx950 correlation pass; every other use (full inner products of
    optim.py:58-99, 1x1 projections, autograd) keeps the reference's function."""
    try:
        omod = importlib.import_module("pytracking.libs.operation")
        tl = importlib.import_module("pytracking.libs.tensorlist")
    except Exception:
        return
    ref_conv2d = omod.conv2d
    ref_single = getattr(ref_conv2d, "__wrapped__", None)     # functools.wraps keeps the un-lifted function
    if ref_single is None:
        return
    orig["operation"] = ref_conv2d

    def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, mode=None):
        fused = (weight is not None and mode == 'same' and bias is None and stride == 1 and dilation == 1 and groups == 1
                 and padding == 0 and isinstance(input, torch.Tensor) and input.is_cuda and input.dtype == torch.float32
                 and input.dim() == 4 and weight.dim() == 4 and weight.shape[0] == 1 and weight.shape[1] == input.shape[1]
                 and weight.shape[2] * weight.shape[3] <= 16 and weight.is_cuda
                 and not (torch.is_grad_enabled() and (input.requires_grad or weight.requires_grad)))
        if mode == 'same' and isinstance(input, torch.Tensor) and input.is_cuda:       # the classification call; the 1x1 projection
            _hit("operation.conv2d[same]", fused)                                       # (mode None) is stock PyTorch by design
        if fused:
            return _filter.corr_raw(input, weight[0], out_hw=tuple(input.shape[-2:])).unsqueeze(1)
        if strict and isinstance(input, torch.Tensor) and input.is_cuda and mode == 'same':
            raise NotImplementedError("operation.conv2d(mode='same'): configuration outside the gfx950 hot path")
        return ref_single(input, weight, bias=bias, stride=stride, padding=padding, dilation=dilation, groups=groups,
                          mode=mode)

    conv2d.__doc__ = ref_single.__doc__
    lifted = tl.tensor_operation(conv2d)
    lifted.__wrapped_reference__ = ref_conv2d
    omod.conv2d = lifted


def _install_preprocessing(orig, strict):
    """`sample_patch` / `sample_patch_multiscale` (pytracking/features/preprocessing.py:33-148): for an image tensor that
    lives on the device, crop + replicate padding + bilinear resize become one gather launch.  The reference trackers
    build the image with `numpy_to_torch` on the CPU, so this route is taken when the caller uploads the frame first;
    CPU images and masks keep the reference's functions; the first-frame augmentation set (`sample_patch_transformed`) runs on
    the device for the transforms pytracking_amd/preprocessing.py covers and falls back to the reference for the rest."""
    from . import preprocessing as _pp
    try:
        pmod = importlib.import_module("pytracking.features.preprocessing")
    except Exception:
        return
    ref_sp, ref_ms, ref_tr = pmod.sample_patch, pmod.sample_patch_multiscale, pmod.sample_patch_transformed
    orig["preprocessing"] = {"functions": (ref_sp, ref_ms, ref_tr), "importers": []}

    def sample_patch(im, pos, sample_sz, output_sz=None, mode='replicate', max_scale_change=None, is_mask=False):
        if _hit("sample_patch", im.is_cuda and im.dtype == torch.float32 and not is_mask and im.dim() == 4 and im.shape[0] == 1):
            return _pp.sample_patch(im, pos, sample_sz, output_sz, mode=mode, max_scale_change=max_scale_change)
        if strict and im.is_cuda:
            raise NotImplementedError("sample_patch: call outside the gfx950 hot path")
        return ref_sp(im, pos, sample_sz, output_sz, mode=mode, max_scale_change=max_scale_change, is_mask=is_mask)

    def sample_patch_multiscale(im, pos, scales, image_sz, mode='replicate', max_scale_change=None):
        n = 1 if isinstance(scales, (int, float)) else len(scales)
        if _hit("sample_patch_multiscale", im.is_cuda and im.dtype == torch.float32 and im.dim() == 4 and im.shape[0] == 1 and n <= 8):
            return _pp.sample_patch_multiscale(im, pos, scales, image_sz, mode=mode, max_scale_change=max_scale_change)
        if strict and im.is_cuda:
            raise NotImplementedError("sample_patch_multiscale: call outside the gfx950 hot path")
        return ref_ms(im, pos, scales, image_sz, mode=mode, max_scale_change=max_scale_change)

    def sample_patch_transformed(im, pos, scale, image_sz, transforms, is_mask=False):
        # first-frame augmentation set (generate_init_samples, dimp.py:329-395): one gather launch over the base patch
        if im.is_cuda and im.dtype == torch.float32 and not is_mask and im.dim() == 4 and im.shape[0] == 1:
            try:
                out = _pp.sample_patch_transformed(im, pos, scale, image_sz, transforms)
                _hit("sample_patch_transformed", True)
                return out
            except NotImplementedError:                       # a transform the device path does not cover (RandomAffine, ...)
                if strict:
                    raise
        elif strict and im.is_cuda:
            raise NotImplementedError("sample_patch_transformed: call outside the gfx950 hot path")
        _hit("sample_patch_transformed", False)
        return ref_tr(im, pos, scale, image_sz, transforms, is_mask=is_mask)

    for fn, ref in ((sample_patch, ref_sp), (sample_patch_multiscale, ref_ms), (sample_patch_transformed, ref_tr)):
        fn.__doc__, fn.__wrapped__ = ref.__doc__, ref
    pmod.sample_patch, pmod.sample_patch_multiscale = sample_patch, sample_patch_multiscale
    pmod.sample_patch_transformed = sample_patch_transformed
    # the trackers import the names (`from pytracking.features.preprocessing import sample_patch_multiscale, ...`)
    # ... and ATOM reaches them through its feature extractors (`pytracking/features/extractor.py:3,112`)
    for modname in ("pytracking.tracker.dimp.dimp", "pytracking.tracker.atom.atom", "pytracking.tracker.tomp.tomp",
                    "pytracking.tracker.kys.kys", "pytracking.tracker.lwl.lwl", "pytracking.features.extractor",
                    "pytracking.features.featurebase"):
        tmod = sys.modules.get(modname)
        if tmod is None:
            try:
                tmod = importlib.import_module(modname)
            except Exception:
                continue
        for name, fn, ref in (("sample_patch", sample_patch, ref_sp), ("sample_patch_multiscale", sample_patch_multiscale, ref_ms),
                              ("sample_patch_transformed", sample_patch_transformed, ref_tr)):
            if getattr(tmod, name, None) is ref:
                setattr(tmod, name, fn)
                orig["preprocessing"]["importers"].append((tmod, name, ref))


def install(strict=False, atom_cg=True, tomp=True, clf_head=True, localization=True, iou_refine=True, preprocessing=True):
    """Rebind the boundary symbols.  Call after the reference is importable (`sys.path`) and before networks or
    trackers are constructed.  Idempotent."""
    if _state["installed"]:
        return
    provide_prroi_module(_state["originals"])                     # must precede `import ltr.models...`
    fmod = importlib.import_module("ltr.models.layers.filter")
    omod = importlib.import_module("ltr.models.target_classifier.optimizer")
    orig = _state["originals"]
    orig["filter"] = (fmod.apply_filter, fmod.apply_feat_transpose, fmod.filter_gradient)
    a, t, g = _make_dispatchers(fmod, strict)
    fmod.apply_filter, fmod.apply_feat_transpose, fmod.filter_gradient = a, t, g
    # modules that did `import ltr.models.layers.filter as filter_layer` see the rebinding through the module
    # object; nothing in the hot path does `from ... import apply_filter`.
    names = ("DiMPSteepestDescentGN", "DiMPL2SteepestDescentGN", "PrDiMPSteepestDescentNewton")
    orig["optimizer"] = tuple(getattr(omod, n) for n in names)
    for n in names:
        setattr(omod, n, _optimizer_class(getattr(_optimizer, n), getattr(omod, n), strict))
    # LWL few-shot learner: the residual module is replaced outright (same parameters); the generic optimiser class
    # dispatches on it so that other residual modules (RTS, dimp_simple) keep the reference implementation
    try:
        rmod = importlib.import_module("ltr.models.lwl.loss_residual_modules")
        smod = importlib.import_module("ltr.models.meta.steepestdescent")
    except Exception:
        rmod = smod = None
    if rmod is not None:
        orig["lwl"] = (rmod.LWTLResidual, smod.GNSteepestDescent)
        ref_gn, ref_res = smod.GNSteepestDescent, rmod.LWTLResidual

        class LWTLResidual(ref_res):
            """No dilation factors -> the gfx950 mirror (same parameter); dilated filters -> the reference class."""

            def __new__(cls, init_filter_reg=1e-2, filter_dilation_factors=None):
                if filter_dilation_factors is None:
                    return _ResMirror(init_filter_reg, filter_dilation_factors)
                if strict:
                    raise NotImplementedError("LWTLResidual: dilated filters are outside the gfx950 hot path")
                return ref_res.__new__(cls)

        class _ResMirror(_sd.LWTLResidual):
            # the residual vectors themselves (only the reference's generic forward asks for them): the reference's
            # own method on the rebound filter layer, so CPU tensors and autograd behave as upstream
            forward = ref_res.forward

        class GNSteepestDescent(ref_gn):
            """LWTLResidual (gfx950 mirror) at inference -> fused solver; any other residual module, a training call
            (grad-enabled inputs) or CPU tensors -> the reference class / the reference forward."""

            def __new__(cls, residual_module=None, *args, **kw):
                if (isinstance(residual_module, _sd.LWTLResidual) and residual_module.filter_dilation_factors is None
                        and kw.get("residual_batch_dim", 0) == 1 and kw.get("parameter_batch_dim", 0) == 0):
                    return _FusedGN(residual_module, *args, **kw)
                return ref_gn.__new__(cls)

        class _FusedGN(_sd.GNSteepestDescent):
            _compute_loss = ref_gn._compute_loss
            _sqr_norm = ref_gn._sqr_norm

            def forward(self, meta_parameter, num_iter=None, *args, **kwargs):
                w = meta_parameter[0] if isinstance(meta_parameter, (list, tuple)) else meta_parameter
                feat = kwargs.get("feat")
                fused = (isinstance(feat, torch.Tensor) and feat.is_cuda and w.is_cuda and not args
                         and not (torch.is_grad_enabled() and (w.requires_grad or feat.requires_grad)))
                if _hit("GNSteepestDescent", fused):
                    return _sd.GNSteepestDescent.forward(self, meta_parameter, num_iter, **kwargs)
                if strict:
                    raise NotImplementedError("GNSteepestDescent: call outside the gfx950 hot path")
                return ref_gn.forward(self, meta_parameter, num_iter, *args, **kwargs)

        rmod.LWTLResidual = LWTLResidual
        smod.GNSteepestDescent = GNSteepestDescent
    if tomp:
        _install_tomp(orig, strict)
    if clf_head:
        _install_clf_head(orig, strict)
    if localization:
        _install_localization(orig, strict)
    if iou_refine:
        _install_iou_refine(orig, strict)
    if preprocessing:
        _install_preprocessing(orig, strict)
    if atom_cg:
        try:
            pmod = importlib.import_module("pytracking.libs.optimization")
            amod = importlib.import_module("pytracking.tracker.atom.optim")
        except Exception:          # pytracking side not importable (missing cv2 ...): the ltr side is still installed
            pmod = amod = None
        if pmod is not None:
            orig["cg"] = pmod.ConjugateGradient
            ref_cg, ref_problem = pmod.ConjugateGradient, amod.ConvProblem

            class ConjugateGradient(ref_cg):
                """ConvProblem + MLU on device -> fused gfx950 CG; any other problem -> the reference class."""

                def __new__(cls, problem, variable, *args, **kw):
                    kind = _optimization.activation_kind(getattr(problem, "response_activation", None))[0]
                    fast = (isinstance(problem, ref_problem) and kind == "mlu" and len(variable) == 1
                            and variable[0].is_cuda and variable[0].shape[-1] == variable[0].shape[-2]
                            and variable[0].shape[-1] ** 2 <= 16 and not kw.get("debug", False)
                            and kw.get("standard_alpha", True) and kw.get("cg_eps", 0.0) == 0.0)
                    if _hit("ConjugateGradient", fast):
                        return _optimization.ConjugateGradient(problem, variable, *args, **kw)
                    if strict:
                        raise NotImplementedError("ConjugateGradient: problem outside the gfx950 hot path")
                    return ref_cg.__new__(cls)

            orig["gn"] = pmod.GaussNewtonCG
            ref_gn_cg, ref_fact = pmod.GaussNewtonCG, amod.FactorizedConvProblem

            class GaussNewtonCG(ref_gn_cg):
                """FactorizedConvProblem, MLU response + identity projection activation, one feature block on device
                -> fused gfx950 joint Gauss-Newton; any other problem -> the reference class."""

                def __new__(cls, problem, variable, *args, **kw):
                    ak = _optimization.activation_kind
                    fast = (isinstance(problem, ref_fact) and len(variable) == 2 and variable[0].is_cuda
                            and ak(getattr(problem, "response_activation", None))[0] == "mlu"
                            and ak(getattr(problem, "projection_activation", None))[0] == "identity"
                            and variable[0].shape[-1] == variable[0].shape[-2] and variable[0].shape[-1] ** 2 <= 16
                            and not any(kw.get(k, False) for k in ("debug", "analyze", "plotting"))
                            and kw.get("standard_alpha", True) and kw.get("cg_eps", 0.0) == 0.0
                            and kw.get("direction_forget_factor", 0) == 0)
                    if _hit("GaussNewtonCG", fast):
                        return _optimization.GaussNewtonCG(problem, variable, *args, **kw)
                    if strict:
                        raise NotImplementedError("GaussNewtonCG: problem outside the gfx950 hot path")
                    return ref_gn_cg.__new__(cls)

            pmod.ConjugateGradient = ConjugateGradient
            pmod.GaussNewtonCG = GaussNewtonCG
            _install_operation(orig, strict)
            tmod = sys.modules.get("pytracking.tracker.atom.atom")
            if tmod is not None and hasattr(tmod, "ConjugateGradient"):
                tmod.ConjugateGradient = ConjugateGradient
            if tmod is not None and hasattr(tmod, "GaussNewtonCG"):
                tmod.GaussNewtonCG = GaussNewtonCG
    _state["installed"] = True


def uninstall():
    if not _state["installed"]:
        return
    orig = _state["originals"]
    fmod = importlib.import_module("ltr.models.layers.filter")
    omod = importlib.import_module("ltr.models.target_classifier.optimizer")
    fmod.apply_filter, fmod.apply_feat_transpose, fmod.filter_gradient = orig["filter"]
    for n, c in zip(("DiMPSteepestDescentGN", "DiMPL2SteepestDescentGN", "PrDiMPSteepestDescentNewton"), orig["optimizer"]):
        setattr(omod, n, c)
    if "cg" in orig:
        importlib.import_module("pytracking.libs.optimization").ConjugateGradient = orig["cg"]
    if "gn" in orig:
        importlib.import_module("pytracking.libs.optimization").GaussNewtonCG = orig["gn"]
    if "operation" in orig:
        importlib.import_module("pytracking.libs.operation").conv2d = orig["operation"]
    if "tomp" in orig:
        importlib.import_module("ltr.models.transformer.transformer").Transformer = orig["tomp"][0]
        importlib.import_module("ltr.models.transformer.filter_predictor").FilterPredictor = orig["tomp"][1]
        hm = importlib.import_module("ltr.models.transformer.heads")
        hm.LinearFilterClassifier, hm.DenseBoxRegressor = orig["tomp"][2], orig["tomp"][3]
    if "clf_head" in orig:
        importlib.import_module("ltr.models.target_classifier.features").residual_bottleneck = orig["clf_head"]
    if "iou_refine" in orig:
        for cls, name, ref in orig["iou_refine"]:
            setattr(cls, name, ref)
    if "localization" in orig:
        for key, ref in orig["localization"].items():
            if key == "max2d":
                importlib.import_module("pytracking.libs.dcf").max2d = ref
            else:
                getattr(importlib.import_module(key[0]), key[1]).localize_advanced = ref
    if "prroi" in orig:
        name, prev, prev_attr, importers = orig["prroi"]
        if prev is not None:
            sys.modules[name] = prev
        else:
            sys.modules.pop(name, None)
        if prev_attr is not None:
            sys.modules["ltr.external.PreciseRoIPooling.pytorch"].prroi_pool = prev_attr
        for mod, cls in importers:
            mod.PrRoIPool2D = cls
    if "preprocessing" in orig:
        pm = importlib.import_module("pytracking.features.preprocessing")
        pm.sample_patch, pm.sample_patch_multiscale, pm.sample_patch_transformed = orig["preprocessing"]["functions"]
        for tmod, name, ref in orig["preprocessing"]["importers"]:
            setattr(tmod, name, ref)
    if "lwl" in orig:
        importlib.import_module("ltr.models.lwl.loss_residual_modules").LWTLResidual = orig["lwl"][0]
        importlib.import_module("ltr.models.meta.steepestdescent").GNSteepestDescent = orig["lwl"][1]
    _state["installed"] = False
    orig.clear()
