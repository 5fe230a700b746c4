"""TEST INFRASTRUCTURE ONLY -- drive the UNMODIFIED reference tracker classes end to end and record every call that
crosses the hot-path boundary, for the trajectory-level parity tests.

    python -B oracle/make_golden.py trackers        (build container only: needs /root/reference)

What runs: `pytracking.tracker.dimp.DiMP.initialize()` + `track()` exactly as the reference ships them, with
`pytracking/parameter/dimp/dimp50.py`'s settings (cadences shortened so that a short sequence reaches every branch), on
a random-init `dimpnet50` built by the reference's own constructor (no checkpoints exist offline).  Only what
BASELINE.json's north_star leaves on stock PyTorch is stubbed: the ResNet-50 backbone and the IoU-feature convolutions
return seeded synthetic feature maps (`StubBackbone`), so the same inputs can be regenerated on the GPU box, where
neither the reference nor torchvision exists.

The recorder wraps the boundary callables ON THE INSTANCES (nothing in the reference tree is edited) and appends one
event per call:  head (classification-feature head), get_filter (initialiser + 10 SD iterations), classify, localize,
refine (IoU-guided box refinement), memory (slot written), optimize (filter_optimizer update).  tests/tracker_replay.py
replays the event list through any implementation of those ops with the solver state (filter, sample memory, head
outputs) carried forward closed-loop and the tracker's own glue (positions, scales, schedules) taken from the log.
"""
import math
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from pytracking_amd import synth  # noqa: E402


class StubBackbone:
    """Seeded stand-in for the stock-PyTorch parts in front of the hot path.  Call k of `backbone(n)` returns the maps
    drawn from `default_rng(seed + k)`; `iou_feat(n)` likewise with its own counter."""

    def __init__(self, seed, dims, device="cpu"):
        self.seed, self.d = int(seed), dict(dims)
        self.k_backbone = 0
        self.k_iou = 0
        self.device = torch.device(device)

    def backbone(self, n):
        out = synth.tracker_backbone(self.seed + self.k_backbone, n, self.d)
        self.k_backbone += 1
        return OrderedDict((k, torch.from_numpy(v).to(self.device)) for k, v in out.items())

    def iou_feat(self, n):
        out = synth.tracker_iou_feat(self.seed + 5000 + self.k_iou, n, self.d)
        self.k_iou += 1
        return [torch.from_numpy(v).to(self.device) for v in out]


def use_device(params, device):
    """`params.use_gpu` / `params.device` as pytracking/parameter/*/ sets them for a GPU run (dimp50.py:10 `use_gpu = True`;
    the trackers derive `params.device` from it, dimp.py:28-29)."""
    device = torch.device(device)
    params.use_gpu = device.type == "cuda"
    params.device = str(device) if device.type == "cuda" else "cpu"
    return params


class host_rng_for_device_tensors:
    """Device runs only.  ATOM draws its start filter and projection matrix with `Tensor.normal_()` on tensors that live on
    the tracker's device (atom.py:147,544); on a GPU that consumes the device generator, so the run would start from other
    random numbers than the recorded CPU run and leave the host generator (proposal jitter, atom.py:715) out of step.  Inside
    this context `normal_` on a device tensor draws on the host generator and copies: same numbers, same stream position."""

    def __enter__(self):
        self._orig = orig = torch.Tensor.normal_

        def normal_(t, mean=0, std=1, *, generator=None):
            if t.is_cuda and generator is None:
                t.copy_(orig(torch.empty(t.shape, dtype=t.dtype), mean, std))
                return t
            return orig(t, mean, std, generator=generator)
        torch.Tensor.normal_ = normal_
        return self

    def __exit__(self, *exc):
        torch.Tensor.normal_ = self._orig
        return False


class Recorder:
    def __init__(self):
        self.events = []
        self.notes = []                 # (event kind, implementing module) pairs: not part of the serialised log

    def add(self, kind, **kw):
        ev = {"kind": kind}
        for k, v in kw.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy().copy()
            ev[k] = v
        self.events.append(ev)

    def to_npz_dict(self):
        out = {"n_events": len(self.events)}
        for i, ev in enumerate(self.events):
            for k, v in ev.items():
                out[f"e{i}_{k}"] = np.asarray(v)
        return out


DIMP50_TEST = dict(C_backbone=1024, C_layer2=512, C=512, H=18, W=18, H2=36, W2=36, C_iou=256, K=4, base_seed=31, noise=0.3)
# the committed run: 10 frames; the not-found threshold is lowered to the score level a random-init network reaches
DIMP_RUN = dict(seed=4100, n_frames=10, dims=DIMP50_TEST, thresholds=dict(target_not_found_threshold=0.1))


# PrDiMP-50 (pytracking/parameter/dimp/prdimp50.py): 22x22 maps from 352x352 samples, Newton / KL optimiser, soft-max score
# preprocessing, box refinement in the relative parametrisation (10 iterations).  Random-init soft-max scores sit near 1 / 23^2: the
# not-found threshold is lowered to that level so that the run reaches refinement and update, the hard-negative threshold to 0.3 so that
# the 8 frames hold both outcomes (6 hard negatives: 1-iteration updates; 2 normal frames: the regular 2-iteration update).
PRDIMP50_TEST = dict(C_backbone=1024, C_layer2=512, C=512, H=22, W=22, H2=44, W2=44, C_iou=256, K=4, base_seed=43, noise=0.3)
PRDIMP_RUN = dict(seed=4700, n_frames=8, dims=PRDIMP50_TEST, variant="prdimp",
                  thresholds=dict(target_not_found_threshold=0.0021, hard_negative_threshold=0.3))


def seed_dimp_net(net, seed, dims, init=True):
    """Overwrite the hot-path parameters of a reference `dimpnet50` with the seeded values of pytracking_amd/synth.py
    (the same generators rebuild them on the GPU box)."""
    p = synth.tracker_dimp_params(seed, dims)
    with torch.no_grad():
        net.classifier.feature_extractor[0].weight.copy_(torch.from_numpy(p["head.weight"]))
        if init:
            net.classifier.filter_initializer.filter_conv.weight.copy_(torch.from_numpy(p["init.weight"]))
            net.classifier.filter_initializer.filter_conv.bias.copy_(torch.from_numpy(p["init.bias"]))
        sd = net.bb_regressor.state_dict()
        for k, v in p.items():
            if k.startswith("iou."):
                sd[k[4:]].copy_(torch.from_numpy(v))
    return p


def build_dimp50(seed, dims=DIMP50_TEST):
    """Random-init network by the reference's constructor with the deployed hyper-parameters
    (ltr/train_settings/dimp/dimp50.py:91-95), hot-path weights seeded."""
    ref_harness.install()
    import ltr.models.tracking.dimpnet as dimpnet
    torch.manual_seed(seed)
    net = dimpnet.dimpnet50(filter_size=4, backbone_pretrained=False, optim_iter=5, clf_feat_norm=True,
                            clf_feat_blocks=0, final_conv=True, out_feature_dim=dims["C"], optim_init_step=0.9,
                            optim_init_reg=0.1, init_gauss_sigma=0.9, num_dist_bins=100, bin_displacement=0.1,
                            mask_init_factor=3.0, target_mask_act='sigmoid', score_act='relu')
    net.eval()
    seed_dimp_net(net, seed, dims)
    return net


def build_prdimp50(seed, dims=PRDIMP50_TEST):
    """`klcedimpnet50` with the deployed hyper-parameters (ltr/train_settings/dimp/prdimp50.py:95-98: zero filter initialiser,
    step 1.0, reg 0.05, alpha_eps 0.05, normalised label density with sigma = output_sigma * feature_sz), head + IoU weights seeded."""
    ref_harness.install()
    import ltr.models.tracking.dimpnet as dimpnet
    torch.manual_seed(seed)
    net = dimpnet.klcedimpnet50(filter_size=4, backbone_pretrained=False, optim_iter=5, clf_feat_norm=True, clf_feat_blocks=0,
                                final_conv=True, out_feature_dim=dims["C"], optim_init_step=1.0, optim_init_reg=0.05,
                                optim_min_reg=0.05, gauss_sigma=(1 / 4 / 6.0) * dims["H"], alpha_eps=0.05, normalize_label=True,
                                init_initializer='zero')
    net.eval()
    seed_dimp_net(net, seed, dims, init=False)
    return net


def prdimp50_params(net_stub):
    """pytracking/parameter/dimp/prdimp50.py on top of the DiMP set (same cadence shortening, cv2-only / random augmentations dropped)."""
    p = dimp50_params(net_stub)
    p.image_sample_size = 22 * 16
    p.search_area_scale = 6
    p.border_mode = 'inside_major'
    p.patch_max_scale_change = 1.5
    p.score_preprocess = 'softmax'
    p.target_not_found_threshold = 0.04
    p.box_refinement_space = 'relative'
    p.box_refinement_iter = 10
    p.box_refinement_step_length = 2.5e-3
    p.box_refinement_step_decay = 1
    return p


class NetStub:
    """What `params.net` has to be for the tracker (pytracking/features/net_wrappers.py:NetWithBackbone): attribute
    access falls through to the network; `initialize()` would load a checkpoint from disk -- there is none offline."""

    def __init__(self, net, stub: StubBackbone):
        self.net, self._stub = net, stub

    def initialize(self, *a, **k):
        pass

    def extract_backbone(self, im):
        return self._stub.backbone(im.shape[0])

    def __getattr__(self, name):
        return getattr(self.__dict__["net"], name)


def dimp50_params(net_stub):
    """pytracking/parameter/dimp/dimp50.py with the cadences shortened (train_skipping 20 -> 2) and the
    cv2-only / random augmentations dropped (rotate needs OpenCV; dropout draws from torch's global RNG)."""
    from pytracking.utils import TrackerParams
    p = TrackerParams()
    p.debug = 0
    p.visualization = False
    p.use_gpu = False
    p.device = "cpu"
    p.image_sample_size = 18 * 16
    p.search_area_scale = 5
    p.sample_memory_size = 50
    p.learning_rate = 0.01
    p.init_samples_minimum_weight = 0.25
    p.train_skipping = 2
    p.update_classifier = True
    p.net_opt_iter = 10
    p.net_opt_update_iter = 2
    p.net_opt_hn_iter = 1
    p.window_output = False
    p.use_augmentation = True
    p.augmentation = {'fliplr': True, 'blur': [(3, 1), (1, 3), (2, 2)],
                      'relativeshift': [(0.6, 0.6), (-0.6, 0.6), (0.6, -0.6), (-0.6, -0.6)]}
    p.augmentation_expansion_factor = 2
    p.random_shift_factor = 1 / 3
    p.advanced_localization = True
    p.target_not_found_threshold = 0.25
    p.distractor_threshold = 0.8
    p.hard_negative_threshold = 0.5
    p.target_neighborhood_scale = 2.2
    p.dispalcement_scale = 0.8
    p.hard_negative_learning_rate = 0.02
    p.update_scale_when_uncertain = True
    p.iounet_augmentation = False
    p.iounet_use_log_scale = True
    p.iounet_k = 3
    p.num_init_random_boxes = 9
    p.box_jitter_pos = 0.1
    p.box_jitter_sz = 0.5
    p.maximal_aspect_ratio = 6
    p.box_refinement_iter = 5
    p.box_refinement_step_length = 1
    p.box_refinement_step_decay = 1
    p.net = net_stub
    p.vot_anno_conversion_type = 'preserve_area'
    return p


def synthetic_frame(rng, hw=(360, 480)):
    return rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)


def run_dimp(seed=4100, n_frames=6, dims=DIMP50_TEST, record=True, score_gain=None, thresholds=None, device="cpu", variant="dimp"):
    """initialize() + n_frames x track() of the reference DiMP on a stubbed backbone.  Returns (outputs, recorder).
    `device="cuda"`: the same unmodified tracker as the reference runs it on a GPU (`params.use_gpu = True`, network and
    features on the device)."""
    ref_harness.install()
    from pytracking.tracker.dimp.dimp import DiMP
    net = (build_prdimp50 if variant == "prdimp" else build_dimp50)(seed, dims).to(device)
    stub = StubBackbone(seed, dims, device)
    ns = NetStub(net, stub)
    params = use_device((prdimp50_params if variant == "prdimp" else dimp50_params)(ns), device)
    if thresholds:
        for k, v in thresholds.items():
            setattr(params, k, v)
    tracker = DiMP(params)
    tracker.visdom = None
    rec = Recorder()
    bbreg = net.bb_regressor
    bbreg.get_iou_feat = lambda feats: stub.iou_feat(feats[0].shape[0])      # stock convs: stubbed like the backbone

    if record:
        rec.add("config", seed=seed, n_frames=n_frames, memory_size=params.sample_memory_size,
                **{f"dim_{k}": v for k, v in dims.items()},
                **{k: getattr(params, k) for k in ("target_not_found_threshold", "distractor_threshold", "hard_negative_threshold",
                                                   "target_neighborhood_scale", "dispalcement_scale", "box_refinement_iter",
                                                   "box_refinement_step_length", "box_refinement_step_decay")})
        # ---- head
        orig_head = tracker.get_classification_features

        def head(backbone_feat):
            out = orig_head(backbone_feat)
            rec.add("head", n=out.shape[0], checksum=float(out.double().abs().sum()))
            return out
        tracker.get_classification_features = head
        # ---- initial filter
        clf = net.classifier
        orig_get_filter = clf.get_filter

        def get_filter(feat, bb, *a, **kw):
            w, its, losses = orig_get_filter(feat, bb, *a, **kw)
            rec.add("get_filter", bb=bb, num_iter=kw.get("num_iter"), filter=w, n=feat.shape[0])
            return w, its, losses
        clf.get_filter = get_filter
        # ---- filter_optimizer on update frames
        opt = clf.filter_optimizer
        orig_fwd = opt.forward
        state = {"in_get_filter": False}

        def opt_forward(weights, *a, **kw):
            out = orig_fwd(weights, *a, **kw)
            if "sample_weight" in kw and kw.get("sample_weight") is not None:      # update call (dimp.py:633-639)
                rec.add("optimize", bb=kw["bb"], sw=kw["sample_weight"], num_iter=kw["num_iter"], n=kw["feat"].shape[0],
                        filter=out[0])
            return out
        opt.forward = opt_forward
        # ---- classify
        orig_cls = tracker.classify_target

        def classify(x):
            s = orig_cls(x)
            rec.add("classify", scores=s)
            return s
        tracker.classify_target = classify
        # ---- localisation
        orig_loc = tracker.localize_advanced

        def localize(scores, sample_pos, sample_scales):
            st = dict(target_sz=tracker.target_sz.clone(), pos=tracker.pos.clone(), kernel_size=tracker.kernel_size.clone(),
                      img_support_sz=tracker.img_support_sz.clone())
            tv, scale_ind, s, flag = orig_loc(scores, sample_pos, sample_scales)
            rec.add("localize", sample_pos=sample_pos, sample_scales=sample_scales, tv=tv, scale_ind=int(scale_ind),
                    flag=flag, **st)
            return tv, scale_ind, s, flag
        tracker.localize_advanced = localize
        # ---- IoU refinement
        for name in ("optimize_boxes_default", "optimize_boxes_relative"):
            orig = getattr(tracker, name)

            def refine(iou_features, init_boxes, _orig=orig, _name=name):
                b, iou = _orig(iou_features, init_boxes)
                rec.add("refine", method=_name, init_boxes=init_boxes, boxes=b, iou=iou,
                        mod3=tracker.iou_modulation[0], mod4=tracker.iou_modulation[1])
                return b, iou
            setattr(tracker, name, refine)
        # ---- memory
        orig_mem = tracker.update_memory

        def update_memory(sample_x, target_box, learning_rate=None):
            orig_mem(sample_x, target_box, learning_rate)
            rec.add("memory", slot=int(tracker.previous_replace_ind[0]), stored=int(tracker.num_stored_samples[0]))
        tracker.update_memory = update_memory

    rng = np.random.default_rng(seed + 77)
    torch.manual_seed(seed)                                    # the tracker draws proposal jitter from torch's RNG
    outs = []
    box = [200.0, 140.0, 70.0, 90.0]
    tracker.initialize(synthetic_frame(rng), {"init_bbox": box})
    opt_mod = net.classifier.filter_optimizer
    if record and variant == "prdimp":
        rec.add("optimizer_params", log_step_length=opt_mod.log_step_length.detach(), filter_reg=opt_mod.filter_reg.detach(),
                min_filter_reg=float(opt_mod.min_filter_reg), gauss_sigma=float(opt_mod.gauss_sigma), alpha_eps=float(opt_mod.alpha_eps))
    elif record:
        rec.add("optimizer_params", log_step_length=opt_mod.log_step_length.detach(), filter_reg=opt_mod.filter_reg.detach(),
                min_filter_reg=float(opt_mod.min_filter_reg), label_lut=opt_mod.label_map_predictor.weight.detach().reshape(-1),
                mask_lut=opt_mod.target_mask_predictor[0].weight.detach().reshape(-1),
                spatial_lut=opt_mod.spatial_weight_predictor.weight.detach().reshape(-1))
    for _ in range(n_frames):
        if record:
            rec.add("frame", frame=tracker.frame_num + 1)
        out = tracker.track(synthetic_frame(rng))
        outs.append(np.array(out["target_bbox"], dtype=np.float64))
        if record:
            rec.add("state", target_bbox=outs[-1], flag=str(tracker.debug_info.get("flag", "")))
    return np.stack(outs), rec, (tracker, net)


class RefOps:
    """The replay ops (tests/tracker_replay.py) served by the REFERENCE's own modules on CPU: replaying a log through
    them must reproduce it bit for bit, which validates the log and the player before the gfx950 path is judged by them."""

    def __init__(self, net, tracker):
        self.net, self.tracker = net, tracker

    def to_numpy(self, t):
        return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)

    def head(self, l3):
        with torch.no_grad():
            return self.net.classifier.extract_classification_feat(torch.from_numpy(l3))

    def get_filter(self, feat, bb, num_iter):
        from ltr.models.target_classifier.linear_filter import LinearFilter
        with torch.no_grad():
            return LinearFilter.get_filter(self.net.classifier, feat, torch.from_numpy(bb), num_iter=num_iter,
                                           compute_losses=False)[0]

    def set_optimizer_params(self, ev):
        pass

    def new_memory(self, size, head):
        mem = head.new_zeros(size, *head.shape[1:])
        mem[:head.shape[0]] = head
        return mem

    def store(self, mem, slot, head):
        mem[slot:slot + 1] = head

    def memory_view(self, mem, n):
        return mem[:n]

    def optimize(self, filt, feat, bb, sw, num_iter):
        with torch.no_grad():
            return type(self.net.classifier.filter_optimizer).forward(
                self.net.classifier.filter_optimizer, filt, num_iter=num_iter, feat=feat, bb=torch.from_numpy(bb),
                sample_weight=torch.from_numpy(sw), compute_losses=False)[0]

    def classify(self, filt, feat):
        with torch.no_grad():
            return self.net.classifier.classify(filt, feat)

    def localize(self, scores, ev, cfg):
        import types
        from pytracking.tracker.dimp.dimp import DiMP
        me = types.SimpleNamespace(params=self.tracker.params, kernel_size=torch.from_numpy(ev["kernel_size"]),
                                   output_window=None, img_support_sz=torch.from_numpy(ev["img_support_sz"]),
                                   target_sz=torch.from_numpy(ev["target_sz"]), pos=torch.from_numpy(ev["pos"]))
        tv, scale_ind, _, flag = DiMP.localize_advanced(me, scores.squeeze(1).clone(), torch.from_numpy(ev["sample_pos"]),
                                                        torch.from_numpy(ev["sample_scales"]))
        return tv, scale_ind, flag

    def refine(self, method, feats, mods, init_boxes, cfg):
        import types
        from pytracking.tracker.dimp.dimp import DiMP
        me = types.SimpleNamespace(params=self.tracker.params, net=self.net,
                                   iou_modulation=[torch.from_numpy(m) for m in mods])
        return getattr(DiMP, method)(me, [torch.from_numpy(f) for f in feats], torch.from_numpy(init_boxes))


# ------------------------------------------------------------------------------------------------------
# ToMP (pytracking/tracker/tomp/tomp.py): every frame predicts the filters with the transformer model predictor
# ------------------------------------------------------------------------------------------------------
TOMP50_TEST = dict(C_backbone=1024, C_layer2=512, C=256, H=18, W=18, H2=36, W2=36, C_iou=256, K=1, base_seed=37, noise=0.3)
TOMP_RUN = dict(seed=4300, n_frames=6, dims=TOMP50_TEST,
                thresholds=dict(conf_ths=0.5))   # random-init scores: let the memory update run


def build_tomp50(seed, dims=TOMP50_TEST):
    """tompnet50 as ltr/train_settings/tomp/tomp50.py builds it (filter size 1, 256 channels, 6 + 6 layers), hot-path
    parameters seeded (synth.tracker_tomp_params)."""
    ref_harness.install()
    import ltr.models.tracking.tompnet as tompnet
    torch.manual_seed(seed)
    cfg = synth.TOMP
    net = tompnet.tompnet50(filter_size=1, backbone_pretrained=False, head_feat_blocks=0, head_feat_norm=True, final_conv=True,
                            out_feature_dim=dims["C"], feature_sz=cfg["feature_sz"], nhead=cfg["nhead"],
                            num_encoder_layers=cfg["n_enc"], num_decoder_layers=cfg["n_dec"], dim_feedforward=cfg["ff"],
                            use_test_frame_encoding=True)
    net.eval()
    p = synth.tracker_tomp_params(seed, dims)
    with torch.no_grad():
        net.head.feature_extractor[0].weight.copy_(torch.from_numpy(p["head.weight"]))
    for mod, pre in ((net.head.filter_predictor, "fp."), (net.head.classifier, "cls."), (net.head.bb_regressor, "reg.")):
        sd = {k[len(pre):]: torch.from_numpy(v.copy()) for k, v in p.items() if k.startswith(pre)}
        if pre == "fp.":
            sd["query_embed_fg_decoder.weight"] = sd["query_embed_fg.weight"]
            for idx in (1, 4):
                sd[f"box_encoding.{idx}.num_batches_tracked"] = torch.tensor(0)
        mod.load_state_dict(sd, strict=True)
    return net


def tomp50_params(net_stub):
    """pytracking/parameter/tomp/tomp50.py."""
    from pytracking.utils import TrackerParams
    p = TrackerParams()
    p.debug = 0
    p.visualization = False
    p.use_gpu = False
    p.device = "cpu"
    p.train_feature_size = 18
    p.feature_stride = 16
    p.image_sample_size = p.train_feature_size * p.feature_stride
    p.search_area_scale = 5
    p.border_mode = 'inside_major'
    p.patch_max_scale_change = 1.5
    p.sample_memory_size = 2
    p.learning_rate = 0.01
    p.init_samples_minimum_weight = 0.25
    p.train_skipping = 20
    p.update_classifier = True
    p.net_opt_iter = 10
    p.net_opt_update_iter = 2
    p.net_opt_hn_iter = 1
    p.window_output = False
    p.use_augmentation = False
    p.augmentation = {}
    p.augmentation_expansion_factor = 2
    p.random_shift_factor = 1 / 3
    p.advanced_localization = True
    p.target_not_found_threshold = 0.25
    p.distractor_threshold = 0.8
    p.hard_negative_threshold = 0.5
    p.target_neighborhood_scale = 2.2
    p.dispalcement_scale = 0.8
    p.hard_negative_learning_rate = 0.02
    p.update_scale_when_uncertain = True
    p.conf_ths = 0.9
    p.search_area_rescaling_at_occlusion = True
    p.net = net_stub
    p.vot_anno_conversion_type = 'preserve_area'
    return p


def run_tomp(seed=4300, n_frames=6, dims=TOMP50_TEST, thresholds=None, device="cpu"):
    """initialize() + n_frames x track() of the reference ToMP on a stubbed backbone; one event per classify_target call
    (head features of the test frame and of the memory frames, filter prediction, classifier, box regressor)."""
    ref_harness.install()
    from pytracking.tracker.tomp.tomp import ToMP
    net = build_tomp50(seed, dims).to(device)
    stub = StubBackbone(seed, dims, device)
    ns = NetStub(net, stub)
    params = use_device(tomp50_params(ns), device)
    for k, v in (thresholds or {}).items():
        setattr(params, k, v)
    tracker = ToMP(params)
    tracker.visdom = None
    rec = Recorder()
    rec.add("config", seed=seed, n_frames=n_frames, memory_size=params.sample_memory_size,
            **{f"dim_{k}": v for k, v in dims.items()},
            **{k: getattr(params, k) for k in ("target_not_found_threshold", "distractor_threshold", "hard_negative_threshold",
                                               "target_neighborhood_scale", "dispalcement_scale")})
    slots = {}                                                  # memory slot -> index of the backbone call that filled it
    orig_cls = tracker.classify_target

    def classify(sample_x):
        n = min(int(tracker.num_stored_samples[0]), int(params.sample_memory_size))   # the slices `[:num_stored]` clamp
        scores, bbox = orig_cls(sample_x)
        rec.add("tomp_classify", n=n, test_call=stub.k_backbone - 1, train_calls=np.array([slots[i] for i in range(n)]),
                labels=tracker.target_labels[0][:n], ltrb=tracker.encode_bbox(tracker.target_boxes[:n]),
                num_gth_frames=int(tracker.num_gth_frames), scores=scores, bbox=bbox)
        return scores, bbox
    tracker.classify_target = classify
    orig_loc = tracker.localize_advanced

    def localize(scores, sample_pos, sample_scales):
        st = dict(target_sz=tracker.target_sz.clone(), pos=tracker.pos.clone(), kernel_size=tracker.kernel_size.clone(),
                  img_support_sz=tracker.img_support_sz.clone())
        tv, scale_ind, s, flag, loc = orig_loc(scores, sample_pos, sample_scales)
        rec.add("tomp_localize", sample_pos=sample_pos, sample_scales=sample_scales, tv=tv, scale_ind=int(scale_ind), flag=flag,
                score_loc=loc, **st)
        return tv, scale_ind, s, flag, loc
    tracker.localize_advanced = localize
    orig_mem = tracker.update_memory

    def update_memory(sample_x, sample_y, target_box, learning_rate=None):
        orig_mem(sample_x, sample_y, target_box, learning_rate)
        slots[int(tracker.previous_replace_ind[0])] = stub.k_backbone - 1
        rec.add("memory", slot=int(tracker.previous_replace_ind[0]), stored=int(tracker.num_stored_samples[0]))
    tracker.update_memory = update_memory

    rng = np.random.default_rng(seed + 77)
    torch.manual_seed(seed)
    tracker.initialize(synthetic_frame(rng), {"init_bbox": [200.0, 140.0, 70.0, 90.0]})
    slots[0] = 0                                                # the first-frame sample (no augmentation: one sample)
    outs = []
    for _ in range(n_frames):
        rec.add("frame", frame=tracker.frame_num + 1)
        out = tracker.track(synthetic_frame(rng))
        outs.append(np.array(out["target_bbox"], dtype=np.float64))
        rec.add("state", target_bbox=outs[-1])
    return np.stack(outs), rec, (tracker, net)


class TompRefOps:
    """tests/tracker_replay.py `replay_tomp` served by the reference's own modules."""

    def __init__(self, net, tracker):
        self.net, self.tracker = net, tracker

    def to_numpy(self, t):
        return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)

    def classify(self, test_l3, train_l3, labels, ltrb, num_gth_frames):
        T = torch.from_numpy
        h = self.net.head
        with torch.no_grad():
            test_feat = h.extract_head_feat(T(test_l3))
            train_feat = h.extract_head_feat(T(train_l3))
            cw, bw, cenc, benc = h.get_filter_and_features_in_parallel(train_feat, test_feat, num_gth_frames=num_gth_frames,
                                                                       train_label=T(labels), train_ltrb_target=T(ltrb))
            return h.classifier(cenc, cw), h.bb_regressor(benc, bw)

    def localize(self, scores, ev, cfg):
        import types
        from pytracking.tracker.tomp.tomp import ToMP
        me = types.SimpleNamespace(params=self.tracker.params, kernel_size=torch.from_numpy(ev["kernel_size"]),
                                   output_window=None, img_support_sz=torch.from_numpy(ev["img_support_sz"]),
                                   target_sz=torch.from_numpy(ev["target_sz"]), pos=torch.from_numpy(ev["pos"]))
        tv, scale_ind, _, flag, loc = ToMP.localize_advanced(me, scores.squeeze(1).clone(), torch.from_numpy(ev["sample_pos"]),
                                                             torch.from_numpy(ev["sample_scales"]))
        return tv, scale_ind, flag, loc


# ------------------------------------------------------------------------------------------------------
# ATOM (pytracking/tracker/atom/atom.py): first-frame joint Gauss-Newton (filter + projection matrix), per-frame
# classification with the compressed features, IoU-guided refinement with backtracking, CG update of the filter
# ------------------------------------------------------------------------------------------------------
ATOM18_TEST = dict(C_backbone=256, C_layer2=128, C=64, H=18, W=18, H2=36, W2=36, C_iou=256, K=4, base_seed=41, noise=0.3)
ATOM_RUN = dict(seed=4500, n_frames=8, dims=ATOM18_TEST, thresholds=dict(target_not_found_threshold=-1e9, train_skipping=2))


def atom_feature_maps(seed, k, n, dims):
    """Raw backbone maps of extract call k (n patches), as the stub feature returns them BEFORE the extractor's own
    normalisation (featurebase.py:104-108, normalize_power = 2)."""
    return synth.tracker_backbone(seed + k, n, dims)


def atom_normalize(feat):
    """`MultiFeatureBase.get_feature` normalisation with normalize_power = 2 (featurebase.py:104-108) -- stock torch."""
    return feat / (torch.sum(feat.abs().view(feat.shape[0], 1, 1, -1) ** 2, dim=3, keepdim=True) /
                   (feat.shape[1] * feat.shape[2] * feat.shape[3]) + 1e-10) ** (1 / 2)


def build_atom_iounet(seed, dims=ATOM18_TEST):
    ref_harness.install()
    from ltr.models.bbreg.atom_iou_net import AtomIoUNet
    torch.manual_seed(seed)
    net = AtomIoUNet(input_dim=(dims["C_layer2"], dims["C_backbone"]), pred_input_dim=(dims["C_iou"], dims["C_iou"]),
                     pred_inter_dim=(dims["C_iou"], dims["C_iou"]))
    net.eval()
    sd = net.state_dict()
    with torch.no_grad():
        for k, v in synth.iou_net_params(seed + 901, dict(C=dims["C_iou"], I=dims["C_iou"])).items():
            sd[k].copy_(torch.from_numpy(v))
    return net


def atom_params(features, thresholds=None):
    """pytracking/parameter/atom/default.py, CPU, without the cv2-only / random augmentations ('rotate' needs OpenCV, 'dropout'
    draws from torch's global RNG inside the feature path)."""
    from pytracking.utils import TrackerParams
    p = TrackerParams()
    p.debug = 0
    p.visualization = False
    p.use_gpu = False
    p.device = "cpu"
    p.max_image_sample_size = (18 * 16) ** 2
    p.min_image_sample_size = (18 * 16) ** 2
    p.search_area_scale = 5
    p.feature_size_odd = False
    p.CG_iter = 5
    p.init_CG_iter = 60
    p.init_GN_iter = 6
    p.post_init_CG_iter = 0
    p.fletcher_reeves = False
    p.standard_alpha = True
    p.CG_forgetting_rate = None
    p.sample_memory_size = 250
    p.train_skipping = 10
    p.feature_window = False
    p.window_output = False
    p.scale_factors = torch.ones(1)
    p.score_upsample_factor = 1
    p.augmentation = {'fliplr': True, 'blur': [(2, 0.2), (0.2, 2), (3, 1), (1, 3), (2, 2)],
                      'relativeshift': [(0.6, 0.6), (-0.6, 0.6), (0.6, -0.6), (-0.6, -0.6)]}
    p.augmentation_expansion_factor = 2
    p.random_shift_factor = 1 / 3
    p.update_projection_matrix = True
    p.proj_init_method = 'randn'
    p.filter_init_method = 'randn'
    p.projection_activation = 'none'
    p.response_activation = ('mlu', 0.05)
    p.advanced_localization = True
    p.target_not_found_threshold = 0.25
    p.distractor_threshold = 0.8
    p.hard_negative_threshold = 0.5
    p.target_neighborhood_scale = 2.2
    p.dispalcement_scale = 0.8
    p.hard_negative_learning_rate = 0.02
    p.hard_negative_CG_iter = 5
    p.update_scale_when_uncertain = True
    p.use_iou_net = True
    p.iounet_augmentation = False
    p.iounet_k = 3
    p.num_init_random_boxes = 9
    p.box_jitter_pos = 0.1
    p.box_jitter_sz = 0.5
    p.maximal_aspect_ratio = 6
    p.box_refinement_iter = 5
    p.box_refinement_step_length = 1
    p.box_refinement_step_decay = 1
    p.features = features
    for k, v in (thresholds or {}).items():
        setattr(p, k, v)
    return p


def run_atom(seed=4500, n_frames=8, dims=ATOM18_TEST, thresholds=None, device="cpu"):
    """initialize() + n_frames x track() of the reference ATOM with a stub deep feature (seeded layer2 / layer3 maps in place of the
    ResNet-18, seeded IoU features in place of the IoU net's stock convolutions).  Events: atom_gn (first-frame joint optimisation),
    atom_classify, atom_localize, atom_refine, atom_memory, atom_cg."""
    ref_harness.install()
    from pytracking import TensorList
    from pytracking.utils import TrackerParams, FeatureParams
    from pytracking.features.featurebase import MultiFeatureBase
    from pytracking.features.extractor import MultiResolutionExtractor
    import pytracking.tracker.atom.atom as atom_mod
    from pytracking.libs import optimization as ropt
    rec = Recorder()
    iounet = build_atom_iounet(seed, dims).to(device)
    calls = {"k": 0, "log": []}
    dv = lambda a: torch.from_numpy(a).to(device)

    class StubAtomFeature(MultiFeatureBase):
        """pytracking/features/deep.py:ATOMResNet18 with the networks replaced by seeded maps."""

        def initialize(self):
            self.pool_stride = [1]
            self.iou_predictor = iounet

        def dim(self):
            return TensorList([dims["C_backbone"]])

        def stride(self):
            return TensorList([16])

        def extract(self, im):
            n = im.shape[0]
            k = calls["k"]
            calls["k"] += 1
            calls["log"].append((k, n))
            m = atom_feature_maps(seed, k, n, dims)
            l2, l3 = dv(m["layer2"]), dv(m["layer3"])
            self.iounet_backbone_features = TensorList([l2.clone(), l3.clone()])
            f3, f4 = synth.tracker_iou_feat(seed + 5000 + k, n, dims)
            self.iounet_features = TensorList([dv(f3), dv(f4)])
            return TensorList([l3])

    deep_params = TrackerParams()
    deep_params.learning_rate = 0.01
    deep_params.init_samples_minimum_weight = 0.25
    deep_params.output_sigma_factor = 1 / 4
    deep_params.kernel_size = (4, 4)
    deep_params.compressed_dim = dims["C"]
    deep_params.filter_reg = 1e-1
    deep_params.projection_reg = 1e-4
    deep_params.use_augmentation = True
    feat = StubAtomFeature(fparams=FeatureParams(feature_params=[deep_params]), normalize_power=2)
    params = use_device(atom_params(MultiResolutionExtractor([feat]), thresholds), device)

    # recording optimisers: the tracker module's own names (atom.py:8 `from pytracking.libs.optimization import ...`) are
    # rebound on the imported module to factories that build WHATEVER class is bound in pytracking.libs.optimization at that
    # moment (the reference's, or the one pytracking_amd.install() put there) and wrap `.run` on the instance; nothing in the
    # tree is edited
    def RecGN(problem, variable, *a, **kw):
        opt = ropt.GaussNewtonCG(problem, variable, *a, **kw)
        opt_run = opt.run

        def run(num_cg_iter, num_gn_iter=None):
            f0, P0 = opt.x[0].clone(), opt.x[1].clone()
            out = opt_run(num_cg_iter, num_gn_iter)
            rec.add("atom_gn", num_cg_iter=num_cg_iter, num_gn_iter=-1 if num_gn_iter is None else num_gn_iter, filter0=f0, proj0=P0,
                    filter=opt.x[0], proj=opt.x[1], init_call=calls["log"][-1][0], n_aug=calls["log"][-1][1],
                    y=opt.problem.y[0], sw=opt.problem.sample_weights[0], filter_reg=opt.problem.filter_reg[0],
                    projection_reg=opt.problem.projection_reg[0])
            rec.notes.append(("atom_gn", type(opt).__module__))
            return out
        opt.run = run
        return opt

    def RecCG(problem, variable, *a, **kw):
        opt = ropt.ConjugateGradient(problem, variable, *a, **kw)
        opt_run = opt.run

        def run(num_cg_iter):
            out = opt_run(num_cg_iter)
            if num_cg_iter > 0:
                rec.add("atom_cg", num_iter=num_cg_iter, filter=opt.x[0], sw=opt.problem.sample_weights[0].clone())
                rec.notes.append(("atom_cg", type(opt).__module__))
            return out
        opt.run = run
        return opt
    orig_gn, orig_cg = atom_mod.GaussNewtonCG, atom_mod.ConjugateGradient
    atom_mod.GaussNewtonCG, atom_mod.ConjugateGradient = RecGN, RecCG
    import contextlib
    rng_ctx = host_rng_for_device_tensors() if torch.device(device).type == "cuda" else contextlib.nullcontext()
    try:
        tracker = atom_mod.ATOM(params)
        tracker.visdom = None
        orig_apply = tracker.apply_filter

        def apply_filter(sample_x):
            s = orig_apply(sample_x)
            rec.add("atom_classify", test_call=calls["k"] - 1, scores=s[0])
            return s
        tracker.apply_filter = apply_filter
        orig_loc = tracker.localize_target

        def localize_target(scores_raw):
            tv, scale_ind, s, flag = orig_loc(scores_raw)
            rec.add("atom_localize", tv=tv, scale_ind=int(scale_ind), flag=str(flag))
            if isinstance(scale_ind, torch.Tensor) and scale_ind.is_cuda:
                # torch >= 2 shim, device runs only (same class as the torch.rfft one in ref_harness.py): atom.py:252 indexes the
                # HOST tensor `sample_scales` with this index, which torch 1.x accepted for a 0-dim device tensor and torch 2
                # rejects ("indices should be either on cpu or on the same device"); the reference without install() stops at
                # the same line on this image
                scale_ind = scale_ind.cpu()
            return tv, scale_ind, s, flag
        tracker.localize_target = localize_target
        orig_ob = tracker.optimize_boxes

        def optimize_boxes(iou_features, init_boxes):
            b, iou = orig_ob(iou_features, init_boxes)
            rec.add("atom_refine", init_boxes=init_boxes, boxes=b, iou=iou, iou_call=calls["k"] - 1,
                    mod3=tracker.target_feat[0], mod4=tracker.target_feat[1])
            return b, iou
        tracker.optimize_boxes = optimize_boxes
        orig_mem = tracker.update_memory

        def update_memory(sample_x, sample_y, learning_rate=None):
            orig_mem(sample_x, sample_y, learning_rate)
            rec.add("atom_memory", slot=int(tracker.previous_replace_ind[0]), stored=int(tracker.num_stored_samples[0]), y=sample_y[0],
                    test_call=calls["k"] - 1)
        tracker.update_memory = update_memory

        rng = np.random.default_rng(seed + 77)
        torch.manual_seed(seed)
        rec.add("config", seed=seed, n_frames=n_frames, memory_size=params.sample_memory_size,
                **{f"dim_{k}": v for k, v in dims.items()}, filter_reg=1e-1, act_min_val=0.05, CG_iter=params.CG_iter,
                box_refinement_iter=params.box_refinement_iter, box_refinement_step_length=params.box_refinement_step_length,
                box_refinement_step_decay=params.box_refinement_step_decay)
        with rng_ctx:
            tracker.initialize(synthetic_frame(rng), {"init_bbox": [200.0, 140.0, 70.0, 90.0]})
        rec.add("atom_init_done", n_init=int(tracker.num_init_samples[0]), sw=tracker.sample_weights[0].clone(),
                y_init=tracker.y[0][:int(tracker.num_init_samples[0])].clone())
        outs = []
        for _ in range(n_frames):
            rec.add("frame", frame=tracker.frame_num + 1)
            out = tracker.track(synthetic_frame(rng))
            outs.append(np.array(out["target_bbox"], dtype=np.float64))
            rec.add("state", target_bbox=outs[-1], flag=str(tracker.debug_info.get("flag", "")))
    finally:
        atom_mod.GaussNewtonCG, atom_mod.ConjugateGradient = orig_gn, orig_cg
    return np.stack(outs), rec, (tracker, iounet)


class AtomRefOps:
    """tests/tracker_replay.py `replay_atom` served by the reference's own classes on CPU (validates the log and the player)."""

    def __init__(self, iounet, params):
        self.iounet, self.params = iounet, params

    def to_numpy(self, t):
        return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)

    def normalize(self, raw):
        return atom_normalize(torch.from_numpy(raw))

    def gn(self, f0, P0, raw, y, sw, num_cg, num_gn, filter_reg, projection_reg):
        import torch.nn.functional as F
        from pytracking import TensorList
        from pytracking.libs import optimization as ropt
        from pytracking.tracker.atom.optim import FactorizedConvProblem
        T = torch.from_numpy
        filt, proj = TensorList([T(f0).clone()]), TensorList([T(P0).clone()])
        prob = FactorizedConvProblem(TensorList([raw]), TensorList([T(y)]), TensorList([filter_reg]), TensorList([projection_reg]),
                                     self.params, TensorList([T(sw)]), lambda x: x, lambda x: F.elu(F.leaky_relu(x, 1 / 0.05), 0.05))
        opt = ropt.GaussNewtonCG(prob, filt.concat(proj))
        if num_gn < 0:
            opt.run(num_cg)
        else:
            opt.run(num_cg, num_gn)
        return opt.x[0], opt.x[1]

    def project(self, raw, proj):
        return torch.nn.functional.conv2d(raw, proj)

    def new_memory(self, size, init_x, y_init, sw):
        mem = init_x.new_zeros(size, *init_x.shape[1:])
        mem[:init_x.shape[0]] = init_x
        y = init_x.new_zeros(size, 1, *init_x.shape[2:])
        y[:init_x.shape[0]] = torch.from_numpy(y_init)
        return mem, y, torch.from_numpy(sw).clone()

    def cg_new(self, memory, y, sw, filt, filter_reg, act_min):
        import torch.nn.functional as F
        from pytracking import TensorList
        from pytracking.libs import optimization as ropt
        from pytracking.tracker.atom.optim import ConvProblem
        x = TensorList([filt.clone()])
        prob = ConvProblem(TensorList([memory]), TensorList([y]), TensorList([filter_reg]), TensorList([sw]),
                           lambda s: F.elu(F.leaky_relu(s, 1 / act_min), act_min))
        return (ropt.ConjugateGradient(prob, x, fletcher_reeves=False, direction_forget_factor=0), x)

    def current_filter(self, cg):
        return cg[1][0]

    def cg_run(self, cg, num_iter):
        cg[0].run(num_iter)
        return cg[1][0]

    def classify(self, filt, x):
        from pytracking import TensorList
        from pytracking.libs import operation
        return operation.conv2d(TensorList([x]), TensorList([filt]), mode='same')[0]

    def store(self, memory, mem_y, slot, x, y):
        memory[slot:slot + 1] = x
        mem_y[slot:slot + 1] = torch.from_numpy(y)

    def set_weights(self, mem_sw, sw):
        mem_sw.copy_(torch.from_numpy(sw))

    def refine(self, feats, mods, init_boxes, cfg):
        import types
        from pytracking import TensorList
        from pytracking.tracker.atom.atom import ATOM
        me = types.SimpleNamespace(params=self.params, iou_predictor=self.iounet,
                                   target_feat=TensorList([torch.from_numpy(m) for m in mods]))
        return ATOM.optimize_boxes(me, TensorList([torch.from_numpy(f) for f in feats]), torch.from_numpy(init_boxes))


# ------------------------------------------------------------------------------------------------------
# LWL (pytracking/tracker/lwl/lwl.py): video object segmentation; the few-shot learner (GNSteepestDescent on LWTLResidual, 16 filters
# 3x3 over 512 x 30 x 52 maps) runs 20 iterations on the first frame and 3 on every later one over the growing memory
# ------------------------------------------------------------------------------------------------------
LWL_TEST = dict(C_layer1=256, C_layer2=512, C_backbone=1024, C_layer4=2048, C=512, H=30, W=52, H2=60, W2=104, base_seed=47, noise=0.3)
LWL_RUN = dict(seed=4900, n_frames=6, dims=LWL_TEST)


def build_lwl(seed, dims=LWL_TEST):
    """`steepest_descent_resnet50` as ltr/train_settings/lwl/lwl_stage2.py:94-102 builds it (3x3 filters, 16 of them, label encoder
    (16, 32, 64) without BatchNorm, no residual blocks + final conv in the target-model feature extractor); the 'imagenet' backbone
    variant is constructed instead of 'mrcnn' -- the backbone is stubbed and never runs.  Random init under `torch.manual_seed`; the
    decoder's and label encoder's BatchNorm / conv parameters are whatever that seed draws (stock modules on both sides)."""
    ref_harness.install()
    import ltr.models.lwl.lwl_net as lwl_net
    torch.manual_seed(seed)
    net = lwl_net.steepest_descent_resnet50(filter_size=3, num_filters=16, optim_iter=5, backbone_pretrained=False, out_feature_dim=dims["C"],
                                            label_encoder_dims=(16, 32, 64), use_bn_in_label_enc=False, clf_feat_blocks=0, final_conv=True,
                                            backbone_type='imagenet')
    net.eval()
    return net


def lwl_params(net_stub):
    """pytracking/parameter/lwl/lwl_ytvos.py."""
    from pytracking.utils import TrackerParams
    p = TrackerParams()
    p.debug = 0
    p.visualization = False
    p.seg_to_bb_mode = 'var'
    p.max_scale_change = (0.95, 1.1)
    p.min_mask_area = 100
    p.use_gpu = False
    p.device = "cpu"
    p.image_sample_size = (30 * 16, 52 * 16)
    p.search_area_scale = 5.0
    p.border_mode = 'inside_major'
    p.patch_max_scale_change = None
    p.sample_memory_size = 32
    p.learning_rate = 0.1
    p.init_samples_minimum_weight = 0.25
    p.train_skipping = 1
    p.update_target_model = True
    p.net_opt_iter = 20
    p.net_opt_update_iter = 3
    p.net = net_stub
    return p


def run_lwl(seed=4900, n_frames=4, dims=LWL_TEST, device="cpu"):
    """initialize() (with a box-shaped first-frame mask) + n_frames x track() of the reference LWL tracker on a stubbed backbone.
    Events: lwl_init (the 20-iteration target model), lwl_segment (mask encoding = the multi-filter apply_filter, and the decoder's
    raw scores), lwl_update (the 3-iteration few-shot learner over the memory), state."""
    ref_harness.install()
    from pytracking.tracker.lwl.lwl import LWL
    net = build_lwl(seed, dims).to(device)
    stub = StubBackbone(seed, dims, device)
    ns = NetStub(net, stub)
    params = use_device(lwl_params(ns), device)
    tracker = LWL(params)
    tracker.visdom = None
    rec = Recorder()
    rec.add("config", seed=seed, n_frames=n_frames, memory_size=params.sample_memory_size, **{f"dim_{k}": v for k, v in dims.items()})
    tm = net.target_model
    orig_get_filter = tm.get_filter

    state = {"in_get_filter": False}

    def get_filter(feat, *a, **kw):
        state["in_get_filter"] = True
        try:
            w, its, losses = orig_get_filter(feat, *a, **kw)
        finally:
            state["in_get_filter"] = False
        rec.add("lwl_init", num_iter=kw.get("num_iter"), filter=w[:, :, ::4], n=feat.shape[0])      # every 4th channel: the log stays small
        rec.notes.append(("lwl_init", type(tm.filter_optimizer).__mro__[1].__module__ if len(type(tm.filter_optimizer).__mro__) > 1 else ""))
        return w, its, losses
    tm.get_filter = get_filter
    opt = tm.filter_optimizer
    orig_fwd = opt.forward

    def opt_forward(meta_parameter, *a, **kw):
        out = orig_fwd(meta_parameter, *a, **kw)
        if not state["in_get_filter"] and kw.get("feat") is not None:                 # the per-frame update (lwl.py:574-578)
            rec.add("lwl_update", num_iter=kw.get("num_iter"), n=kw["feat"].shape[0], filter=out[0][0][:, :, ::4])
            rec.notes.append(("lwl_update", type(opt).__mro__[1].__module__))
        return out
    opt.forward = opt_forward
    orig_seg = tracker.segment_target

    def segment_target(sample_tm_feat, sample_x):
        with torch.no_grad():
            scores, enc = tracker.net.segment_target(tracker.target_filter, sample_tm_feat, sample_x)
        rec.add("lwl_segment", mask_encoding=enc, scores=scores[..., ::8, ::8])                 # decoder output (stock), subsampled
        return scores
    tracker.segment_target = segment_target

    rng = np.random.default_rng(seed + 77)
    torch.manual_seed(seed)
    hw = (480, 854)
    box = [300.0, 180.0, 160.0, 120.0]                           # x, y, w, h
    init_mask = np.zeros(hw, dtype=np.float32)
    init_mask[int(box[1]):int(box[1] + box[3]), int(box[0]):int(box[0] + box[2])] = 1.0
    out = tracker.initialize(synthetic_frame(rng, hw), {"init_bbox": box, "init_mask": init_mask})
    prev = {"segmentation_raw": init_mask}
    outs = []
    for _ in range(n_frames):
        rec.add("frame", frame=tracker.frame_num + 1)
        out = tracker.track(synthetic_frame(rng, hw), {"previous_output": prev})
        prev = {"segmentation_raw": np.asarray(out["segmentation_raw"]).reshape(hw)}
        outs.append(np.array(out["target_bbox"], dtype=np.float64))
        rec.add("state", target_bbox=outs[-1], mask_area=float(np.asarray(out["segmentation"]).sum()))
    return np.stack(outs), rec, (tracker, net)
