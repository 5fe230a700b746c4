"""Seeded synthetic workloads for the hot path.

There are no checkpoints and no datasets (no network), so every test, golden vector and benchmark
draws its inputs from the generators below.  They use numpy's PCG64 `default_rng(seed)` so the same
seed reproduces the same arrays in the build container (where the goldens are generated from the
reference) and on the GPU box (where the HIP path is checked against them) without shipping the
33 MB feature tensors.

Hyper-parameters are the deployed values the reference trains/tracks with (no learned values
exist here):
  DiMP-50   ltr/train_settings/dimp/dimp50.py:91-95, pytracking/parameter/dimp/dimp50.py
  PrDiMP-50 ltr/train_settings/dimp/prdimp50.py:95-98, pytracking/parameter/dimp/prdimp50.py
  ATOM      pytracking/parameter/atom/default.py
"""
import math

import numpy as np


def clf_features(rng, n, C, H, W, K):
    """N(0,1) features, instance-L2-normalised like `InstanceL2Norm(scale=sqrt(1/(C*K*K)))`
    (ltr/models/layers/normalization.py:15-20, dimpnet.py:159): x * sqrt(C*H*W / sum x^2) * scale."""
    x = rng.standard_normal((n, C, H, W), dtype=np.float32)
    ss = (x.astype(np.float64) ** 2).sum(axis=(1, 2, 3), keepdims=True)
    scale = math.sqrt(1.0 / (C * K * K))
    return (x * np.sqrt(C * H * W / ss) * scale).astype(np.float32)


def target_boxes(rng, n, crop=288.0):
    """xywh boxes around the crop centre: w,h ~ U(40,80), centre jitter U(-8,8) px (section 8d)."""
    wh = rng.uniform(40.0, 80.0, size=(n, 2))
    ctr = crop / 2.0 + rng.uniform(-8.0, 8.0, size=(n, 2))
    return np.concatenate((ctr - wh / 2.0, wh), axis=1).astype(np.float32)


def decay_weights(n, lr=0.01):
    """Steady-state DiMP memory weights (pytracking/tracker/dimp/dimp.py:445-484): the sample
    inserted j updates ago carries lr*(1-lr)^j, normalised to sum 1."""
    w = lr * (1.0 - lr) ** np.arange(n - 1, -1, -1, dtype=np.float64)
    return (w / w.sum()).astype(np.float32)


def gauss_lut(num_bins, bin_displacement, sigma):
    """`label_map_predictor` initial weights (optimizer.py:45-54)."""
    d = np.arange(num_bins, dtype=np.float32) * np.float32(bin_displacement)
    if sigma == 0:
        g = np.zeros_like(d)
        g[0] = 1
    else:
        g = np.exp(-0.5 * (d / np.float32(sigma)) ** 2)
    return (g - g.min()).astype(np.float32)


def mask_lut(num_bins, bin_displacement, mask_init_factor, mask_act="sigmoid"):
    """`target_mask_predictor[0]` initial weights (optimizer.py:57-66)."""
    d = np.arange(num_bins, dtype=np.float32) * np.float32(bin_displacement)
    bias = 0.0 if mask_act == "sigmoid" else 0.5
    return (mask_init_factor * np.tanh(2.0 - d) + bias).astype(np.float32)


DIMP50 = dict(  # dimpnet50(...) as instantiated by train_settings/dimp/dimp50.py:91-95
    C=512, H=18, W=18, K=4, feat_stride=16, num_iter=5, memory=50,
    init_step_length=0.9, init_filter_reg=0.1, min_filter_reg=1e-3, init_gauss_sigma=0.9,
    num_dist_bins=100, bin_displacement=0.1, mask_init_factor=3.0, mask_act="sigmoid",
    score_act="relu", alpha_eps=0.0,
)

PRDIMP50 = dict(  # klcedimpnet50(...) per train_settings/dimp/prdimp50.py:95-98; 22x22 per parameter/dimp/prdimp50.py:12
    C=512, H=22, W=22, K=4, feat_stride=16, num_iter=5, memory=50,
    init_step_length=1.0, init_filter_reg=0.05, min_filter_reg=0.05, gauss_sigma=0.9,
    alpha_eps=0.05, normalize_label=True, init_uni_weight=None, label_shrink=0.0,
    softmax_reg=None, label_threshold=0.0,
)

ATOM18 = dict(  # pytracking/parameter/atom/default.py:20-21,26-28,40,44-45,58-62,74
    C=64, H=18, W=18, K=4, memory=250, cg_iter=5, filter_reg=0.1, act_min_val=0.05,
    output_sigma_factor=0.25, search_area_scale=5.0,
)


def dimp_problem(seed, n, cfg=DIMP50, with_weights=True, small=None):
    """Inputs of one `filter_optimizer` call: (w0, feat, bb, sw).  `small` overrides C/H/W for the
    oracle-sized parity cases."""
    c = dict(cfg)
    if small:
        c.update(small)
    rng = np.random.default_rng(seed)
    feat = clf_features(rng, n, c["C"], c["H"], c["W"], c["K"])
    crop = c["feat_stride"] * c["H"]
    bb = target_boxes(rng, n, crop=float(crop))
    if crop < 200:      # tiny maps: shrink the boxes with the crop
        bb = (bb * (crop / 288.0)).astype(np.float32)
    sw = decay_weights(n) if with_weights else None
    w0 = (rng.standard_normal((c["C"], c["K"], c["K"]), dtype=np.float32) *
          np.float32(0.5 * math.sqrt(1.0 / (c["C"] * c["K"] ** 2))))
    return w0, feat, bb, sw


def atom_problem(seed, n, cfg=ATOM18, small=None):
    """Inputs of one ATOM `ConjugateGradient.run` call (section 8d cfg1)."""
    c = dict(cfg)
    if small:
        c.update(small)
    rng = np.random.default_rng(seed)
    C, H, W, K = c["C"], c["H"], c["W"], c["K"]
    samples = (rng.standard_normal((n, C, H, W), dtype=np.float32) * np.float32(0.1))
    # label_function_spatial Gaussians (pytracking/libs/dcf.py:56-72 semantics): centred labels with jitter
    sigma = c["output_sigma_factor"] * H / c["search_area_scale"] * 2.0
    ctr = np.stack((H / 2.0 + rng.uniform(-2, 2, n), W / 2.0 + rng.uniform(-2, 2, n)), axis=1)
    yy = np.arange(H, dtype=np.float64).reshape(1, -1, 1)
    xx = np.arange(W, dtype=np.float64).reshape(1, 1, -1)
    y = np.exp(-0.5 * ((yy - ctr[:, 0].reshape(-1, 1, 1)) ** 2 + (xx - ctr[:, 1].reshape(-1, 1, 1)) ** 2) / sigma ** 2)
    sw = decay_weights(n)
    x0 = rng.standard_normal((C, K, K), dtype=np.float32) * np.float32(0.05)
    return x0, samples, y.astype(np.float32), sw


# ------------------------------------------------------------------------------------------------------
# ToMP transformer model predictor (BASELINE configs[3])
# ------------------------------------------------------------------------------------------------------
TOMP = dict(  # tompnet50/101 as instantiated by ltr/train_settings/tomp/tomp50.py, tomp101.py (out_feature_dim=256)
    D=256, nhead=8, ff=2048, n_enc=6, n_dec=6, H=18, W=18, feature_sz=18, n_train=2, num_gth_frames=1)
TOMP_SMALL = dict(D=128, nhead=4, ff=256, n_enc=2, n_dec=2, H=6, W=6, feature_sz=6, n_train=2, num_gth_frames=1)


def _xavier(rng, shape, fan_in, fan_out):
    a = np.float32(math.sqrt(6.0 / (fan_in + fan_out)))
    return (rng.random(shape, dtype=np.float32) * 2 - 1) * a


def tomp_params(seed, cfg=TOMP):
    """Seeded parameters of FilterPredictor ('fp.'), LinearFilterClassifier ('cls.') and DenseBoxRegressor ('reg.'),
    keyed by the reference's state_dict names (filter_predictor.py:19-39, transformer.py:152-190,
    heads.py:83-117).  Xavier-uniform matrices as `Transformer._reset_parameters` (transformer.py:85-88) draws them;
    biases, LayerNorm / GroupNorm / BatchNorm affine terms and running statistics are perturbed away from their
    initial 0/1 so that a dropped term cannot go unnoticed."""
    rng = np.random.default_rng(seed)
    D, ff = cfg["D"], cfg["ff"]
    p = {}

    def vec(n, centre=0.0, spread=0.1):
        return (np.float32(centre) + np.float32(spread) * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)

    def mha(prefix):
        p[prefix + "in_proj_weight"] = _xavier(rng, (3 * D, D), D, 3 * D)
        p[prefix + "in_proj_bias"] = vec(3 * D)
        p[prefix + "out_proj.weight"] = _xavier(rng, (D, D), D, D)
        p[prefix + "out_proj.bias"] = vec(D)

    def ffn_norms(prefix, norms):
        p[prefix + "linear1.weight"] = _xavier(rng, (ff, D), D, ff)
        p[prefix + "linear1.bias"] = vec(ff)
        p[prefix + "linear2.weight"] = _xavier(rng, (D, ff), ff, D)
        p[prefix + "linear2.bias"] = vec(D)
        for nm in norms:
            p[prefix + nm + ".weight"] = vec(D, 1.0)
            p[prefix + nm + ".bias"] = vec(D)

    for i in range(cfg["n_enc"]):
        pre = f"fp.transformer.encoder.layers.{i}."
        mha(pre + "self_attn.")
        ffn_norms(pre, ("norm1", "norm2"))
    for i in range(cfg["n_dec"]):
        pre = f"fp.transformer.decoder.layers.{i}."
        mha(pre + "self_attn.")
        mha(pre + "multihead_attn.")
        ffn_norms(pre, ("norm1", "norm2", "norm3"))
    p["fp.transformer.decoder.norm.weight"] = vec(D, 1.0)
    p["fp.transformer.decoder.norm.bias"] = vec(D)
    dims = [4, D // 4, D, D]                                        # MLP([4, d/4, d, d]) (filter_predictor.py:6-17,27)
    for li, idx in enumerate((0, 3, 6)):
        p[f"fp.box_encoding.{idx}.weight"] = _xavier(rng, (dims[li + 1], dims[li], 1), dims[li], dims[li + 1])
        p[f"fp.box_encoding.{idx}.bias"] = vec(dims[li + 1])
    for li, idx in enumerate((1, 4)):
        n = dims[li + 1]
        p[f"fp.box_encoding.{idx}.weight"] = vec(n, 1.0)
        p[f"fp.box_encoding.{idx}.bias"] = vec(n)
        p[f"fp.box_encoding.{idx}.running_mean"] = vec(n, 0.0, 0.2)
        p[f"fp.box_encoding.{idx}.running_var"] = (0.5 + rng.random(n, dtype=np.float32)).astype(np.float32)
    p["fp.query_embed_fg.weight"] = rng.standard_normal((1, D), dtype=np.float32)
    p["fp.query_embed_test.weight"] = rng.standard_normal((1, D), dtype=np.float32)
    p["cls.linear.weight"] = _xavier(rng, (D, D), D, D)
    p["cls.linear.bias"] = vec(D)
    p["reg.linear.weight"] = _xavier(rng, (D, D), D, D)
    p["reg.linear.bias"] = vec(D)
    for i in range(4):                                              # conv_layer x4 (heads.py:9-16,110-115)
        p[f"reg.tower.{3 * i}.weight"] = _xavier(rng, (D, D, 3, 3), 9 * D, 9 * D)
        p[f"reg.tower.{3 * i}.bias"] = vec(D)
        p[f"reg.tower.{3 * i + 1}.weight"] = vec(D, 1.0)
        p[f"reg.tower.{3 * i + 1}.bias"] = vec(D)
    p["reg.bbreg_layer.weight"] = _xavier(rng, (4, D, 3, 3), 9 * D, 9 * 4) * np.float32(0.3)
    p["reg.bbreg_layer.bias"] = vec(4)
    return p


def tomp_inputs(seed, cfg=TOMP):
    """Head features of the two memory frames and the test frame, Gaussian train labels and ltrb box maps
    (tomp.py:282-295, 552-570): shapes (n_train,1,D,H,W), (1,1,D,H,W), (n_train,1,H,W), (n_train,1,4,H,W)."""
    rng = np.random.default_rng(seed)
    D, H, W, n = cfg["D"], cfg["H"], cfg["W"], cfg["n_train"]
    scale = np.float32(1.0 / math.sqrt(D))
    train = rng.standard_normal((n, 1, D, H, W), dtype=np.float32) * scale * np.float32(4.0)
    test = rng.standard_normal((1, 1, D, H, W), dtype=np.float32) * scale * np.float32(4.0)
    ctr = np.stack((H / 2.0 + rng.uniform(-2, 2, n), W / 2.0 + rng.uniform(-2, 2, n)), axis=1)
    yy = np.arange(H, dtype=np.float64).reshape(1, -1, 1)
    xx = np.arange(W, dtype=np.float64).reshape(1, 1, -1)
    lab = np.exp(-0.5 * ((yy - ctr[:, 0].reshape(-1, 1, 1)) ** 2 + (xx - ctr[:, 1].reshape(-1, 1, 1)) ** 2) / 1.5 ** 2)
    half = rng.uniform(1.5, 4.0, (n, 2))
    l = (xx - (ctr[:, 1] - half[:, 1]).reshape(-1, 1, 1)) / W + 0 * yy
    r = ((ctr[:, 1] + half[:, 1]).reshape(-1, 1, 1) - xx) / W + 0 * yy
    t = (yy - (ctr[:, 0] - half[:, 0]).reshape(-1, 1, 1)) / H + 0 * xx
    b = ((ctr[:, 0] + half[:, 0]).reshape(-1, 1, 1) - yy) / H + 0 * xx
    ltrb = np.stack((l, t, r, b), axis=1)
    return (train.astype(np.float32), test.astype(np.float32), lab.astype(np.float32)[:, None],
            ltrb.astype(np.float32)[:, None])


# ------------------------------------------------------------------------------------------------------
# LWL few-shot learner and the IoU-guided refinement at deployed sizes
# ------------------------------------------------------------------------------------------------------
LWL = dict(n=32, F=16, C=512, H=30, W=52, K=3, filter_reg=0.05)   # lwl_ytvos.py:17,23; lwl_stage2.py:94-100


def lwl_problem(seed, cfg=LWL):
    """(w0, feat, label, sw) of one `GNSteepestDescent(LWTLResidual)` call: zero initial filter, per-element sample
    weights (`lwl_net` feeds a (n,1,F,H,W) weight tensor), labels in [0,1)."""
    rng = np.random.default_rng(seed)
    n, F, C, H, W, K = (cfg[k] for k in ("n", "F", "C", "H", "W", "K"))
    feat = clf_features(rng, n, C, H, W, K)
    label = rng.random((n, F, H, W), dtype=np.float32)
    sw = (np.float32(0.2) + np.float32(0.8) * rng.random((n, F, H, W), dtype=np.float32)).astype(np.float32)
    w0 = np.zeros((F, C, K, K), np.float32)
    return w0, feat, label, sw


IOU50 = dict(C=256, I=256, H3=36, W3=36, H4=18, W4=18, proposals=10)   # dimpnet50: AtomIoUNet(pred_input_dim=(256,256),
#                                                                        pred_inter_dim=(256,256)), dimpnet.py:189-190


def iou_net_params(seed, cfg=IOU50):
    """Seeded weights of the IoU predictor's test branch (atom_iou_net.py:44-49) keyed by the reference's state_dict
    names: two LinearBlocks (Linear + BatchNorm2d + ReLU) and Linear(2I, 1), magnitudes as `kaiming_normal_(fan_in)`
    (atom_iou_net.py:53-64) with the BatchNorm statistics moved off their initial 0/1."""
    rng = np.random.default_rng(seed)
    C, I = cfg["C"], cfg["I"]
    p = {}
    for name, k in (("fc3_rt", 5), ("fc4_rt", 3)):
        fan = C * k * k
        p[f"{name}.linear.weight"] = rng.standard_normal((I, fan), dtype=np.float32) * np.float32(math.sqrt(2.0 / fan))
        p[f"{name}.linear.bias"] = rng.standard_normal(I, dtype=np.float32) * np.float32(0.1)
        p[f"{name}.bn.weight"] = (np.float32(1.0) + np.float32(0.1) * rng.standard_normal(I, dtype=np.float32)).astype(np.float32)
        p[f"{name}.bn.bias"] = rng.standard_normal(I, dtype=np.float32) * np.float32(0.1)
        p[f"{name}.bn.running_mean"] = rng.standard_normal(I, dtype=np.float32) * np.float32(0.1)
        p[f"{name}.bn.running_var"] = (np.float32(0.5) + rng.random(I, dtype=np.float32)).astype(np.float32)
    p["iou_predictor.weight"] = rng.standard_normal((1, 2 * I), dtype=np.float32) * np.float32(math.sqrt(2.0 / (2 * I)))
    p["iou_predictor.bias"] = np.full(1, 0.3, np.float32)
    return p


def iou_inputs(seed, cfg=IOU50):
    """(c3, c4, mod3, mod4, boxes): IoU features of the test frame, modulation vectors, 10 jittered xywh proposals."""
    rng = np.random.default_rng(seed)
    C = cfg["C"]
    c3 = rng.standard_normal((1, C, cfg["H3"], cfg["W3"]), dtype=np.float32)
    c4 = rng.standard_normal((1, C, cfg["H4"], cfg["W4"]), dtype=np.float32)
    mod3 = (np.float32(1.0) + np.float32(0.5) * rng.standard_normal((1, C), dtype=np.float32)).astype(np.float32)
    mod4 = (np.float32(1.0) + np.float32(0.5) * rng.standard_normal((1, C), dtype=np.float32)).astype(np.float32)
    base = np.array([100.0, 90.0, 80.0, 110.0], np.float32)
    boxes = np.stack([base] + [base + np.concatenate((rng.uniform(-12, 12, 2), rng.uniform(-25, 25, 2))).astype(np.float32)
                               for _ in range(cfg["proposals"] - 1)])
    return c3, c4, mod3, mod4, boxes.astype(np.float32)


def atom_gn_problem(seed, n=30, M=256, Kc=64, H=18, W=18, K=4, cfg=ATOM18):
    """ATOM first frame (parameter/atom/default.py:27-28,40-45; atom.py:140-176): n augmented samples of M backbone
    channels, zero initial filter, PCA-like projection matrix.  Returns (f0, P0, samples, y, sw)."""
    rng = np.random.default_rng(seed)
    samples = rng.standard_normal((n, M, H, W), dtype=np.float32) * np.float32(0.1)
    _, _, y, sw = atom_problem(seed, n, cfg, small=dict(C=Kc, H=H, W=W))
    f0 = np.zeros((Kc, K, K), np.float32)
    P0 = rng.standard_normal((Kc, M), dtype=np.float32) * np.float32(1.0 / math.sqrt(M))
    return f0, P0, samples, y, sw


# ------------------------------------------------------------------------------------------------------
# Tracker-level replay (tests/tracker_replay.py): what stays on stock PyTorch in front of the hot path is synthetic
# ------------------------------------------------------------------------------------------------------
def tracker_backbone(seed, n, dims):
    """Backbone feature maps of `n` image patches: {'layer2': (n,C_layer2,H2,W2), 'layer3': (n,C_backbone,H,W)}.
    Every patch shows the sequence's appearance pattern (drawn from dims['base_seed'], the same in every call) plus
    per-call noise, so that a filter learnt on the first frame finds the target again in the later ones."""
    base = np.random.default_rng(dims.get("base_seed", 0))
    b3 = base.standard_normal((1, dims["C_backbone"], dims["H"], dims["W"]), dtype=np.float32)
    b2 = base.standard_normal((1, dims["C_layer2"], dims["H2"], dims["W2"]), dtype=np.float32)
    rng = np.random.default_rng(seed)
    sg = np.float32(dims.get("noise", 1.0))
    l3 = b3 + sg * rng.standard_normal((n, dims["C_backbone"], dims["H"], dims["W"]), dtype=np.float32)
    l2 = b2 + sg * rng.standard_normal((n, dims["C_layer2"], dims["H2"], dims["W2"]), dtype=np.float32)
    out = {"layer2": l2, "layer3": l3}
    # the segmentation trackers' decoder also reads layer1 (stride 4) and layer4 (stride 32); drawn BEHIND the two maps above, so the
    # streams of the dims without them are what they always were
    if "C_layer4" in dims:
        b4 = base.standard_normal((1, dims["C_layer4"], (dims["H"] + 1) // 2, (dims["W"] + 1) // 2), dtype=np.float32)
        b1 = base.standard_normal((1, dims["C_layer1"], 2 * dims["H2"], 2 * dims["W2"]), dtype=np.float32)
        out["layer4"] = b4 + sg * rng.standard_normal((n,) + b4.shape[1:], dtype=np.float32)
        out["layer1"] = b1 + sg * rng.standard_normal((n,) + b1.shape[1:], dtype=np.float32)
    return out


def tracker_iou_feat(seed, n, dims):
    """IoU features of the test frame (the output of `AtomIoUNet.get_iou_feat`, stock convolutions)."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n, dims["C_iou"], dims["H2"], dims["W2"]), dtype=np.float32),
            rng.standard_normal((n, dims["C_iou"], dims["H"], dims["W"]), dtype=np.float32))


def tracker_dimp_params(seed, dims):
    """Seeded hot-path weights of a DiMP network: classification-feature head conv ('head.weight'), the filter
    initialiser's conv ('init.weight', 'init.bias') and the IoU predictor's test branch ('iou.*', synth.iou_net_params)."""
    rng = np.random.default_rng(seed + 900)
    C, Cb = dims["C"], dims["C_backbone"]
    p = {"head.weight": rng.standard_normal((C, Cb, 3, 3), dtype=np.float32) * np.float32(math.sqrt(2.0 / (9 * C))),
         "init.weight": rng.standard_normal((C, C, 3, 3), dtype=np.float32) * np.float32(math.sqrt(2.0 / (9 * C))),
         "init.bias": rng.standard_normal(C, dtype=np.float32) * np.float32(0.01)}
    for k, v in iou_net_params(seed + 901, dict(C=dims["C_iou"], I=dims["C_iou"])).items():
        p["iou." + k] = v
    return p


def tracker_tomp_params(seed, dims, cfg=None):
    """Seeded hot-path weights of a ToMP network: the head's final conv ('head.weight') + synth.tomp_params."""
    cfg = cfg or TOMP
    rng = np.random.default_rng(seed + 910)
    C, Cb = dims["C"], dims["C_backbone"]
    p = {"head.weight": rng.standard_normal((C, Cb, 3, 3), dtype=np.float32) * np.float32(math.sqrt(2.0 / (9 * C)))}
    p.update(tomp_params(seed + 911, cfg))
    # random-init scores come out negative everywhere, which degenerates the tracker's peak logic (masked cells are 0):
    # flip the sign of the classifier's filter projection so that the score maps are positive
    p["cls.linear.weight"] = -p["cls.linear.weight"]
    p["cls.linear.bias"] = -p["cls.linear.bias"]
    return p
