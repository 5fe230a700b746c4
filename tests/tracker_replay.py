"""TEST INFRASTRUCTURE -- replay the boundary-call log of an unmodified reference tracker run (oracle/tracker_harness.py)
through an implementation of the hot-path ops, in the tracker's own call order.

Closed loop on the solver state: the filter, the sample memory and the classification features come from the
implementation under test and are carried from call to call exactly as the tracker carries them
(dimp.py:410-441 init_memory / update_memory, :536-648 init_classifier / update_classifier).  The tracker's glue --
positions, scales, proposal jitter, schedules -- is taken from the log.  What stays on stock PyTorch in front of the
path (backbone, IoU-feature convolutions) is regenerated from the seeds of pytracking_amd/synth.py.

`ops` supplies: head(l3) -> (n,C,H,W); get_filter(feat, bb, num_iter) -> (1,C,K,K); optimize(filt, feat, bb, sw, num_iter);
classify(filt, feat) -> (1,1,O,O); localize(scores (S,O,O), ev, cfg) -> (tv, scale_ind, flag);
refine(method, (c3, c4), (mod3, mod4), init_boxes, cfg) -> (boxes, iou); to_numpy(t).
"""
import numpy as np

from pytracking_amd import synth


def events_from_npz(g):
    n = int(g["n_events"])
    evs = [dict() for _ in range(n)]
    for key, v in g.items():
        if key.startswith("e") and "_" in key and key[1:key.index("_")].isdigit():
            i = int(key[1:key.index("_")])
            evs[i][key[key.index("_") + 1:]] = v
    for ev in evs:
        ev["kind"] = str(ev["kind"])
    return evs


def replay(events, ops, atol=1e-4, exact=False):
    """Walk the log; returns a dict of the maximum deviation per event kind.  `exact`: demand bit equality (the ops ARE
    the reference's own modules: validates the log and this player)."""
    cfg = events[0]
    assert cfg["kind"] == "config"
    dims = {k[4:]: (int(v) if float(v).is_integer() else float(v)) for k, v in cfg.items() if k.startswith("dim_")}
    seed = int(cfg["seed"])
    k_backbone = k_iou = 0
    head = filt = memory = None
    scores = None
    dev = {}

    def check(kind, got, want, tol=None):
        got, want = np.asarray(ops.to_numpy(got), dtype=np.float64), np.asarray(want, dtype=np.float64)
        assert got.shape == want.shape, (kind, got.shape, want.shape)
        err = float(np.abs(got - want).max()) if got.size else 0.0
        dev[kind] = max(dev.get(kind, 0.0), err)
        if exact:
            assert err == 0.0, (kind, err)
        else:
            assert err <= (atol if tol is None else tol), (kind, err)

    for ev in events[1:]:
        kind = ev["kind"]
        if kind == "head":
            n = int(ev["n"])
            l3 = synth.tracker_backbone(seed + k_backbone, n, dims)["layer3"]
            k_backbone += 1
            head = ops.head(l3)
            got = float(np.abs(np.asarray(ops.to_numpy(head), dtype=np.float64)).sum())
            assert abs(got - float(ev["checksum"])) <= 1e-5 * float(ev["checksum"]), ("head", got, float(ev["checksum"]))
        elif kind == "get_filter":
            filt = ops.get_filter(head, ev["bb"], int(ev["num_iter"]))
            check("get_filter", filt, ev["filter"])
            memory = ops.new_memory(int(cfg["memory_size"]), head)
        elif kind == "optimizer_params":
            ops.set_optimizer_params(ev)
        elif kind == "classify":
            scores = ops.classify(filt, head)
            check("classify", scores, ev["scores"])
        elif kind == "localize":
            tv, scale_ind, flag = ops.localize(scores, ev, cfg)
            assert flag == str(ev["flag"]) and int(scale_ind) == int(ev["scale_ind"]), (flag, str(ev["flag"]))
            check("localize", tv, ev["tv"], tol=1e-3)
        elif kind == "refine":
            c3, c4 = synth.tracker_iou_feat(seed + 5000 + k_iou, 1, dims)
            k_iou += 1
            boxes, iou = ops.refine(str(ev["method"]), (c3, c4), (ev["mod3"], ev["mod4"]), ev["init_boxes"], cfg)
            check("refine_iou", iou, ev["iou"])
            check("refine_boxes", boxes, ev["boxes"], tol=None if exact else 2e-4)
        elif kind == "memory":
            ops.store(memory, int(ev["slot"]), head)
        elif kind == "optimize":
            filt = ops.optimize(filt, ops.memory_view(memory, int(ev["n"])), ev["bb"], ev["sw"], int(ev["num_iter"]))
            check("optimize", filt, ev["filter"])
    return dev


class MirrorOps:
    """The ops served by the gfx950 path: the modules `pytracking_amd.install()` binds under the reference's names, built
    here directly (the GPU box has no reference tree) with the seeded weights of synth.tracker_dimp_params."""

    def __init__(self, events, device="cuda"):
        import math
        import types
        import torch
        from pytracking_amd import features, optimizer
        from pytracking_amd.prroi_pool import PrRoIPool2D
        self.torch, self.types, self.dev = torch, types, torch.device(device)
        cfg = events[0]
        self.dims = dims = {k[4:]: (int(v) if float(v).is_integer() else float(v)) for k, v in cfg.items() if k.startswith("dim_")}
        p = synth.tracker_dimp_params(int(cfg["seed"]), dims)
        C, K = dims["C"], dims["K"]
        self.K = K
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.T = T
        # classification-feature head as dimpnet50 builds it (dimpnet.py:159-172)
        self.head_mod = features.residual_bottleneck(feature_dim=dims["C_backbone"] // 4, num_blocks=0, l2norm=True,
                                                     final_conv=True, norm_scale=math.sqrt(1.0 / (C * K * K)), out_dim=C)
        self.head_mod.load_state_dict({"0.weight": torch.from_numpy(p["head.weight"])}, strict=True)
        self.head_mod.to(self.dev).eval()
        self.init_w, self.init_b = T(p["init.weight"]), T(p["init.bias"])
        self.pool = PrRoIPool2D(K, K, 1 / 16)
        self.opt = optimizer.DiMPSteepestDescentGN(num_iter=5, feat_stride=16, init_step_length=0.9, init_filter_reg=0.1,
                                                   init_gauss_sigma=0.9, num_dist_bins=100, bin_displacement=0.1,
                                                   mask_init_factor=3.0, score_act='relu', mask_act='sigmoid').to(self.dev).eval()
        self.iou_params = {k[4:]: v for k, v in p.items() if k.startswith("iou.")}
        self.iou_net = None

    def to_numpy(self, t):
        return t.detach().cpu().numpy() if isinstance(t, self.torch.Tensor) else np.asarray(t)

    def head(self, l3):
        with self.torch.no_grad():
            return self.head_mod(self.T(l3))

    def set_optimizer_params(self, ev):
        sd = self.opt.state_dict()
        sd["log_step_length"] = self.T(ev["log_step_length"])
        sd["filter_reg"] = self.T(ev["filter_reg"])
        sd["label_map_predictor.weight"] = self.T(ev["label_lut"]).reshape(1, -1, 1, 1)
        sd["target_mask_predictor.0.weight"] = self.T(ev["mask_lut"]).reshape(1, -1, 1, 1)
        sd["spatial_weight_predictor.weight"] = self.T(ev["spatial_lut"]).reshape(1, -1, 1, 1)
        self.opt.load_state_dict(sd, strict=True)
        self.opt.min_filter_reg = float(ev["min_filter_reg"])

    def get_filter(self, feat, bb, num_iter):
        """LinearFilter.get_filter (linear_filter.py:82-103): FilterInitializerLinear (conv -> PrRoIPool(K,K,1/16) -> mean
        over the images, initializer.py:21-45,151-173) followed by the filter optimiser."""
        torch = self.torch
        bb = self.T(bb)
        n = feat.shape[0]
        with torch.no_grad():
            conv = torch.nn.functional.conv2d(feat, self.init_w, self.init_b, padding=1)
            rois = torch.cat((torch.arange(n, dtype=torch.float32, device=self.dev).reshape(-1, 1), bb[:, :2],
                              bb[:, :2] + bb[:, 2:]), dim=1)
            w0 = self.pool(conv, rois).mean(dim=0, keepdim=True)
            return self.opt(w0, feat=feat, bb=bb, num_iter=num_iter, compute_losses=False)[0]

    def new_memory(self, size, head):
        mem = head.new_zeros(size, *head.shape[1:])
        mem[:head.shape[0]] = head
        return mem

    def store(self, mem, slot, head):
        mem[slot:slot + 1] = head

    def memory_view(self, mem, n):
        return mem[:n]

    def optimize(self, filt, feat, bb, sw, num_iter):
        with self.torch.no_grad():
            return self.opt(filt, num_iter=num_iter, feat=feat, bb=self.T(bb), sample_weight=self.T(sw),
                            compute_losses=False)[0]

    def classify(self, filt, feat):
        from pytracking_amd import filter as F
        return F.apply_filter(feat, filt)

    class _Params:
        def __init__(self, **kw):
            self.__dict__.update(kw)

        def get(self, name, default=None):
            return getattr(self, name, default)

    def _params(self, cfg):
        return self._Params(**{k: float(cfg[k]) for k in ("target_not_found_threshold", "distractor_threshold",
                                                           "hard_negative_threshold", "target_neighborhood_scale",
                                                           "dispalcement_scale", "box_refinement_step_length",
                                                           "box_refinement_step_decay")},
                            box_refinement_iter=int(cfg["box_refinement_iter"]))

    def localize(self, scores, ev, cfg):
        from pytracking_amd import localization as LM
        torch = self.torch
        me = self.types.SimpleNamespace(params=self._params(cfg), kernel_size=torch.from_numpy(ev["kernel_size"]),
                                        output_window=None, img_support_sz=torch.from_numpy(ev["img_support_sz"]),
                                        target_sz=torch.from_numpy(ev["target_sz"]), pos=torch.from_numpy(ev["pos"]))
        tv, scale_ind, _, flag = LM.localize_advanced(me, scores.squeeze(1), torch.from_numpy(ev["sample_pos"]),
                                                      torch.from_numpy(ev["sample_scales"]))
        return tv, scale_ind, flag

    def refine(self, method, feats, mods, init_boxes, cfg):
        from pytracking_amd import iou_refine as IR
        torch = self.torch
        if self.iou_net is None:
            C = self.dims["C_iou"]

            class Net(torch.nn.Module):
                def __init__(net):
                    super().__init__()
                    for name, k in (("fc3_rt", 5), ("fc4_rt", 3)):
                        blk = torch.nn.Module()
                        blk.linear, blk.bn, blk.relu = torch.nn.Linear(C * k * k, C), torch.nn.BatchNorm2d(C), torch.nn.ReLU()
                        setattr(net, name, blk)
                    net.iou_predictor = torch.nn.Linear(2 * C, 1)
                    net.prroi_pool3t = self.types.SimpleNamespace(pooled_height=5, pooled_width=5, spatial_scale=1 / 8)
                    net.prroi_pool4t = self.types.SimpleNamespace(pooled_height=3, pooled_width=3, spatial_scale=1 / 16)
            net = Net()
            net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in self.iou_params.items()}, strict=False)
            self.iou_net = net.to(self.dev).eval()
        me = self.types.SimpleNamespace(params=self._params(cfg), net=self.types.SimpleNamespace(bb_regressor=self.iou_net),
                                        iou_modulation=[self.T(m) for m in mods])
        return getattr(IR, method)(me, [self.T(f) for f in feats], torch.from_numpy(np.ascontiguousarray(init_boxes)))


# ------------------------------------------------------------------------------------------------------
# ToMP: every frame = head features of the test frame and of the memory frames -> filter prediction -> classifier and
# box regressor -> localisation (tomp.py:142-303).  The memory holds BACKBONE features, i.e. seeds here.
# ------------------------------------------------------------------------------------------------------
def replay_tomp(events, ops, atol=1e-4):
    cfg = events[0]
    dims = {k[4:]: (int(v) if float(v).is_integer() else float(v)) for k, v in cfg.items() if k.startswith("dim_")}
    seed = int(cfg["seed"])
    dev = {}

    def check(kind, got, want, tol):
        got, want = np.asarray(ops.to_numpy(got), dtype=np.float64), np.asarray(want, dtype=np.float64)
        assert got.shape == want.shape, (kind, got.shape, want.shape)
        err = float(np.abs(got - want).max())
        dev[kind] = max(dev.get(kind, 0.0), err)
        assert err <= tol, (kind, err)

    scores = None
    for ev in events[1:]:
        if ev["kind"] == "tomp_classify":
            l3 = lambda k: synth.tracker_backbone(seed + int(k), 1, dims)["layer3"]
            test = l3(ev["test_call"])
            train = np.concatenate([l3(k) for k in np.atleast_1d(ev["train_calls"])], axis=0)
            scores, bbox = ops.classify(test, train, ev["labels"], ev["ltrb"], int(ev["num_gth_frames"]))
            check("scores", scores, ev["scores"], atol)
            # the regressor ends in exp(): compare in the log domain where the 1e-4 bound is meaningful
            check("log_bbox", np.log(np.maximum(np.asarray(ops.to_numpy(bbox), dtype=np.float64), 1e-30)),
                  np.log(np.maximum(ev["bbox"].astype(np.float64), 1e-30)), 10 * atol)
        elif ev["kind"] == "tomp_localize":
            tv, scale_ind, flag, loc = ops.localize(scores, ev, cfg)
            assert flag == str(ev["flag"]) and int(scale_ind) == int(ev["scale_ind"]), (flag, str(ev["flag"]))
            check("localize", tv, ev["tv"], 1e-3)
            check("score_loc", loc, ev["score_loc"], 0.0)
    return dev


class TompMirrorOps:
    """`replay_tomp` served by the gfx950 ToMP modules (pytracking_amd.transformer / features / localization)."""

    def __init__(self, events, device="cuda"):
        import math
        import types
        import torch
        from pytracking_amd import features, transformer as TM
        self.torch, self.types, self.dev = torch, types, torch.device(device)
        cfg = events[0]
        dims = {k[4:]: (int(v) if float(v).is_integer() else float(v)) for k, v in cfg.items() if k.startswith("dim_")}
        p = synth.tracker_tomp_params(int(cfg["seed"]), dims)
        c = synth.TOMP
        C = dims["C"]
        self.T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.head = features.residual_bottleneck(feature_dim=dims["C_backbone"] // 4, num_blocks=0, l2norm=True, final_conv=True,
                                                 norm_scale=math.sqrt(1.0 / (C * 1 * 1)), out_dim=C)
        self.head.load_state_dict({"0.weight": torch.from_numpy(p["head.weight"])}, strict=True)
        self.head.to(self.dev).eval()
        tr = TM.Transformer(d_model=C, nhead=c["nhead"], num_encoder_layers=c["n_enc"], num_decoder_layers=c["n_dec"],
                            dim_feedforward=c["ff"])
        self.pred = TM.FilterPredictor(tr, feature_sz=c["feature_sz"], use_test_frame_encoding=True)
        self.cls = TM.LinearFilterClassifier(num_channels=C)
        self.reg = TM.DenseBoxRegressor(num_channels=C)
        for mod, pre in ((self.pred, "fp."), (self.cls, "cls."), (self.reg, "reg.")):
            sd = {k[len(pre):]: torch.from_numpy(v.copy()) for k, v in p.items() if k.startswith(pre)}
            if pre == "fp.":
                sd["query_embed_fg_decoder.weight"] = sd["query_embed_fg.weight"]
                for idx in (1, 4):
                    sd[f"box_encoding.{idx}.num_batches_tracked"] = torch.tensor(0)
            mod.load_state_dict(sd, strict=True)
            mod.to(self.dev).eval()

    def to_numpy(self, t):
        return t.detach().cpu().numpy() if isinstance(t, self.torch.Tensor) else np.asarray(t)

    def classify(self, test_l3, train_l3, labels, ltrb, num_gth_frames):
        with self.torch.no_grad():
            test_feat = self.head(self.T(test_l3))                      # Head.extract_head_feat (heads.py:56-64)
            train_feat = self.head(self.T(train_l3))
            cw, bw, cenc, benc = self.pred.predict_cls_bbreg_filters_parallel(train_feat, test_feat, self.T(labels),
                                                                              num_gth_frames, self.T(ltrb))
            return self.cls(cenc, cw), self.reg(benc, bw)

    def localize(self, scores, ev, cfg):
        from pytracking_amd import localization as LM
        torch = self.torch
        params = MirrorOps._Params(**{k: float(cfg[k]) for k in ("target_not_found_threshold", "distractor_threshold",
                                                                 "hard_negative_threshold", "target_neighborhood_scale",
                                                                 "dispalcement_scale")})
        me = self.types.SimpleNamespace(params=params, kernel_size=torch.from_numpy(ev["kernel_size"]), output_window=None,
                                        img_support_sz=torch.from_numpy(ev["img_support_sz"]),
                                        target_sz=torch.from_numpy(ev["target_sz"]), pos=torch.from_numpy(ev["pos"]))
        tv, scale_ind, _, flag, loc = LM.localize_advanced_tomp(me, scores.reshape(-1, *scores.shape[-2:]),
                                                                torch.from_numpy(ev["sample_pos"]),
                                                                torch.from_numpy(ev["sample_scales"]))
        return tv, scale_ind, flag, loc


# ------------------------------------------------------------------------------------------------------
# ATOM (atom.py): first-frame joint Gauss-Newton on (filter, projection matrix), per-frame classification of the compressed test
# sample (`operation.conv2d(mode='same')`), IoU-guided refinement with per-proposal backtracking, CG update of the filter over the
# 250-slot memory of compressed samples.  Closed loop on projection matrix, filter and memory; the glue (positions, flags, labels,
# sample weights, proposals) from the log; backbone / IoU features regenerated from the seeds.
#   ops: gn(f0, P0, raw, y, sw, num_cg, num_gn, filter_reg, projection_reg) -> (f, P);  project(raw, P) -> compressed;
#        classify(f, x) -> (1,1,H,W);  cg_new(memory, y, sw, f, filter_reg, act_min) -> handle;  cg_run(handle, num_iter) -> f;
#        refine((c3, c4), (mod3, mod4), init_boxes, cfg) -> (boxes, iou);  normalize(raw) -> torch;  to_numpy(t)
# ------------------------------------------------------------------------------------------------------
def replay_atom(events, ops, atol=1e-4):
    cfg = events[0]
    dims = {k[4:]: (int(v) if float(v).is_integer() else float(v)) for k, v in cfg.items() if k.startswith("dim_")}
    seed = int(cfg["seed"])
    dev = {}

    def check(kind, got, want, tol):
        got, want = np.asarray(ops.to_numpy(got), dtype=np.float64), np.asarray(want, dtype=np.float64)
        assert got.shape == want.shape, (kind, got.shape, want.shape)
        err = float(np.abs(got - want).max())
        dev[kind] = max(dev.get(kind, 0.0), err)
        assert err <= tol, (kind, err)

    raw = lambda k, n: ops.normalize(synth.tracker_backbone(seed + int(k), int(n), dims)["layer3"])
    filt = proj = memory = mem_y = mem_sw = cg = test_x = None
    for ev in events[1:]:
        kind = ev["kind"]
        if kind == "atom_gn":
            init_raw = raw(ev["init_call"], ev["n_aug"])
            filt, proj = ops.gn(ev["filter0"], ev["proj0"], init_raw, ev["y"], ev["sw"], int(ev["num_cg_iter"]), int(ev["num_gn_iter"]),
                                float(ev["filter_reg"]), float(ev["projection_reg"]))
            check("gn_filter", filt, ev["filter"], atol)
            check("gn_projection", proj, ev["proj"], atol)
            init_x = ops.project(init_raw, proj)                            # atom.py:186-189: re-project with the new matrix
        elif kind == "atom_init_done":
            memory, mem_y, mem_sw = ops.new_memory(int(cfg["memory_size"]), init_x, ev["y_init"], ev["sw"])
            cg = ops.cg_new(memory, mem_y, mem_sw, filt, float(cfg["filter_reg"]), float(cfg["act_min_val"]))
        elif kind == "atom_classify":
            test_x = ops.project(raw(ev["test_call"], 1), proj)
            scores = ops.classify(ops.current_filter(cg), test_x)
            check("classify", scores, ev["scores"], atol)
        elif kind == "atom_refine":
            c3, c4 = synth.tracker_iou_feat(seed + 5000 + int(ev["iou_call"]), 1, dims)
            boxes, iou = ops.refine((c3, c4), (ev["mod3"], ev["mod4"]), ev["init_boxes"], cfg)
            check("refine_iou", iou, ev["iou"], atol)
            check("refine_boxes", boxes, ev["boxes"], 2e-4)
        elif kind == "atom_memory":
            ops.store(memory, mem_y, int(ev["slot"]), test_x, ev["y"])
        elif kind == "atom_cg":
            ops.set_weights(mem_sw, ev["sw"])
            f = ops.cg_run(cg, int(ev["num_iter"]))
            check("cg_filter", f, ev["filter"], atol)
    return dev


class AtomMirrorOps:
    """`replay_atom` served by the gfx950 modules: pytracking_amd.optimization (GaussNewtonCG / ConjugateGradient on the explicit
    Gauss-Newton operators), filter.corr_raw (operation.conv2d 'same'), iou_refine.optimize_boxes_atom."""

    def __init__(self, events, device="cuda"):
        import types
        import torch
        self.torch, self.types, self.dev = torch, types, torch.device(device)
        self.T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        cfg = events[0]
        self.dims = {k[4:]: (int(v) if float(v).is_integer() else float(v)) for k, v in cfg.items() if k.startswith("dim_")}
        self.iou_params = synth.iou_net_params(int(cfg["seed"]) + 901, dict(C=self.dims["C_iou"], I=self.dims["C_iou"]))
        self.iou_net = None

    def to_numpy(self, t):
        return t.detach().cpu().numpy() if isinstance(t, self.torch.Tensor) else np.asarray(t)

    def normalize(self, raw):                                               # featurebase.py:104-108 (stock torch)
        x = self.T(raw)
        return x / (self.torch.sum(x.abs().view(x.shape[0], 1, 1, -1) ** 2, dim=3, keepdim=True) /
                    (x.shape[1] * x.shape[2] * x.shape[3]) + 1e-10) ** (1 / 2)

    def gn(self, f0, P0, raw, y, sw, num_cg, num_gn, filter_reg, projection_reg):
        import torch.nn.functional as F
        from pytracking_amd.optimization import FactorizedConvProblem, GaussNewtonCG
        prob = FactorizedConvProblem([raw], [self.T(y)], [filter_reg], [projection_reg], None, [self.T(sw)],
                                     lambda x: x, lambda x: F.elu(F.leaky_relu(x, 1 / 0.05), 0.05))
        filt, proj = self.T(f0).clone(), self.T(P0).clone()
        opt = GaussNewtonCG(prob, [filt, proj])
        if num_gn < 0:
            opt.run(num_cg)
        else:
            opt.run(num_cg, num_gn)
        return filt, proj

    def project(self, raw, proj):
        return self.torch.nn.functional.conv2d(raw, proj)                   # stock 1x1 convolution (operation.conv2d)

    def new_memory(self, size, init_x, y_init, sw):
        mem = init_x.new_zeros(size, *init_x.shape[1:])
        mem[:init_x.shape[0]] = init_x
        y = init_x.new_zeros(size, 1, *init_x.shape[2:])
        y[:init_x.shape[0]] = self.T(y_init)
        return mem, y, self.T(sw).clone()

    def cg_new(self, memory, y, sw, filt, filter_reg, act_min):
        from pytracking_amd.optimization import ConjugateGradient, ConvProblem, MLU
        x = [filt.clone()]
        opt = ConjugateGradient(ConvProblem([memory], [y], [filter_reg], [sw], MLU(act_min)), x, fletcher_reeves=False,
                                direction_forget_factor=0)
        return (opt, x)

    def current_filter(self, cg):
        return cg[1][0]

    def cg_run(self, cg, num_iter):
        cg[0].run(num_iter)
        return cg[1][0]

    def classify(self, filt, x):
        from pytracking_amd import filter as FL
        return FL.corr_raw(x, filt[0], out_hw=tuple(x.shape[-2:])).unsqueeze(1)

    def store(self, memory, mem_y, slot, x, y):
        memory[slot:slot + 1] = x
        mem_y[slot:slot + 1] = self.T(y)

    def set_weights(self, mem_sw, sw):
        mem_sw.copy_(self.T(sw))

    def refine(self, feats, mods, init_boxes, cfg):
        from pytracking_amd import iou_refine as IR
        torch = self.torch
        if self.iou_net is None:
            C = self.dims["C_iou"]
            types = self.types

            class Net(torch.nn.Module):
                def __init__(net):
                    super().__init__()
                    for name, k in (("fc3_rt", 5), ("fc4_rt", 3)):
                        blk = torch.nn.Module()
                        blk.linear, blk.bn, blk.relu = torch.nn.Linear(C * k * k, C), torch.nn.BatchNorm2d(C), torch.nn.ReLU()
                        setattr(net, name, blk)
                    net.iou_predictor = torch.nn.Linear(2 * C, 1)
                    net.prroi_pool3t = types.SimpleNamespace(pooled_height=5, pooled_width=5, spatial_scale=1 / 8)
                    net.prroi_pool4t = types.SimpleNamespace(pooled_height=3, pooled_width=3, spatial_scale=1 / 16)
            net = Net()
            net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in self.iou_params.items()}, strict=False)
            self.iou_net = net.to(self.dev).eval()
        params = MirrorOps._Params(box_refinement_iter=int(cfg["box_refinement_iter"]),
                                   box_refinement_step_length=float(cfg["box_refinement_step_length"]),
                                   box_refinement_step_decay=float(cfg["box_refinement_step_decay"]))
        me = self.types.SimpleNamespace(params=params, iou_predictor=self.iou_net, target_feat=[self.T(m) for m in mods])
        return IR.optimize_boxes_atom(me, [self.T(f) for f in feats], torch.from_numpy(np.ascontiguousarray(init_boxes)))
