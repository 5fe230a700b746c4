"""TEST / BASELINE INFRASTRUCTURE ONLY -- the synthetic tracking frame of bench.py executed by the REFERENCE's own modules.

`ReferenceTracker` has the interface of oracle/frame_port.TorchCpuTracker (`step(test_feat, slot, num_iter)`), but every hot-path op
is the reference's Python, imported unmodified (oracle/ref_harness.py: /root/reference in the build container, the byte-for-byte
bundle oracle/_ref/reference on the GPU box):

    classify   ltr.models.layers.filter.apply_filter                       (filter.py:5-57)
    solve      ltr.models.target_classifier.optimizer.DiMPSteepestDescentGN  (optimizer.py:10-170), module dispatch and all
               / PrDiMPSteepestDescentNewton (optimizer.py:274-439) for kind="prdimp"

bench.py times it as `cpu_baseline` (kind "reference") on the host cores and, with device="cuda", as `gpu_stock_baseline`
(the same modules `.to('cuda')`: what a user of the reference gets on ROCm without the gfx950 library).  The frame composition around
the two ops (arg-max, re-centred box, memory insert) is the one of `TorchCpuTracker.step` / `bench_frame.TrackState.step`.
Never imported by the product path.
"""
from oracle import ref_harness
from pytracking_amd import synth


def available():
    return ref_harness.available()


class ReferenceTracker:
    def __init__(self, cfg, n, seed, threads=None, device="cpu", kind="dimp"):
        import torch
        ref_harness.install()
        import ltr.models.layers.filter as rfilter
        import ltr.models.target_classifier.optimizer as roptim
        from pytracking_amd import install as amd
        if amd._state["installed"]:
            raise RuntimeError("ReferenceTracker is the baseline: pytracking_amd.install() must not be active")
        self.torch, self.rfilter = torch, rfilter
        if threads:
            torch.set_num_threads(int(threads))
        self.cfg, self.n, self.kind = dict(cfg), n, kind
        self.dev = dev = torch.device(device)
        w0, feat, bb, sw = synth.dimp_problem(seed, n, cfg)
        T = lambda a: torch.from_numpy(a).to(dev)
        if kind == "prdimp":
            w0 = w0 * 0
            self.opt = roptim.PrDiMPSteepestDescentNewton(
                num_iter=cfg["num_iter"], feat_stride=cfg["feat_stride"], init_step_length=cfg["init_step_length"],
                init_filter_reg=cfg["init_filter_reg"], gauss_sigma=cfg["gauss_sigma"], min_filter_reg=cfg["min_filter_reg"],
                alpha_eps=cfg["alpha_eps"], init_uni_weight=cfg["init_uni_weight"], normalize_label=cfg["normalize_label"],
                label_shrink=cfg["label_shrink"], softmax_reg=cfg["softmax_reg"], label_threshold=cfg["label_threshold"])
        else:
            self.opt = roptim.DiMPSteepestDescentGN(
                num_iter=cfg["num_iter"], feat_stride=cfg["feat_stride"], init_step_length=cfg["init_step_length"],
                init_filter_reg=cfg["init_filter_reg"], init_gauss_sigma=cfg["init_gauss_sigma"],
                num_dist_bins=cfg["num_dist_bins"], bin_displacement=cfg["bin_displacement"],
                mask_init_factor=cfg["mask_init_factor"], score_act=cfg["score_act"], mask_act=cfg["mask_act"],
                min_filter_reg=cfg["min_filter_reg"], alpha_eps=cfg["alpha_eps"])
        self.opt = self.opt.eval().to(dev)
        self.mem_feat, self.mem_bb, self.sw, self.filter = T(feat), T(bb), T(sw), T(w0)[None]

    def step(self, test_feat, slot, num_iter):
        torch, c = self.torch, self.cfg
        with torch.no_grad():
            scores = self.rfilter.apply_filter(test_feat[None], self.filter)[0, 0]
            flat = int(torch.argmax(scores))
            row, col = divmod(flat, scores.shape[1])
            off = (c["K"] % 2) / 2.0
            self.mem_bb[slot, 0] = (col + off) * c["feat_stride"] - self.mem_bb[slot, 2] / 2.0
            self.mem_bb[slot, 1] = (row + off) * c["feat_stride"] - self.mem_bb[slot, 3] / 2.0
            self.mem_feat[slot] = test_feat
            self.filter = self.opt(self.filter, self.mem_feat, self.mem_bb, sample_weight=self.sw, num_iter=num_iter,
                                   compute_losses=False)[0]
        return scores
