import sys, types, numpy as np, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from conftest import load_golden
import test_gpu_parity as G
from oracle import iou_oracle as IO
from pytracking_amd import iou_refine as IR
g = load_golden("iou_refine")
net = G._IoUNetStandIn(g).to("cuda").eval()
T=G.T
t64 = lambda a: torch.from_numpy(a.astype(np.float64))
p = {k[2:]: t64(v) for k, v in g.items() if k.startswith("w_")}
class P(types.SimpleNamespace):
    def get(self, name, default=None): return getattr(self, name, default)
for tag, rel, atom in (("default",False,False),("default_decay",False,False),("relative",True,False),("atom_default",False,True),("atom_relative",True,True),("atom_nodecay",False,True)):
    iters, step, decay = g[f"{tag}_cfg"]
    fn = IO.refine_atom if atom else IO.refine
    out = fn(p, (t64(g["mod3"]), t64(g["mod4"])), (t64(g["c3"]), t64(g["c4"])), t64(g["boxes"]), int(iters), float(step), float(decay), rel)
    b64=out[0].numpy()
    params = P(box_refinement_iter=int(iters), box_refinement_step_length=float(step), box_refinement_step_decay=float(decay), box_refinement_space="relative" if rel else "default")
    if atom:
        me = types.SimpleNamespace(params=params, iou_predictor=net, target_feat=(T(g["mod3"]), T(g["mod4"])))
        b,i = IR.optimize_boxes_atom(me, (T(g["c3"]), T(g["c4"])), torch.from_numpy(g["boxes"].copy()))
    else:
        me = types.SimpleNamespace(params=params, net=types.SimpleNamespace(bb_regressor=net), iou_modulation=(T(g["mod3"]), T(g["mod4"])))
        b,i = (IR.optimize_boxes_relative if rel else IR.optimize_boxes_default)(me, (T(g["c3"]), T(g["c4"])), torch.from_numpy(g["boxes"].copy()))
    print(tag, "ref-vs-f64 %.2e  gpu-vs-f64 %.2e  gpu-vs-ref %.2e" % (np.abs(g[f"{tag}_boxes"]-b64).max(), np.abs(b.numpy()-b64).max(), np.abs(b.numpy()-g[f"{tag}_boxes"]).max()))
