"""Self-pinning of the Precise RoI Pooling restatements (PARITY UNPINNED against upstream: the
reference's submodule is empty, see oracle/prroi_torch.py).  CPU only.

  * closed form vs brute-force quadrature of the bilinear interpolant,
  * analytic cases: constant map, linear ramp, impulse, RoI outside the map, zero-area RoI,
  * torch.autograd.gradcheck (fp64) of the torch restatement,
  * analytic numpy backward (features + RoI coordinates) vs torch autograd,
  * the reference modules that consume the op (FilterInitializerLinear, AtomIoUNet.predict_iou)
    executed with the restatement plugged in (tests/golden/filter_init_linear.npz, iou_predict.npz).
"""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from oracle import prroi_torch
from conftest import load_golden


def _bilinear(F, x, y):
    """Interpolant with zero outside the map (hat basis)."""
    H, W = F.shape
    val = 0.0
    j0, i0 = int(np.floor(y)), int(np.floor(x))
    for j in (j0, j0 + 1):
        for i in (i0, i0 + 1):
            if 0 <= j < H and 0 <= i < W:
                val += F[j, i] * max(0, 1 - abs(x - i)) * max(0, 1 - abs(y - j))
    return val


def test_closed_form_vs_quadrature():
    rng = np.random.default_rng(0)
    F = rng.standard_normal((1, 1, 7, 9))
    rois = np.array([[0, 8.0, 4.0, 70.0, 50.0], [0, -20.0, -10.0, 40.0, 30.0]])
    PH, PW, scale = 3, 2, 1 / 8
    out = O.prroi_forward(F, rois, PH, PW, scale)
    M = 120
    for r in range(2):
        x0, y0, x1, y1 = rois[r, 1:] * scale
        bw, bh = (x1 - x0) / PW, (y1 - y0) / PH
        for p in range(PH):
            for q in range(PW):
                xs = x0 + q * bw + (np.arange(M) + 0.5) * bw / M
                ys = y0 + p * bh + (np.arange(M) + 0.5) * bh / M
                acc = np.mean([[_bilinear(F[0, 0], x, y) for x in xs] for y in ys])
                assert abs(acc - out[r, 0, p, q]) < 2e-4


def test_analytic_cases():
    H, W = 12, 10
    const = np.full((1, 2, H, W), 3.5)
    rois = np.array([[0, 2.0, 3.0, 6.5, 8.25]])
    np.testing.assert_allclose(O.prroi_forward(const, rois, 3, 3, 1.0), 3.5, rtol=1e-12)
    jj, ii = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    ramp = (0.3 * ii + 0.7 * jj + 1.0)[None, None].astype(np.float64)
    out = O.prroi_forward(ramp, rois, 2, 3, 1.0)
    x0, y0, x1, y1 = rois[0, 1:]
    for p in range(2):
        for q in range(3):
            cx = x0 + (q + 0.5) * (x1 - x0) / 3
            cy = y0 + (p + 0.5) * (y1 - y0) / 2
            assert abs(out[0, 0, p, q] - (0.3 * cx + 0.7 * cy + 1.0)) < 1e-12
    imp = np.zeros((1, 1, H, W))
    imp[0, 0, 5, 4] = 1.0
    out = O.prroi_forward(imp, np.array([[0, 3.5, 4.5, 4.5, 5.5]]), 1, 1, 1.0)
    wx = O._G(4.5 - 4) - O._G(3.5 - 4)
    assert abs(out[0, 0, 0, 0] - wx * wx) < 1e-12
    far = O.prroi_forward(const, np.array([[0, 30.0, 30.0, 40.0, 40.0]]), 2, 2, 1.0)
    assert np.all(far == 0)
    assert np.all(O.prroi_forward(const, np.array([[0, 3.0, 3.0, 3.0, 8.0]]), 2, 2, 1.0) == 0)
    assert np.all(O.prroi_forward(const, np.array([[0, 5.0, 3.0, 2.0, 8.0]]), 2, 2, 1.0) == 0)


def test_torch_restatement_gradcheck_and_numpy_backward():
    torch.manual_seed(0)
    F = torch.randn(2, 3, 6, 7, dtype=torch.float64, requires_grad=True)
    rois = torch.tensor([[0, 4.3, 6.1, 40.7, 35.2], [1, 10.0, 2.5, 50.0, 44.0], [1, -6.0, 3.0, 20.0, 60.0]],
                        dtype=torch.float64)
    coords = rois[:, 1:].clone().requires_grad_(True)

    def fn(F_, c_):
        return prroi_torch.prroi_pool2d(F_, torch.cat((rois[:, :1], c_), 1), 3, 2, 1 / 8)
    assert torch.autograd.gradcheck(fn, (F, coords), eps=1e-6, atol=1e-6)
    out = fn(F, coords)
    g = torch.randn_like(out)
    gF, gc = torch.autograd.grad(out, (F, coords), g)
    Fn, rn, gn = F.detach().numpy(), rois.numpy(), g.numpy()
    np.testing.assert_allclose(O.prroi_forward(Fn, rn, 3, 2, 1 / 8), out.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(O.prroi_backward_feat(gn, Fn.shape, rn, 3, 2, 1 / 8), gF.numpy(), atol=1e-12)
    np.testing.assert_allclose(O.prroi_backward_coor(gn, Fn, rn, 3, 2, 1 / 8)[:, 1:], gc.numpy(), atol=1e-10)


def test_reference_filter_initializer_consumer():
    """FilterInitializerLinear (initializer.py:151-173) = conv -> FilterPool (xywh->xyxy rois,
    :30-43) -> PrRoIPool2D(4,4,1/16) -> mean over images; golden produced by the reference module."""
    g = load_golden("filter_init_linear")
    bb = g["bb"].astype(np.float64)
    n = bb.shape[0]
    rois = np.concatenate((np.arange(n, dtype=np.float64)[:, None], bb[:, :2], bb[:, :2] + bb[:, 2:]), axis=1)
    pooled = O.prroi_forward(g["conv_out"].astype(np.float64), rois, 4, 4, 1 / 16)
    np.testing.assert_allclose(pooled.mean(0), g["weights"][0], atol=2e-6)


def test_reference_iou_predictor_consumer():
    """AtomIoUNet.predict_iou (atom_iou_net.py:96-136) forward and d iou / d proposals, rebuilt from
    the numpy PrRoIPool (+ its coordinate backward) and the stored LinearBlock weights."""
    g = load_golden("iou_predict")
    sd = {k[3:].replace("__", "."): v.astype(np.float64) for k, v in g.items() if k.startswith("sd_")}
    c3 = g["c3"].astype(np.float64) * g["mod3"].astype(np.float64).reshape(1, -1, 1, 1)
    c4 = g["c4"].astype(np.float64) * g["mod4"].astype(np.float64).reshape(1, -1, 1, 1)
    props = g["proposals"][0].astype(np.float64)
    R = props.shape[0]
    rois = np.concatenate((np.zeros((R, 1)), props[:, :2], props[:, :2] + props[:, 2:]), axis=1)
    r3 = O.prroi_forward(c3, rois, 5, 5, 1 / 8)
    r4 = O.prroi_forward(c4, rois, 3, 3, 1 / 16)

    def block(x, pre):      # LinearBlock: Linear -> BatchNorm2d(eval) -> ReLU   (ltr/models/layers/blocks.py)
        z = x.reshape(R, -1) @ sd[pre + ".linear.weight"].T + sd[pre + ".linear.bias"]
        zn = (z - sd[pre + ".bn.running_mean"]) / np.sqrt(sd[pre + ".bn.running_var"] + 1e-5)
        zn = zn * sd[pre + ".bn.weight"] + sd[pre + ".bn.bias"]
        return np.maximum(zn, 0), (zn > 0), sd[pre + ".bn.weight"] / np.sqrt(sd[pre + ".bn.running_var"] + 1e-5)
    f3, m3, s3 = block(r3, "fc3_rt")
    f4, m4, s4 = block(r4, "fc4_rt")
    cat = np.concatenate((f3, f4), axis=1)
    iou = cat @ sd["iou_predictor.weight"].T + sd["iou_predictor.bias"]
    np.testing.assert_allclose(iou[:, 0], g["iou"][0], atol=2e-5)
    # backward of sum(iou) to the pooled features, then to the roi coordinates
    gcat = np.repeat(sd["iou_predictor.weight"], R, axis=0)
    n3 = f3.shape[1]
    g3 = ((gcat[:, :n3] * m3 * s3) @ sd["fc3_rt.linear.weight"]).reshape(r3.shape)
    g4 = ((gcat[:, n3:] * m4 * s4) @ sd["fc4_rt.linear.weight"]).reshape(r4.shape)
    gr = O.prroi_backward_coor(g3, c3, rois, 5, 5, 1 / 8) + O.prroi_backward_coor(g4, c4, rois, 3, 3, 1 / 16)
    # xyxy -> xywh chain rule: x1 = x + w, so d/dx = d/dx0 + d/dx1, d/dw = d/dx1
    gxywh = np.stack((gr[:, 1] + gr[:, 3], gr[:, 2] + gr[:, 4], gr[:, 3], gr[:, 4]), axis=1)
    np.testing.assert_allclose(gxywh, g["grad"][0], atol=2e-5, rtol=1e-4)
