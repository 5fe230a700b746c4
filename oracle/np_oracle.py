"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's online-optimisation hot path.

Every function cites the reference file:line it restates (paths relative to /root/reference).  The
restatement is pinned against the reference itself: oracle/make_golden.py imports the unmodified
reference in the build container, runs it on seeded inputs and commits inputs+outputs under
tests/golden/; tests/test_oracle_golden.py checks this module against those vectors.  The one piece
with no reference source (PrRoIPool, empty submodule) is "parity unpinned" and self-pinned instead
(tests/test_oracle_prroi.py).

All routines are dtype-generic: pass float64 arrays for a high-precision oracle, float32 arrays to
mimic the reference's arithmetic type.  Pure numpy (no torch) so it runs anywhere.
"""
import math

import numpy as np


# --------------------------------------------------------------------------------------------
# filter layer: ltr/models/layers/filter.py
# --------------------------------------------------------------------------------------------

def apply_filter(feat, filt, out_hw=None):
    """Cross-correlation of one filter (or F filters) with n feature maps.

    Restates `apply_filter` (ltr/models/layers/filter.py:5-57) for one sequence: zero padding
    K//2 (:16), stride 1, so an even K yields H+1 outputs and an odd K yields H.
      feat (n,C,H,W); filt (C,KH,KW) -> (n,OH,OW)   or   filt (F,C,KH,KW) -> (n,F,OH,OW)
    `out_hw` crops the result to its top-left corner (operation.conv2d mode='same',
    pytracking/libs/operation.py:17-32, drops the last row/col for even K).
    """
    multi = filt.ndim == 4
    f = filt if multi else filt[None]
    n, C, H, W = feat.shape
    F, C2, KH, KW = f.shape
    assert C == C2
    ph, pw = KH // 2, KW // 2
    OH, OW = H + 2 * ph - KH + 1, W + 2 * pw - KW + 1
    fp = np.zeros((n, C, H + 2 * ph, W + 2 * pw), dtype=feat.dtype)
    fp[:, :, ph:ph + H, pw:pw + W] = feat
    out = np.zeros((n, F, OH, OW), dtype=feat.dtype)
    for u in range(KH):
        for v in range(KW):
            out += np.einsum("nchw,fc->nfhw", fp[:, :, u:u + OH, v:v + OW], f[:, :, u, v], optimize=True)
    if out_hw is not None:
        out = out[:, :, :out_hw[0], :out_hw[1]]
    return out if multi else out[:, 0]


def apply_feat_transpose(feat, inp, ksz):
    """Adjoint of `apply_filter` w.r.t. the filter.

    Restates `apply_feat_transpose` -> `_apply_feat_transpose_v2/_v3` (filter.py:91-182):
      grad[c,u,v] = sum_{i,y,x} feat[i,c,y+u-p,x+v-p] * inp[i,y,x]   (p = K//2, zero padded)
      feat (n,C,H,W); inp (n,OH,OW) -> (C,KH,KW);  inp (n,F,OH,OW) -> (F,C,KH,KW).
    `inp` may be smaller than the full correlation output (ATOM's 'same' crop): missing rows/cols
    are treated as zero.
    """
    KH, KW = (ksz, ksz) if isinstance(ksz, int) else ksz
    multi = inp.ndim == 4
    r = inp if multi else inp[:, None]
    n, C, H, W = feat.shape
    F = r.shape[1]
    ph, pw = KH // 2, KW // 2
    OH, OW = r.shape[-2:]
    fp = np.zeros((n, C, H + 2 * ph, W + 2 * pw), dtype=feat.dtype)
    fp[:, :, ph:ph + H, pw:pw + W] = feat
    g = np.zeros((F, C, KH, KW), dtype=feat.dtype)
    for u in range(KH):
        for v in range(KW):
            g[:, :, u, v] = np.einsum("nchw,nfhw->fc", fp[:, :, u:u + OH, v:v + OW], r, optimize=True)
    return g if multi else g[0]


# --------------------------------------------------------------------------------------------
# activations + distance map: ltr/models/layers/activation.py, distance.py
# --------------------------------------------------------------------------------------------

def leaky_relu_par(x, a):
    """activation.py:32-37  LeakyReluPar."""
    return (1.0 - a) / 2.0 * np.abs(x) + (1.0 + a) / 2.0 * x


def leaky_relu_par_deriv(x, a):
    """activation.py:39-44  LeakyReluParDeriv (sign(0) = 0)."""
    return (1.0 - a) / 2.0 * np.sign(x) + (1.0 + a) / 2.0


def bent_ident_par(x, a, b=1.0):
    """activation.py:47-55."""
    return (1.0 - a) / 2.0 * (np.sqrt(x * x + 4.0 * b * b) - 2.0 * b) + (1.0 + a) / 2.0 * x


def bent_ident_par_deriv(x, a, b=1.0):
    """activation.py:58-66."""
    return (1.0 - a) / 2.0 * (x / np.sqrt(x * x + 4.0 * b * b)) + (1.0 + a) / 2.0


def softmax_reg(x, reg=None):
    """activation.py:7-16: softmax over the last axis with an optional extra constant logit."""
    m = x.max(axis=-1, keepdims=True)
    if reg is not None:
        m = np.maximum(m, reg)
    e = np.exp(x - m)
    den = e.sum(axis=-1, keepdims=True)
    if reg is not None:
        den = den + np.exp(reg - m)
    return e / den


def distance_map(center, output_sz, num_bins, bin_displacement):
    """distance.py:17-39  DistanceMap.forward.  center (n,2) = (row, col) -> (n,bins,OH,OW)."""
    dt = center.dtype
    k = np.arange(num_bins, dtype=dt).reshape(1, -1, 1, 1)
    k0 = np.arange(output_sz[0], dtype=dt).reshape(1, 1, -1, 1)
    k1 = np.arange(output_sz[1], dtype=dt).reshape(1, 1, 1, -1)
    d0 = k0 - center[:, 0].reshape(-1, 1, 1, 1)
    d1 = k1 - center[:, 1].reshape(-1, 1, 1, 1)
    dist = np.sqrt(d0 * d0 + d1 * d1)
    bin_diff = dist / dt.type(bin_displacement) - k
    head = np.maximum(1.0 - np.abs(bin_diff[:, :-1]), 0.0)
    tail = np.clip(1.0 + bin_diff[:, -1:], 0.0, 1.0)
    return np.concatenate((head, tail), axis=1).astype(dt)


# --------------------------------------------------------------------------------------------
# DiMP / DiMP-L2 / PrDiMP steepest descent: ltr/models/target_classifier/optimizer.py
# --------------------------------------------------------------------------------------------

def _centers(bb, K, feat_stride):
    """optimizer.py:112-113: (row, col) target centre in score-map cells."""
    off = (K % 2) / 2.0
    c = (bb[:, :2] + bb[:, 2:] / 2.0) / bb.dtype.type(feat_stride)
    return c[:, ::-1] - bb.dtype.type(off)


def dimp_sd(w0, feat, bb, sample_weight, *, num_iter, step_length, filter_reg, min_filter_reg,
            feat_stride, label_w, mask_w, spatial_w, bin_displacement, alpha_eps=0.0,
            mask_act="sigmoid", score_act="relu", act_param=None, compute_losses=True):
    """`DiMPSteepestDescentGN.forward` (optimizer.py:85-170), one sequence.

    w0 (C,K,K); feat (n,C,H,W); bb (n,4) xywh; sample_weight (n,) or None.
    label_w/mask_w/spatial_w: the (bins,) weights of the three 1x1 predictors (:57-72).
    Returns (iterates (T+1,C,K,K), losses list).
    """
    dt = feat.dtype
    n, C, H, W = feat.shape
    K = w0.shape[-1]
    O = (H + (K + 1) % 2, W + (K + 1) % 2)
    step = dt.type(step_length)
    reg = dt.type(max(filter_reg * filter_reg, min_filter_reg ** 2))      # :109
    dmap = distance_map(_centers(bb.astype(dt), K, feat_stride), O, len(label_w), bin_displacement)
    label = np.einsum("nbhw,b->nhw", dmap, label_w.astype(dt))           # :117
    mask = np.einsum("nbhw,b->nhw", dmap, mask_w.astype(dt))             # :118
    if mask_act == "sigmoid":
        mask = 1.0 / (1.0 + np.exp(-mask))
    spatial = np.einsum("nbhw,b->nhw", dmap, spatial_w.astype(dt))       # :119
    if sample_weight is None:                                            # :122-125
        sws = dt.type(math.sqrt(1.0 / n)) * spatial
    else:
        sws = np.sqrt(sample_weight.astype(dt)).reshape(n, 1, 1) * spatial
    if score_act == "relu":
        act, dact = leaky_relu_par, leaky_relu_par_deriv
    else:
        act = lambda x, a: bent_ident_par(x, a, act_param)
        dact = lambda x, a: bent_ident_par_deriv(x, a, act_param)
    w = w0.astype(dt)
    iterates, losses = [w], []
    for _ in range(num_iter):
        s = apply_filter(feat, w)                                        # :137
        sa = act(s, mask)
        m = dact(s, mask)
        r = sws * (sa - label)                                           # :140
        if compute_losses:
            losses.append((r ** 2).sum() + reg * (w ** 2).sum())         # :143
        g = apply_feat_transpose(feat, m * (sws * r), K) + reg * w       # :146-148
        q = sws * (m * apply_filter(feat, g))                            # :151-152
        a_num = (g * g).sum()                                            # :155
        a_den = max((q * q).sum() + (reg + dt.type(alpha_eps)) * a_num, dt.type(1e-8))
        w = w - (step * (a_num / a_den)) * g                             # :160
        iterates.append(w)
    if compute_losses:
        sa = act(apply_filter(feat, w), mask)                            # :165-168
        losses.append(((sws * (sa - label)) ** 2).sum() + reg * (w ** 2).sum())
    return np.stack(iterates), losses


def dimp_l2_sd(w0, feat, bb, sample_weight, *, num_iter, step_length, filter_reg, min_filter_reg,
               feat_stride, gauss_sigma, hinge_threshold, alpha_eps=0.0, compute_losses=True):
    """`DiMPL2SteepestDescentGN.forward` (optimizer.py:211-291), one sequence."""
    dt = feat.dtype
    n, C, H, W = feat.shape
    K = w0.shape[-1]
    O = (H + (K + 1) % 2, W + (K + 1) % 2)
    step = dt.type(step_length)
    reg = dt.type(max(filter_reg * filter_reg, min_filter_reg ** 2))
    ctr = _centers(bb.astype(dt), K, feat_stride)
    k0 = np.arange(O[0], dtype=dt).reshape(1, -1, 1)
    k1 = np.arange(O[1], dtype=dt).reshape(1, 1, -1)
    coef = dt.type(-1.0 / (2 * gauss_sigma ** 2))
    g0 = np.exp(coef * (k0 - ctr[:, 0].reshape(-1, 1, 1)) ** 2)         # :201-208
    g1 = np.exp(coef * (k1 - ctr[:, 1].reshape(-1, 1, 1)) ** 2)
    label = g0 * g1
    mask = (label > hinge_threshold).astype(dt)                          # :245
    label = label * mask
    if sample_weight is None:
        sws = np.full((n, 1, 1), math.sqrt(1.0 / n), dtype=dt)
    else:
        sws = np.sqrt(sample_weight.astype(dt)).reshape(n, 1, 1)
    w = w0.astype(dt)
    iterates, losses = [w], []
    for _ in range(num_iter):
        s = apply_filter(feat, w)
        sa = mask * s + (1.0 - mask) * np.maximum(s, 0)                  # :262
        m = mask + (1.0 - mask) * (s > 0).astype(dt)                     # :263
        r = sws * (sa - label)
        if compute_losses:
            losses.append((r ** 2).sum() + reg * (w ** 2).sum())
        g = apply_feat_transpose(feat, m * (sws * r), K) + reg * w
        q = sws * (m * apply_filter(feat, g))
        a_num = (g * g).sum()
        a_den = max((q * q).sum() + (reg + dt.type(alpha_eps)) * a_num, dt.type(1e-8))
        w = w - (step * (a_num / a_den)) * g
        iterates.append(w)
    if compute_losses:
        s = apply_filter(feat, w)
        sa = mask * s + (1.0 - mask) * np.maximum(s, 0)
        losses.append(((sws * (sa - label)) ** 2).sum() + reg * (w ** 2).sum())
    return np.stack(iterates), losses


def prdimp_label_density(ctr, O, *, gauss_sigma, label_threshold=0.0, normalize_label=False,
                         label_shrink=0.0, uni_weight=0.0):
    """`PrDiMPSteepestDescentNewton.get_label_density` (optimizer.py:331-353)."""
    dt = ctr.dtype
    k0 = np.arange(O[0], dtype=dt).reshape(1, -1, 1)
    k1 = np.arange(O[1], dtype=dt).reshape(1, 1, -1)
    d0 = (k0 - ctr[:, 0].reshape(-1, 1, 1)) ** 2
    d1 = (k1 - ctr[:, 1].reshape(-1, 1, 1)) ** 2
    if gauss_sigma == 0:                                                 # :337-344 one-hot argmin
        oh0 = np.zeros_like(d0)
        oh1 = np.zeros_like(d1)
        oh0[np.arange(d0.shape[0]), d0[:, :, 0].argmin(-1), 0] = 1.0
        oh1[np.arange(d1.shape[0]), 0, d1[:, 0, :].argmin(-1)] = 1.0
        gauss = oh0 * oh1
    else:
        coef = dt.type(-1.0 / (2 * gauss_sigma ** 2))
        g0 = np.exp(coef * d0)
        g1 = np.exp(coef * d1)
        gauss = (g0 / dt.type(2 * math.pi * gauss_sigma ** 2)) * g1      # :348
    gauss = gauss * (gauss > label_threshold).astype(dt)                 # :349
    if normalize_label:
        gauss = gauss / (gauss.sum(axis=(-2, -1), keepdims=True) + dt.type(1e-8))
    return (dt.type(1.0 - label_shrink) *
            (dt.type(1.0 - uni_weight) * gauss + dt.type(uni_weight / (O[0] * O[1]))))


def prdimp_sd(w0, feat, bb, sample_weight, *, num_iter, step_length, filter_reg, min_filter_reg,
              feat_stride, gauss_sigma, alpha_eps=0.0, uni_weight=0.0, normalize_label=False,
              label_shrink=0.0, softmax_reg_val=None, label_threshold=0.0, compute_losses=True):
    """`PrDiMPSteepestDescentNewton.forward` (optimizer.py:355-439), one sequence."""
    dt = feat.dtype
    n, C, H, W = feat.shape
    K = w0.shape[-1]
    O = (H + (K + 1) % 2, W + (K + 1) % 2)
    step = dt.type(step_length)
    reg = dt.type(max(filter_reg * filter_reg, min_filter_reg ** 2))
    ctr = _centers(bb.astype(dt), K, feat_stride)
    label = prdimp_label_density(ctr, O, gauss_sigma=gauss_sigma, label_threshold=label_threshold,
                                 normalize_label=normalize_label, label_shrink=label_shrink,
                                 uni_weight=uni_weight)
    if sample_weight is None:                                            # :387-390 (NOT sqrt'ed)
        swp = np.full((n,), 1.0 / n, dtype=dt)
    else:
        swp = sample_weight.astype(dt).reshape(n)
    exp_reg = dt.type(0 if softmax_reg_val is None else math.exp(softmax_reg_val))

    def loss_fn(s, w):                                                   # :393-396
        lse = np.log(np.exp(s).sum(axis=(-2, -1)) + exp_reg)
        return (swp * (lse - (label * s).sum(axis=(-2, -1)))).sum() + reg * (w ** 2).sum()

    w = w0.astype(dt)
    iterates, losses = [w], []
    for _ in range(num_iter):
        s = apply_filter(feat, w)                                        # :406
        P = softmax_reg(s.reshape(n, -1), softmax_reg_val).reshape(s.shape)
        res = swp.reshape(n, 1, 1) * (P - label)                         # :408
        if compute_losses:
            losses.append(loss_fn(s, w))
        g = apply_feat_transpose(feat, res, K) + reg * w                 # :414-415
        sg = apply_filter(feat, g)                                       # :418
        psg = P * sg
        h = psg - P * psg.sum(axis=(-2, -1), keepdims=True)              # :420
        ghg = np.maximum((sg * h).reshape(n, -1).sum(axis=1), 0)         # :421
        ghg = (swp * ghg).sum()                                          # :422
        a_num = (g * g).sum()
        a_den = max(ghg + (reg + dt.type(alpha_eps)) * a_num, dt.type(1e-8))
        w = w - (step * (a_num / a_den)) * g                             # :430
        iterates.append(w)
    if compute_losses:
        losses.append(loss_fn(apply_filter(feat, w), w))
    return np.stack(iterates), losses


# --------------------------------------------------------------------------------------------
# ATOM first-frame joint optimisation: GaussNewtonCG on FactorizedConvProblem
#   pytracking/libs/optimization.py:293-421, 72-163;  pytracking/tracker/atom/optim.py:6-68
# --------------------------------------------------------------------------------------------

def conv1x1(samples, P):
    """operation.conv1x1 (pytracking/libs/operation.py:35-42): samples (n,M,H,W), P (Kc,M) -> (n,Kc,H,W)."""
    return np.einsum("nmhw,km->nkhw", samples, P, optimize=True)


def conv_same_input_grad(v, filt, H, W):
    """Gradient of conv2d(c, filt, mode='same') w.r.t. the input c for an output-side map v (n,H,W):
    gc[i,k,yy,xx] = sum_{u,v} v[i, yy-u+p, xx-v+p] * filt[k,u,v]   (zero outside the HxW output)."""
    n = v.shape[0]
    Kc, KH, KW = filt.shape
    ph, pw = KH // 2, KW // 2
    vp = np.zeros((n, H + KH, W + KW), dtype=v.dtype)
    vp[:, KH - 1 - ph + 0:KH - 1 - ph + H, KW - 1 - pw:KW - 1 - pw + W] = v
    gc = np.zeros((n, Kc, H, W), dtype=v.dtype)
    for u in range(KH):
        for w_ in range(KW):
            # v[yy-u+p] lives at padded row yy - u + p + (KH-1-p) = yy + KH - 1 - u
            gc += vp[:, None, KH - 1 - u:KH - 1 - u + H, KW - 1 - w_:KW - 1 - w_ + W] * filt[None, :, u, w_, None, None]
    return gc


def atom_gn_cg(filter0, P0, samples, y, sample_weights, *, filter_reg, projection_reg, act_min_val, cg_iters,
               fletcher_reeves=True):
    """`GaussNewtonCG.run` (optimization.py:328-407) on `FactorizedConvProblem` (atom/optim.py:6-68), identity
    projection activation, MLU response activation, explicit Jacobian instead of double autograd.

    filter0 (Kc,K,K); P0 (Kc,M) [= (Kc,M,1,1)]; samples (n,M,H,W); y (n,H,W); cg_iters: list, one entry per GN iteration.
    Residuals [sqrt(sw)*(MLU(conv_same(conv1x1(S,P), f)) - y), sqrt(lf)*f, sqrt(lP)*P]  (optim.py:19-45);
    joint inner product over both blocks (:48-65); preconditioner M1 = diag(1/lf, 1/lP) (:67-68, :17).
    Returns (filter, P)."""
    dt = samples.dtype
    n, M, H, W = samples.shape
    K = filter0.shape[-1]
    lf, lP = dt.type(filter_reg), dt.type(projection_reg)
    sqf, sqP = dt.type(math.sqrt(filter_reg)), dt.type(math.sqrt(projection_reg))
    f, P = filter0.astype(dt), P0.astype(dt)
    ssw = np.sqrt(sample_weights).reshape(-1, 1, 1).astype(dt)
    for num_cg in cg_iters:
        c = conv1x1(samples, P)
        s = apply_filter(c, f, out_hw=(H, W))
        d = ssw * mlu_deriv(s, act_min_val)
        r0 = (ssw * (mlu(s, act_min_val) - y), sqf * f, sqP * P)

        def JT(ud, uf, uP):
            v = d * ud
            gf = apply_feat_transpose(c, v, K) + sqf * uf
            gc = conv_same_input_grad(v, f, H, W)
            gP = np.einsum("nkhw,nmhw->km", gc, samples, optimize=True) + sqP * uP
            return gf, gP

        def J(pf, pP):
            dc = conv1x1(samples, pP)
            return d * (apply_filter(c, pf, out_hw=(H, W)) + apply_filter(dc, f, out_hw=(H, W))), sqf * pf, sqP * pP

        ip = lambda a, b: (a[0] * b[0]).sum() + (a[1] * b[1]).sum()
        bf, bP = JT(*r0)
        r = (-bf, -bP)                                                   # optimization.py:389-392
        p, rho, r_prev, delta = None, dt.type(1.0), None, None           # direction_forget_factor == 0: reset (:82-83)
        for ii in range(num_cg):
            z = (r[0] / lf, r[1] / lP)                                   # M1, M2 = identity
            rho1, rho = rho, ip(r, z)
            if rho == 0:
                break
            if p is None:
                p = z
            else:
                beta = rho / rho1 if fletcher_reeves else (rho - ip(r_prev, z)) / rho1
                beta = max(beta, dt.type(0))
                p = (z[0] + p[0] * beta, z[1] + p[1] * beta)
            q = JT(*J(*p))
            alpha = rho / ip(p, q)
            if not fletcher_reeves:
                r_prev = r
            delta = (p[0] * alpha, p[1] * alpha) if delta is None else (delta[0] + p[0] * alpha, delta[1] + p[1] * alpha)
            if ii < num_cg - 1:
                r = (r[0] - q[0] * alpha, r[1] - q[1] * alpha)
        if delta is not None:
            f, P = f + delta[0], P + delta[1]
    return f, P


# --------------------------------------------------------------------------------------------
# LWL few-shot learner: ltr/models/meta/steepestdescent.py + ltr/models/lwl/loss_residual_modules.py
# --------------------------------------------------------------------------------------------

def lwl_gn_sd(w0, feat, label, sample_weight, *, num_iter, filter_reg, steplength_reg=0.0, compute_losses=True):
    """`GNSteepestDescent.forward` (steepestdescent.py:32-105) on `LWTLResidual` (loss_residual_modules.py:16-41),
    one sequence.

    w0 (F,C,K,K); feat (n,C,H,W); label (n,F,H,W); sample_weight None | (n,) | (n,F,H,W).
    Residuals r = [sw*(apply_filter(feat,w) - label), filter_reg*w]  (:24-41); the autograd calls of the reference
    (:70-73) are, for this linear residual,  g = J^T r = adj(feat, sw*r_data) + filter_reg*r_reg  and
    h = J g = [sw*apply_filter(feat,g), filter_reg*g];  alpha = |g|^2 / max(|h|^2 + steplength_reg*|g|^2, 1e-8) (:76-80).
    Returns (iterates (T+1,F,C,K,K), losses list); loss = sum r^2 / numel(r)  (:28-29).
    """
    dt = feat.dtype
    n = feat.shape[0]
    K = w0.shape[-1]
    if sample_weight is None:
        sw = dt.type(math.sqrt(1.0 / n))                                   # :27-28
    else:
        sw = np.asarray(sample_weight, dtype=dt)
        sw = sw.reshape(label.shape) if sw.size == label.size else sw.reshape(-1, 1, 1, 1)   # :29-33
    lam = dt.type(filter_reg)
    w = w0.astype(dt)
    iterates, losses = [w], []

    def residuals(w):
        return sw * (apply_filter(feat, w) - label), lam * w

    for _ in range(num_iter):
        rd, rr = residuals(w)
        if compute_losses:
            losses.append(((rd ** 2).sum() + (rr ** 2).sum()) / (rd.size + rr.size))
        g = apply_feat_transpose(feat, sw * rd, K) + lam * rr
        hd, hr = sw * apply_filter(feat, g), lam * g
        ip_gg = (g * g).sum()
        ip_hh = (hd * hd).sum() + (hr * hr).sum()
        alpha = ip_gg / max(ip_hh + dt.type(steplength_reg) * ip_gg, dt.type(1e-8))
        w = w - alpha * g
        iterates.append(w)
    if compute_losses:
        rd, rr = residuals(w)
        losses.append(((rd ** 2).sum() + (rr ** 2).sum()) / (rd.size + rr.size))
    return np.stack(iterates), losses


# --------------------------------------------------------------------------------------------
# ATOM: pytracking/libs/optimization.py (CG), pytracking/tracker/atom/optim.py (ConvProblem)
# --------------------------------------------------------------------------------------------

def mlu(x, min_val):
    """activation.py:20-29  MLU(x) = elu(leaky_relu(x, 1/min_val), min_val)."""
    y = np.where(x >= 0, x, x / min_val)
    return np.where(y > 0, y, min_val * (np.exp(np.minimum(y, 0)) - 1.0)).astype(x.dtype)


def mlu_deriv(x, min_val):
    """d MLU/dx: 1 for x>=0, exp(x/min_val) for x<0."""
    return np.where(x >= 0, 1.0, np.exp(np.minimum(x, 0) / min_val)).astype(x.dtype)


def atom_conv_residuals(x, samples, y, sample_weights, filter_reg, act_min_val):
    """`ConvProblem.__call__` (atom/optim.py:79-94): returns (data residuals (n,H,W), reg residual)."""
    H, W = samples.shape[-2:]
    s = apply_filter(samples, x, out_hw=(H, W))                          # conv2d mode='same'
    r = np.sqrt(sample_weights).reshape(-1, 1, 1) * (mlu(s, act_min_val) - y)
    return s, r, math.sqrt(filter_reg) * x


def atom_cg(x0, samples, y, sample_weights, *, filter_reg, act_min_val, num_iter,
            fletcher_reeves=False, state=None, direction_forget_factor=0.0):
    """`ConjugateGradient.run` + `ConjugateGradientBase.run_CG` for `ConvProblem`
    (optimization.py:72-163, 227-289; atom/optim.py:71-99), explicit J / J^T instead of autograd.

    x0 (C,K,K); samples (n,C,H,W); y (n,H,W); sample_weights (n,).
    state: dict(p, rho, r_prev) carried across calls (only used when direction_forget_factor != 0).
    Returns (x_new, state).
    """
    dt = samples.dtype
    if num_iter == 0:
        return x0, state
    H, W = samples.shape[-2:]
    K = x0.shape[-1]
    lam = dt.type(filter_reg)
    s0, r0, _ = atom_conv_residuals(x0, samples, y, sample_weights, filter_reg, act_min_val)
    d = np.sqrt(sample_weights).reshape(-1, 1, 1).astype(dt) * mlu_deriv(s0, act_min_val)

    def JT(rd, rr):
        return apply_feat_transpose(samples, d * rd, K) + dt.type(math.sqrt(filter_reg)) * rr

    def A(p):
        Jp = d * apply_filter(samples, p, out_hw=(H, W))
        return apply_feat_transpose(samples, d * Jp, K) + lam * p

    b = -JT(r0, dt.type(math.sqrt(filter_reg)) * x0)                     # optimization.py:262-265
    if direction_forget_factor == 0 or state is None:                    # :82-85
        p, rho, r_prev = None, dt.type(1.0), None
    else:
        p, rho, r_prev = state["p"], state["rho"], state["r_prev"]
        if p is not None:
            rho = rho / dt.type(direction_forget_factor)
    r = b.copy()
    delta = None
    for ii in range(num_iter):
        z = r                                                            # M1 = M2 = identity
        rho1 = rho
        rho = (r * z).sum()
        if rho == 0:                                                     # :108-113
            break
        if p is None:
            p = z.copy()
        else:
            if fletcher_reeves:
                beta = rho / rho1
            else:
                beta = (rho - (r_prev * z).sum()) / rho1                 # :121-122
            beta = max(beta, dt.type(0))                                 # :124
            p = z + p * beta
        q = A(p)
        pq = (p * q).sum()
        alpha = rho / pq                                                 # :131
        if not fletcher_reeves:
            r_prev = r.copy()
        delta = p * alpha if delta is None else delta + p * alpha
        if ii < num_iter - 1:                                            # :145-146
            r = r - q * alpha
    x = x0 if delta is None else x0 + delta
    return x, dict(p=p, rho=rho, r_prev=r_prev)


# --------------------------------------------------------------------------------------------
# Precise RoI Pooling (PARITY UNPINNED: submodule empty; see oracle/prroi_torch.py header)
# --------------------------------------------------------------------------------------------

def _G(u):
    t = np.clip(u, -1.0, 1.0)
    return np.where(t <= 0, 0.5 * (t + 1.0) ** 2, 1.0 - 0.5 * (1.0 - t) ** 2)


def _hat(u):
    return np.maximum(0.0, 1.0 - np.abs(u))


def _prroi_geometry(rois, PH, PW, scale, H, W):
    dt = rois.dtype
    x0, y0, x1, y1 = (rois[:, k] * dt.type(scale) for k in (1, 2, 3, 4))
    bw = np.maximum(x1 - x0, 0) / PW
    bh = np.maximum(y1 - y0, 0) / PH
    q = np.arange(PW, dtype=dt)
    p = np.arange(PH, dtype=dt)
    xs = x0[:, None] + q[None] * bw[:, None]
    xe = xs + bw[:, None]
    ys = y0[:, None] + p[None] * bh[:, None]
    ye = ys + bh[:, None]
    ii = np.arange(W, dtype=dt)
    jj = np.arange(H, dtype=dt)
    wx = _G(xe[:, :, None] - ii) - _G(xs[:, :, None] - ii)
    wy = _G(ye[:, :, None] - jj) - _G(ys[:, :, None] - jj)
    return xs, xe, ys, ye, bw, bh, wx.astype(dt), wy.astype(dt)


def prroi_forward(features, rois, PH, PW, scale):
    """out[r,c,p,q] = (1/A) * integral over bin (p,q) of the bilinear interpolant."""
    N, C, H, W = features.shape
    xs, xe, ys, ye, bw, bh, wx, wy = _prroi_geometry(rois, PH, PW, scale, H, W)
    area = bw * bh
    fr = features[rois[:, 0].astype(np.int64)]
    integ = np.einsum("rcji,rpj,rqi->rcpq", fr, wy, wx, optimize=True)
    safe = np.where(area > 0, area, 1.0)
    out = integ / safe[:, None, None, None]
    return np.where((area > 0)[:, None, None, None], out, 0.0).astype(features.dtype)


def prroi_backward_feat(grad_out, features_shape, rois, PH, PW, scale):
    """d out / d features: scatter of wy*wx/A (Appendix A)."""
    N, C, H, W = features_shape
    xs, xe, ys, ye, bw, bh, wx, wy = _prroi_geometry(rois, PH, PW, scale, H, W)
    area = bw * bh
    inv = np.where(area > 0, 1.0 / np.where(area > 0, area, 1.0), 0.0)
    contrib = np.einsum("rcpq,rpj,rqi,r->rcji", grad_out, wy, wx, inv, optimize=True)
    g = np.zeros(features_shape, dtype=grad_out.dtype)
    np.add.at(g, rois[:, 0].astype(np.int64), contrib)
    return g


def prroi_backward_coor(grad_out, features, rois, PH, PW, scale):
    """d out / d rois[:,1:5] via the boundary integrals of Appendix A.  Returns (R,5), column 0 = 0."""
    N, C, H, W = features.shape
    dt = features.dtype
    xs, xe, ys, ye, bw, bh, wx, wy = _prroi_geometry(rois, PH, PW, scale, H, W)
    area = bw * bh
    ok = area > 0
    inv = np.where(ok, 1.0 / np.where(ok, area, 1.0), 0.0)
    fr = features[rois[:, 0].astype(np.int64)]
    out = np.einsum("rcji,rpj,rqi->rcpq", fr, wy, wx, optimize=True) * inv[:, None, None, None]
    ii = np.arange(W, dtype=dt)
    jj = np.arange(H, dtype=dt)
    hx_s = _hat(xs[:, :, None] - ii)
    hx_e = _hat(xe[:, :, None] - ii)
    hy_s = _hat(ys[:, :, None] - jj)
    hy_e = _hat(ye[:, :, None] - jj)
    Lx_s = np.einsum("rcji,rpj,rqi->rcpq", fr, wy, hx_s, optimize=True)
    Lx_e = np.einsum("rcji,rpj,rqi->rcpq", fr, wy, hx_e, optimize=True)
    Ly_s = np.einsum("rcji,rpj,rqi->rcpq", fr, hy_s, wx, optimize=True)
    Ly_e = np.einsum("rcji,rpj,rqi->rcpq", fr, hy_e, wx, optimize=True)
    bh4 = bh[:, None, None, None]
    bw4 = bw[:, None, None, None]
    inv4 = inv[:, None, None, None]
    d_xs = (-Lx_s + bh4 * out) * inv4
    d_xe = (Lx_e - bh4 * out) * inv4
    d_ys = (-Ly_s + bw4 * out) * inv4
    d_ye = (Ly_e - bw4 * out) * inv4
    q = np.arange(PW, dtype=dt).reshape(1, 1, 1, PW)
    p = np.arange(PH, dtype=dt).reshape(1, 1, PH, 1)
    g = grad_out
    gr = np.zeros((rois.shape[0], 5), dtype=dt)
    gr[:, 1] = (g * (d_xs * (1 - q / PW) + d_xe * (1 - (q + 1) / PW))).sum(axis=(1, 2, 3))
    gr[:, 3] = (g * (d_xs * (q / PW) + d_xe * ((q + 1) / PW))).sum(axis=(1, 2, 3))
    gr[:, 2] = (g * (d_ys * (1 - p / PH) + d_ye * (1 - (p + 1) / PH))).sum(axis=(1, 2, 3))
    gr[:, 4] = (g * (d_ys * (p / PH) + d_ye * ((p + 1) / PH))).sum(axis=(1, 2, 3))
    return gr * dt.type(scale)


# --------------------------------------------------------------------------------------------
# classification-feature head: ltr/models/target_classifier/features.py:49-73 (num_blocks=0, final_conv, l2norm)
# --------------------------------------------------------------------------------------------

def clf_head(feat, weight, scale=1.0, eps=1e-5):
    """Conv2d(Cin, Cout, 3, padding=1, bias=False) (features.py:66) followed by InstanceL2Norm(size_average=True)
    (normalization.py:15-18): x * scale * sqrt(C*H*W / (sum_{c,h,w} x^2 + eps)) per image.
      feat (n,Cin,H,W), weight (Cout,Cin,3,3) -> (n,Cout,H,W)."""
    x = apply_filter(feat, weight)
    n, C, H, W = x.shape
    ss = (x * x).reshape(n, -1).sum(axis=1).reshape(n, 1, 1, 1)
    return x * (scale * np.sqrt((C * H * W) / (ss + eps)))


# --------------------------------------------------------------------------------------------
# target localisation: pytracking/libs/dcf.py:156-164, pytracking/tracker/dimp/dimp.py:238-303
# --------------------------------------------------------------------------------------------

def max2d(a):
    """dcf.max2d for one (H, W) map: maximum over rows, then over columns, first index on ties."""
    rows = a.argmax(axis=0)                       # numpy argmax returns the first maximum, like torch.max
    col = int(a.max(axis=0).argmax())
    return a[rows[col], col], (int(rows[col]), col)


def two_peaks(scores, scores_hn, neigh):
    """The 8 numbers `localize_advanced` derives from the score maps (dimp.py:252-281): scores / scores_hn (S,H,W),
    neigh (S,2) target neighbourhood in cells.  Python round() (half to even) on doubles for the bounds."""
    S, H, W = scores.shape
    peaks = [max2d(scores[s]) for s in range(S)]
    s1 = int(np.argmax(np.array([p[0] for p in peaks], dtype=np.float32)))
    m1, (r1, c1) = peaks[s1]
    top = max(round(float(r1) - float(neigh[s1][0]) / 2), 0)
    bottom = min(round(float(r1) + float(neigh[s1][0]) / 2 + 1), H)
    left = max(round(float(c1) - float(neigh[s1][1]) / 2), 0)
    right = min(round(float(c1) + float(neigh[s1][1]) / 2 + 1), W)
    masked = np.array(scores_hn[s1], copy=True)
    masked[top:bottom, left:right] = 0
    m2, (r2, c2) = max2d(masked)
    return np.array([m1, r1, c1, s1, m2, r2, c2, 0], dtype=np.float64)


LOC_FLAGS = ("normal", "hard_negative", "uncertain", "not_found")


def localize_decide(scores, scores_hn, q):
    """Outcome of `DiMP.localize_advanced` (pytracking/tracker/dimp/dimp.py:252-303) from the two peaks, as the 16 floats
    `pt_localize_decide_f32` leaves: [code, scale, row, col, translation_row, translation_col, max1, row1, col1, max2,
    row2, col2, pick, 0, 0, 0].  `q`: the fields of `pt_localize_params` (include/pt_hot.h) -- thresholds, map centre,
    support / output ratio and per scale: sample scale, neighbourhood (dimp.py:268), previous target offset (dimp.py:285).
    Types as the reference's expressions evaluate under torch: `.item() < float` in double (dimp.py:258-263), everything
    else float32 with Python scalars rounded to float32 first (dimp.py:288-301)."""
    f32 = np.float32
    v = two_peaks(scores, scores_hn, list(zip(q["neigh_r"], q["neigh_c"])))
    m1, r1, c1, s1, m2, r2, c2 = f32(v[0]), int(v[1]), int(v[2]), int(v[3]), f32(v[4]), int(v[5]), int(v[6])
    ctr = (f32(q["center_r"]), f32(q["center_c"]))
    d1 = (f32(r1) - ctr[0], f32(c1) - ctr[1])                                  # target_disp1, dimp.py:255
    d2 = (f32(r2) - ctr[0], f32(c2) - ctr[1])                                  # target_disp2, dimp.py:281
    pick = 1
    if float(m1) < q["target_not_found_threshold"]:
        code = 3
    elif float(m1) < q["uncertain_threshold"]:
        code = 2
    elif float(m1) < q["hard_sample_threshold"]:
        code = 1
    elif m2 > f32(q["distractor_threshold"]) * m1:                             # dimp.py:288
        prev = (f32(q["prev_r"][s1]), f32(q["prev_c"][s1]))
        n1 = np.sqrt((d1[0] - prev[0]) ** 2 + (d1[1] - prev[1]) ** 2, dtype=f32)
        n2 = np.sqrt((d2[0] - prev[0]) ** 2 + (d2[1] - prev[1]) ** 2, dtype=f32)
        thr = f32(q["disp_threshold"])
        if n2 > thr and n1 < thr:
            code = 1
        elif n2 < thr and n1 > thr:
            code, pick = 1, 2
        else:
            code = 2
    elif m2 > f32(q["hard_negative_threshold"]) * m1 and m2 > f32(q["target_not_found_f32"]):
        code = 1
    else:
        code = 0
    d, rc = (d1, (r1, c1)) if pick == 1 else (d2, (r2, c2))
    sc = f32(q["scale"][s1])
    tv = (d[0] * f32(q["ratio_r"]) * sc, d[1] * f32(q["ratio_c"]) * sc)       # dimp.py:256 / :282
    return np.array([code, s1, rc[0], rc[1], tv[0], tv[1], m1, r1, c1, m2, r2, c2, pick, 0, 0, 0], dtype=np.float64)


# --------------------------------------------------------------------------------------------
# image-patch sampling (pytracking/features/preprocessing.py:54-148), pixel part
# --------------------------------------------------------------------------------------------
def _fma32(a, b, c):
    """float32 fused multiply-add: the product of two float32 values is exact in float64; one rounding of the sum.
    (The float64 sum is itself rounded before the final rounding -- a double rounding that needs a tie at the 29th extra bit;
    it has not occurred on any golden.)"""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def sample_patch_pixels(im, df, os0, os1, tl0, tl1, crop_h, crop_w, out_hw):
    """Strided view `im[..., os0::df, os1::df]`, crop [tl, tl + crop) with replicate padding, bilinear resize with
    `F.interpolate(mode='bilinear', align_corners=False)` semantics as ATen's CPU kernel executes them in this image
    (UpSampleKernel.cpp, generic separable path): scale = in / out in float32, src = fma(scale, dst + 0.5, -0.5) clamped at 0
    -- the build contracts that expression into ONE fused multiply-add (found by matching its output: with the unfused form
    the weights of a 403 -> 576 resize are an ulp of src ~ 3e-5 off and pixels move by up to 3.6e-3, with the fused form the
    restatement is within 5e-5 = 3 ulp of a 0..255 pixel) -- lambda1 = src - floor(src), row blends then the column blend.
    How the compiler contracted the BLENDS differs between the kernel's loop specialisations (shape dependent; 17 of the 24
    golden patches are bit-identical with either form), so they are left unfused here.  im (C,H,W) float32 -> (C,oh,ow)."""
    f32 = np.float32
    im2 = im[:, os0::df, os1::df]
    H2, W2 = im2.shape[1:]
    rows = np.clip(tl0 + np.arange(crop_h), 0, H2 - 1)
    cols = np.clip(tl1 + np.arange(crop_w), 0, W2 - 1)
    patch = im2[:, rows][:, :, cols].astype(f32)
    oh, ow = out_hw
    if (oh, ow) == (crop_h, crop_w):
        return patch                                                     # preprocessing.py:139-140: no resampling

    def axis(n_in, n_out):
        if n_in == n_out:                                                # ATen: scale factor 1 copies (lambda0 = 1, lambda1 = 0)
            i0 = np.arange(n_out)
            return i0, i0, np.ones(n_out, f32), np.zeros(n_out, f32)
        scale = f32(n_in) / f32(n_out)
        src = _fma32(scale, np.arange(n_out, dtype=f32) + f32(0.5), f32(-0.5))
        src = np.maximum(src, f32(0))
        i0 = np.minimum(src.astype(np.int64), n_in - 1)
        i1 = i0 + (i0 < n_in - 1)
        l1 = np.clip(src - i0.astype(f32), f32(0), f32(1)).astype(f32)
        return i0, i1, (f32(1) - l1).astype(f32), l1
    y0, y1, ly0, ly1 = axis(crop_h, oh)
    x0, x1, lx0, lx1 = axis(crop_w, ow)
    top = patch[:, y0][:, :, x0] * lx0 + patch[:, y0][:, :, x1] * lx1
    bot = patch[:, y1][:, :, x0] * lx0 + patch[:, y1][:, :, x1] * lx1
    return (top * ly0[None, :, None] + bot * ly1[None, :, None]).astype(f32)


# --------------------------------------------------------------------------------------------
# first-frame augmentation set (pytracking/features/augmentation.py:11-147, preprocessing.py:13-30)
# --------------------------------------------------------------------------------------------
def aug_crop_to_output(img, output_sz, shift):
    """`Transform.crop_to_output` (augmentation.py:20-37): F.pad(mode='replicate') with pads floor/ceil((out - in) / 2) +- shift;
    negative pads crop.  img (C, H, W)."""
    H, W = img.shape[1:]
    pad_h = 0.0 if output_sz is None else (output_sz[0] - H) / 2
    pad_w = 0.0 if output_sz is None else (output_sz[1] - W) / 2
    top, bottom = math.floor(pad_h) + shift[0], math.ceil(pad_h) - shift[0]
    left, right = math.floor(pad_w) + shift[1], math.ceil(pad_w) - shift[1]
    rows = np.clip(np.arange(H + top + bottom) - top, 0, H - 1)
    cols = np.clip(np.arange(W + left + right) - left, 0, W - 1)
    return img[:, rows][:, :, cols]


def aug_blur(img, f0, f1):
    """`Blur.__call__` (:142-147): correlation with f0 along the rows (zero padding), the float32 result correlated with f1
    along the columns.  Accumulation in float32, taps in ascending order."""
    f32 = np.float32
    C, H, W = img.shape
    fs0, fs1 = (len(f0) - 1) // 2, (len(f1) - 1) // 2
    p = np.zeros((C, H + 2 * fs0, W), f32)
    p[:, fs0:fs0 + H] = img
    im1 = np.zeros((C, H, W), f32)
    for i in range(2 * fs0 + 1):
        im1 = (im1 + f32(f0[i]) * p[:, i:i + H]).astype(f32)
    p = np.zeros((C, H, W + 2 * fs1), f32)
    p[:, :, fs1:fs1 + W] = im1
    out = np.zeros((C, H, W), f32)
    for j in range(2 * fs1 + 1):
        out = (out + f32(f1[j]) * p[:, :, j:j + W]).astype(f32)
    return out


def aug_scale_size(h_orig, scale_factor):
    h_new = round(h_orig / scale_factor)                                   # :84-87
    return h_new + (h_new - h_orig) % 2


def aug_rotate(img, angle_deg):
    """`Rotate.__call__` (:118-126) = cv2.warpAffine(image, [R | c - R c], (W, H), INTER_LINEAR, BORDER_REPLICATE), restated
    from OpenCV's published algorithm (imgwarp.cpp: the matrix is inverted in double, source coordinates are formed in
    fixed point with 10 fractional bits, rounded to 1/32 pixel, bilinear weights k/32 in float).  PARITY UNPINNED: cv2 is
    not installed in the build container, no golden exists for this transform."""
    f32 = np.float32
    C, H, W = img.shape
    a = math.pi * angle_deg / 180
    ca, sa = math.cos(a), math.sin(a)
    c0, c1 = (H - 1) / 2, (W - 1) / 2
    M = [ca, sa, c0 - (ca * c0 + sa * c1), -sa, ca, c1 - (-sa * c0 + ca * c1)]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    m0, m1, m3, m4 = M[4] * D, M[1] * (-D), M[3] * (-D), M[0] * D
    m2, m5 = -m0 * M[2] - m1 * M[5], -m3 * M[2] - m4 * M[5]
    xs, ys = np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64)
    X = (np.rint(m0 * xs * 1024.0).astype(np.int64)[None, :] + (np.rint((m1 * ys + m2) * 1024.0).astype(np.int64) + 16)[:, None]) >> 5
    Y = (np.rint(m3 * xs * 1024.0).astype(np.int64)[None, :] + (np.rint((m4 * ys + m5) * 1024.0).astype(np.int64) + 16)[:, None]) >> 5
    ix, iy = X >> 5, Y >> 5
    ax, ay = (X & 31).astype(f32) / f32(32), (Y & 31).astype(f32) / f32(32)
    cy = lambda v: np.clip(v, 0, H - 1)
    cx = lambda v: np.clip(v, 0, W - 1)
    p00, p01 = img[:, cy(iy), cx(ix)], img[:, cy(iy), cx(ix + 1)]
    p10, p11 = img[:, cy(iy + 1), cx(ix)], img[:, cy(iy + 1), cx(ix + 1)]
    one = f32(1)
    return (p00 * ((one - ax) * (one - ay)) + p01 * (ax * (one - ay)) + p10 * ((one - ax) * ay) + p11 * (ax * ay)).astype(f32)


def augment_patch(patch, specs):
    """`torch.cat([T(im_patch) for T in transforms])` (preprocessing.py:28).  patch (C, EH, EW) float32; specs: list of dicts
    {kind: identity | fliplr | flipud | blur | scale | rotate, output_sz, shift, f0 / f1, scale_factor, angle}."""
    outs = []
    for sp in specs:
        k = sp["kind"]
        if k == "identity":
            img = patch
        elif k == "fliplr":
            img = patch[:, :, ::-1]
        elif k == "flipud":
            img = patch[:, ::-1]
        elif k == "blur":
            img = aug_blur(patch, sp["f0"], sp["f1"])
        elif k == "scale":
            hn, wn = aug_scale_size(patch.shape[1], sp["scale_factor"]), aug_scale_size(patch.shape[2], sp["scale_factor"])
            img = sample_patch_pixels(patch, 1, 0, 0, 0, 0, patch.shape[1], patch.shape[2], (hn, wn))
        elif k == "rotate":
            img = aug_rotate(patch, sp["angle"])
        else:
            raise ValueError(k)
        outs.append(aug_crop_to_output(img, sp.get("output_sz"), sp.get("shift", (0, 0))))
    return np.stack(outs).astype(np.float32)
