"""Host-side mirror of the image-patch sampling in front of the backbone: `pytracking/features/preprocessing.py`
`sample_patch` (:54-148) and `sample_patch_multiscale` (:33-51), for images that already live on the device.

The reference crops on the CPU (strided view, F.pad, F.interpolate) and uploads every patch; here the frame is uploaded
once and crop + replicate padding + bilinear resize are ONE gather launch for all scales (pt_sample_patch_f32).  The
integer geometry -- pre-downsampling stride, crop corners, the 'inside' / 'inside_major' shifts -- decides which pixels
are read and is therefore computed here on the host with the reference's own rules, in plain Python integers.
`sample_patch_transformed` (:13-30) -- the first-frame augmentation set built by `generate_init_samples`
(pytracking/tracker/dimp/dimp.py:329-395) -- runs as a second gather launch over the base patch for the whole transform list
(pt_augment_patches_f32): the reference's own `augmentation.Transform` objects are read for their parameters (shift, output
size, blur taps, scale factor, angle), never called.  `Rotate` follows OpenCV's published warpAffine arithmetic and is
PARITY-UNPINNED (cv2 is not installed where the goldens are generated); everything else is pinned by `augment.npz`.
Masks (nearest resampling, zero padding) are not on the per-frame path and are not covered (NotImplementedError;
`pytracking_amd.install` hands such calls to the reference).
"""
import ctypes
import math

import torch

from . import _lib
from .filter import _ptr, _require_device, _stream, device_guarded


def _round_half_even(v):
    return int(round(v))                                        # torch.round and Python's round both go to even


def patch_geometry(im_hw, pos, sample_sz, output_sz, mode='replicate', max_scale_change=None):
    """The integers `sample_patch` derives before touching pixels (preprocessing.py:69-128).
    Returns (_lib.PatchGeom, patch_coord [y0, x0, y1, x1] in image pixels, floats like the reference's)."""
    H, W = int(im_hw[0]), int(im_hw[1])
    posl = [int(pos[0]), int(pos[1])]                           # .long(): truncation
    ssz = [float(sample_sz[0]), float(sample_sz[1])]
    if mode in ('inside', 'inside_major'):
        ratio = [ssz[0] / H, ssz[1] / W]
        shrink = max(ratio) if mode == 'inside' else min(ratio)
        shrink = max(shrink, 1.0)
        if max_scale_change is not None:
            shrink = min(shrink, float(max_scale_change))
        shrink = float(torch.tensor(shrink, dtype=torch.float32))            # the reference divides in float32
        ssz = [float(int(torch.tensor(v, dtype=torch.float32) / shrink)) for v in ssz]      # .long()
    elif mode != 'replicate':
        raise ValueError("Unknown border mode '{}'.".format(mode))
    if output_sz is not None:
        resize_factor = min(float(torch.tensor(ssz[0], dtype=torch.float32) / float(output_sz[0])),
                            float(torch.tensor(ssz[1], dtype=torch.float32) / float(output_sz[1])))
        df = int(max(int(resize_factor - 0.1), 1))
    else:
        df = 1
    sz = [float(torch.tensor(v, dtype=torch.float32) / df) for v in ssz]
    if df > 1:
        os_ = [posl[0] % df, posl[1] % df]
        posl = [(posl[0] - os_[0]) // df, (posl[1] - os_[1]) // df]     # exact: divisible
        H2, W2 = len(range(os_[0], H, df)), len(range(os_[1], W, df))
    else:
        os_, H2, W2 = [0, 0], H, W
    szl = [max(_round_half_even(v), 2) for v in sz]
    # Under torch >= 1.5 `/` on LongTensors is true division: the reference's corners are FLOATS (x.5 for even crop
    # sizes), shifted as floats in the 'inside' modes, and truncated towards zero only when the padding is formed
    # (preprocessing.py:108-128 as executed by this image's torch -- that execution is the oracle).
    tl = [posl[k] - (szl[k] - 1) / 2 for k in range(2)]
    br = [posl[k] + szl[k] / 2 + 1 for k in range(2)]
    if mode in ('inside', 'inside_major'):
        lim = [H2, W2]
        for k in range(2):
            shift = max(-tl[k], 0.0) - max(br[k] - lim[k], 0.0)
            tl[k] += shift
            br[k] += shift
        for k in range(2):
            outside = math.floor((max(-tl[k], 0.0) + max(br[k] - lim[k], 0.0)) / 2)
            shift = (-tl[k] - outside) * (1 if outside > 0 else 0)
            tl[k] += shift
            br[k] += shift
    ti = [int(tl[0]), int(tl[1])]                               # .int(): towards zero
    bi = [int(br[0]), int(br[1])]
    g = _lib.PatchGeom(df, os_[0], os_[1], ti[0], ti[1], bi[0] - ti[0], bi[1] - ti[1])
    coord = [df * tl[0], df * tl[1], df * br[0], df * br[1]]
    return g, coord


@device_guarded
def sample_patch_multiscale(im, pos, scales, image_sz, mode='replicate', max_scale_change=None):
    """(im_patches (S,C,oh,ow) on the device, patch_coords (S,4) on the host) -- preprocessing.py:33-51."""
    _require_device(im)
    if isinstance(scales, (int, float)):
        scales = [scales]
    if im.dim() != 4 or im.shape[0] != 1:
        raise NotImplementedError("sample_patch: one (1, C, H, W) image")
    oh, ow = int(image_sz[0]), int(image_sz[1])
    S = len(scales)
    if S > 8:
        raise NotImplementedError("more than 8 scales per call")
    geoms, coords = (_lib.PatchGeom * S)(), []
    for k, s in enumerate(scales):
        s = float(s)
        g, c = patch_geometry(im.shape[-2:], pos, [s * float(image_sz[0]), s * float(image_sz[1])], image_sz, mode,
                              max_scale_change)
        geoms[k] = g
        coords.append(c)
    im = im.contiguous()
    C, H, W = im.shape[1:]
    out = torch.empty((S, C, oh, ow), dtype=torch.float32, device=im.device)
    rc = _lib.lib().pt_sample_patch_f32(_ptr(im), C, H, W, geoms, S, _ptr(out), oh, ow, _stream())
    _lib.check(rc, "pt_sample_patch_f32")
    return out, torch.tensor(coords, dtype=torch.float32)


@device_guarded
def sample_patch(im, pos, sample_sz, output_sz=None, mode='replicate', max_scale_change=None, is_mask=False):
    """preprocessing.py:54-148 for a device image: (im_patch (1,C,oh,ow), patch_coord (1,4))."""
    if is_mask:
        raise NotImplementedError("mask patches (nearest resampling) are not on the per-frame path")
    _require_device(im)
    if im.dim() != 4 or im.shape[0] != 1:
        raise NotImplementedError("sample_patch takes one image (1, C, H, W)")
    g, c = patch_geometry(im.shape[-2:], pos, sample_sz, output_sz, mode, max_scale_change)
    oh, ow = (g.crop_h, g.crop_w) if output_sz is None else (int(output_sz[0]), int(output_sz[1]))
    im = im.contiguous()
    C, H, W = im.shape[1:]
    out = torch.empty((1, C, oh, ow), dtype=torch.float32, device=im.device)
    rc = _lib.lib().pt_sample_patch_f32(_ptr(im), C, H, W, (_lib.PatchGeom * 1)(g), 1, _ptr(out), oh, ow, _stream())
    _lib.check(rc, "pt_sample_patch_f32")
    return out, torch.tensor([c], dtype=torch.float32)


# ---------------------------------------------------------------------------------------------------------------------
# first-frame augmentation set
# ---------------------------------------------------------------------------------------------------------------------
def _invert_affine(M):
    """cv::invertAffineTransform on a 2x3 matrix given as 6 Python floats (double arithmetic, OpenCV's operation order)."""
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    m0, m1, m3, m4 = A11, M[1] * (-D), M[3] * (-D), A22
    return [m0, m1, -m0 * M[2] - m1 * M[5], m3, m4, -m3 * M[2] - m4 * M[5]]


def transform_descriptors(transforms, patch_hw):
    """(AugDesc array, taps list, (OH, OW)) for a list of the reference's transform objects (augmentation.py) applied to a
    patch of size `patch_hw`.  Classes are recognised by name so that the reference module is not imported here."""
    EH, EW = int(patch_hw[0]), int(patch_hw[1])
    descs = (_lib.AugDesc * len(transforms))()
    taps, out_hw = [], None
    for k, T in enumerate(transforms):
        name = type(T).__name__
        d = descs[k]
        th, tw = EH, EW
        if name in ("Identity", "Translation"):
            d.kind = _lib.PT_AUG_IDENTITY
        elif name == "FlipHorizontal":
            d.kind = _lib.PT_AUG_FLIP_H
        elif name == "FlipVertical":
            d.kind = _lib.PT_AUG_FLIP_V
        elif name == "Blur":                                             # augmentation.py:128-147
            d.kind = _lib.PT_AUG_BLUR
            d.fs0, d.fs1 = int(T.filter_size[0]), int(T.filter_size[1])
            f0 = [float(v) for v in T.filter[0].reshape(-1)]
            f1 = [float(v) for v in T.filter[1].reshape(-1)]
            if len(f0) != 2 * d.fs0 + 1 or len(f1) != 2 * d.fs1 + 1:
                raise ValueError("Blur: filter length does not match filter_size")
            d.tap_off0 = len(taps)
            taps += f0
            d.tap_off1 = len(taps)
            taps += f1
        elif name == "Scale":                                            # :80-92
            if EH != EW:
                raise NotImplementedError
            d.kind = _lib.PT_AUG_SCALE
            th = round(EH / T.scale_factor)
            th += (th - EH) % 2
            tw = round(EW / T.scale_factor)
            tw += (tw - EW) % 2
        elif name == "Rotate":                                           # :118-126 (UNPINNED: OpenCV arithmetic restated)
            d.kind = _lib.PT_AUG_ROTATE
            ca, sa = math.cos(T.angle), math.sin(T.angle)
            c0, c1 = (EH - 1) / 2, (EW - 1) / 2                          # the reference's centre vector (rows first, as written)
            M = [ca, sa, c0 - (ca * c0 + sa * c1), -sa, ca, c1 - (-sa * c0 + ca * c1)]
            inv = _invert_affine(M)
            for q in range(6):
                d.m[q] = inv[q]
        else:
            raise NotImplementedError("augmentation transform '%s' is not covered on the device" % name)
        d.th, d.tw = th, tw
        osz = getattr(T, "output_sz", None)
        sh = getattr(T, "shift", (0, 0))
        if osz is None:
            pad_h = pad_w = 0.0
        else:
            pad_h, pad_w = (osz[0] - th) / 2, (osz[1] - tw) / 2
        d.pad_top = math.floor(pad_h) + int(sh[0])                       # :30-33
        d.pad_left = math.floor(pad_w) + int(sh[1])
        hw = (th + math.floor(pad_h) + math.ceil(pad_h), tw + math.floor(pad_w) + math.ceil(pad_w))
        if out_hw is None:
            out_hw = hw
        elif hw != out_hw:
            raise ValueError("transforms produce different output sizes (torch.cat would fail in the reference)")
    if len(taps) > _lib.PT_AUG_MAX_TAPS:
        raise NotImplementedError("more than %d blur taps in one transform list" % _lib.PT_AUG_MAX_TAPS)
    return descs, taps, out_hw


@device_guarded
def augment_patch(im_patch, transforms):
    """`torch.cat([T(im_patch) for T in transforms])` (preprocessing.py:28) for a device patch (1, C, EH, EW)."""
    _require_device(im_patch)
    if im_patch.dim() != 4 or im_patch.shape[0] != 1:
        raise NotImplementedError("augment_patch takes one patch (1, C, H, W)")
    if len(transforms) == 0:
        raise ValueError("empty transform list")
    C, EH, EW = (int(v) for v in im_patch.shape[1:])
    descs, taps, (oh, ow) = transform_descriptors(transforms, (EH, EW))
    if oh < 1 or ow < 1:
        raise ValueError("transforms crop the patch to nothing")
    im_patch = im_patch.contiguous()
    out = torch.empty((len(transforms), C, oh, ow), dtype=torch.float32, device=im_patch.device)
    tap_arr = (ctypes.c_float * max(len(taps), 1))(*taps)
    rc = _lib.lib().pt_augment_patches_f32(_ptr(im_patch), C, EH, EW, descs, len(transforms), tap_arr, len(taps), _ptr(out), oh, ow,
                                           _stream())
    _lib.check(rc, "pt_augment_patches_f32")
    return out


@device_guarded
def sample_patch_transformed(im, pos, scale, image_sz, transforms, is_mask=False):
    """preprocessing.py:13-30 for a device image: base patch at `image_sz`, then every transform of the list."""
    if is_mask:
        raise NotImplementedError("mask patches (nearest resampling) are not on the per-frame path")
    im_patch, _ = sample_patch(im, pos, scale * image_sz, image_sz)
    return augment_patch(im_patch, transforms)
