"""Readers for tests/golden/augment.npz (oracle/make_golden.py: gen_augment): the transform lists of the reference runs as
(a) oracle specs and (b) stand-ins for the reference's `pytracking.features.augmentation` objects -- same class names and
attributes, no behaviour -- so that the product's `transform_descriptors` can be exercised where /root/reference is absent."""
import math

import numpy as np
import torch


class _Tr:
    def __init__(self, output_sz, shift):
        self.output_sz, self.shift = output_sz, shift


class Identity(_Tr):
    pass


class Translation(_Tr):
    pass


class FlipHorizontal(_Tr):
    pass


class FlipVertical(_Tr):
    pass


class Blur(_Tr):
    pass


class Scale(_Tr):
    pass


class Rotate(_Tr):
    pass


_CLS = {c.__name__: c for c in (Identity, Translation, FlipHorizontal, FlipVertical, Blur, Scale, Rotate)}
_KIND = {"Identity": "identity", "Translation": "identity", "FlipHorizontal": "fliplr", "FlipVertical": "flipud", "Blur": "blur",
         "Scale": "scale", "Rotate": "rotate"}


def read_case(g, tag):
    """-> dict(im (1,C,H,W) float32, pos, scale, image_sz, stride, out, shape, sums, specs (oracle), objs (stand-ins))."""
    specs, objs = [], []
    for k in range(int(g[f"{tag}_n"])):
        cls = str(g[f"{tag}_t{k}_cls"])
        osz = g[f"{tag}_t{k}_output_sz"]
        osz = None if osz.ndim == 0 else [int(v) for v in osz]
        shift = tuple(int(v) for v in g[f"{tag}_t{k}_shift"])
        sp = {"kind": _KIND[cls], "output_sz": osz, "shift": shift}
        ob = _CLS[cls](osz, shift)
        if cls == "Blur":
            sp["f0"], sp["f1"] = g[f"{tag}_t{k}_f0"], g[f"{tag}_t{k}_f1"]
            ob.filter = [torch.from_numpy(sp["f0"]).view(1, 1, -1, 1), torch.from_numpy(sp["f1"]).view(1, 1, 1, -1)]
            ob.filter_size = [(len(sp["f0"]) - 1) // 2, (len(sp["f1"]) - 1) // 2]
        if cls == "Scale":
            sp["scale_factor"] = ob.scale_factor = float(g[f"{tag}_t{k}_scale_factor"])
        specs.append(sp)
        objs.append(ob)
    return dict(im=g[f"{tag}_im"].astype(np.float32), pos=g[f"{tag}_pos"], scale=float(g[f"{tag}_scale"]),
                image_sz=g[f"{tag}_image_sz"], stride=int(g[f"{tag}_stride"]), out=g[f"{tag}_out"],
                shape=tuple(int(v) for v in g[f"{tag}_shape"]), sums=g[f"{tag}_sums"], specs=specs, objs=objs)


def rotate_pair(angle_deg, output_sz, shift):
    """(oracle spec, stand-in object) of a `Rotate(angle)` (augmentation.py:111-116) -- unpinned transform."""
    ob = Rotate(output_sz, shift)
    ob.angle = math.pi * angle_deg / 180
    return {"kind": "rotate", "angle": angle_deg, "output_sz": output_sz, "shift": shift}, ob
