"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Import shim that lets the *unmodified* reference (visionml/pytracking mounted read-only at
/root/reference) execute on CPU inside the build container so that golden vectors can be
generated from the reference's own code.  /root/reference is absent on
the GPU box: there the same shim resolves to oracle/_ref/reference, the byte-for-byte bundle oracle/make_ref_bundle.py
writes (git-ignored, sha256 manifest), used by tests/test_trackers_on_device.py and bench.py's baseline legs;
tests/golden/*.npz stay the portable artefact of every other test (oracle/make_golden.py).

What it does (none of it edits /root/reference):
  * stubs the third-party imports the reference pulls in but the path never executes
    (cv2, visdom, torchvision, jpeg4py, ... -- pytracking/__init__.py:7-8, evaluation/tracker.py:7);
  * injects pytracking.evaluation.local / ltr.admin.local so env_settings() never writes
    local.py into the reference tree (evaluation/environment.py:31-68, admin/environment.py:6-56);
  * provides ltr.external.PreciseRoIPooling.pytorch.prroi_pool (git submodule that is EMPTY in the
    snapshot) from oracle/prroi_torch.py, the autograd restatement of Appendix A;
  * fixes TensorList.__getattr__ so torch>=2 does not mistake it for a __torch_function__ carrier
    (pytracking/libs/tensorlist.py:173-180).
"""
import os
import sys
import types
from unittest import mock

_BUNDLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "reference")


def _resolve_root():
    """$PYTRACKING_REFERENCE, else the mounted tree (build container), else the recipe-built byte-for-byte bundle
    oracle/_ref/reference (oracle/make_ref_bundle.py: git-ignored, travels to the GPU box with the snapshot)."""
    env = os.environ.get("PYTRACKING_REFERENCE")
    if env:
        return env
    if os.path.isdir("/root/reference/ltr"):
        return "/root/reference"
    return _BUNDLE


REFERENCE_ROOT = _resolve_root()

_STUBS = [
    "cv2", "visdom", "visdom.server", "jpeg4py", "torchvision", "torchvision.models",
    "torchvision.models.resnet", "torchvision.ops", "torchvision.transforms", "pycocotools",
    "pycocotools.coco", "pycocotools.mask", "lvis", "tikzplotlib", "skimage", "tensorboardX",
    "timm", "timm.models", "timm.models.layers", "matplotlib", "matplotlib.pyplot",
    "matplotlib.patches", "pandas",
]


class _Permissive(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = mock.MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, m)
        return m


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "ltr"))


_installed = False


def install():
    """Make `import ltr...` / `import pytracking...` resolve to the reference, side-effect free."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    for name in _STUBS:
        if name in sys.modules:
            continue
        try:
            if name in ("matplotlib", "matplotlib.pyplot", "matplotlib.patches", "pandas"):
                __import__(name)
                continue
        except Exception:
            pass
        mod = _Permissive(name)
        mod.__path__ = []
        sys.modules[name] = mod
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    # PrRoIPool restatement under the reference's import path (initializer.py:4, atom_iou_net.py:4)
    here = os.path.dirname(os.path.abspath(__file__))
    if os.path.dirname(here) not in sys.path:
        sys.path.insert(0, os.path.dirname(here))
    from oracle import prroi_torch
    import ltr  # noqa: F401  (real package from the reference)
    for pkg in ("ltr.external", "ltr.external.PreciseRoIPooling", "ltr.external.PreciseRoIPooling.pytorch"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    pm = types.ModuleType("ltr.external.PreciseRoIPooling.pytorch.prroi_pool")
    pm.PrRoIPool2D = prroi_torch.PrRoIPool2D
    sys.modules[pm.__name__] = pm

    # local env modules so nothing is written into the reference tree
    lm = types.ModuleType("pytracking.evaluation.local")

    def local_env_settings():
        from pytracking.evaluation.environment import EnvSettings
        return EnvSettings()
    lm.local_env_settings = local_env_settings
    sys.modules[lm.__name__] = lm
    am = types.ModuleType("ltr.admin.local")

    class EnvironmentSettings:
        def __init__(self):
            self.workspace_dir = "/tmp/ltr_ws"
            self.tensorboard_dir = "/tmp/ltr_ws/tb"
            self.pretrained_networks = "/tmp/ltr_ws/nets"
    am.EnvironmentSettings = EnvironmentSettings
    sys.modules[am.__name__] = am

    # TensorList vs torch>=2 (__torch_function__ probe)
    from pytracking.libs import tensorlist as _tl
    _orig = _tl.TensorList.__getattr__

    def _safe_getattr(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _orig(self, name)
    _tl.TensorList.__getattr__ = _safe_getattr

    # torch.rfft / torch.irfft (removed in torch 1.8) as pytracking/libs/fourier.py:24,31 call them (ATOM's localisation):
    # onesided real FFT over the last `signal_ndim` dimensions, complex numbers as a trailing dimension of 2
    import torch
    if not hasattr(torch, "rfft") or isinstance(getattr(torch, "rfft", None), types.ModuleType):
        def _rfft(a, signal_ndim, normalized=False, onesided=True):
            assert onesided and not normalized
            return torch.view_as_real(torch.fft.rfftn(a, dim=tuple(range(-signal_ndim, 0))))

        def _irfft(a, signal_ndim, normalized=False, onesided=True, signal_sizes=None):
            assert onesided and not normalized
            return torch.fft.irfftn(torch.view_as_complex(a.contiguous()), s=signal_sizes, dim=tuple(range(-signal_ndim, 0)))
        torch.rfft, torch.irfft = _rfft, _irfft
    _installed = True
