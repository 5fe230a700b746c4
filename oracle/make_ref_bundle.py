"""TEST INFRASTRUCTURE ONLY -- recipe that puts the UNMODIFIED reference where the GPU box can import it.

    python -B oracle/make_ref_bundle.py            (build container only: needs /root/reference)

/root/reference does not exist on the GPU box; `oracle/_ref/` is git-ignored (nothing of the reference enters the history)
but NOT gpurun-ignored, so what this recipe writes there travels with the snapshot like the built libpt_hot.so.  It copies
the Python sources of the two packages the trackers import -- `ltr/` and `pytracking/` -- byte for byte into
`oracle/_ref/reference/`, leaves out what the hot path never touches (dataset loaders and their spec lists, training
settings / actors / trainers, notebooks, analysis, VOT glue, util scripts), and writes `MANIFEST.json` with the sha256 of
every file so that `verify()` (tests/test_ref_bundle_cpu.py) can show the bundle IS the reference, not an edited copy.

What uses it (checker / baseline legs only, never the product path):
  * tests/test_trackers_on_device.py: `pytracking.tracker.{dimp,tomp,atom}` as shipped, `params.use_gpu = True`, on the
    MI355X through `pytracking_amd.install()`, against the CPU logs tests/golden/tracker_*.npz;
  * bench.py `cpu_baseline` (kind "reference") and `gpu_stock_baseline`: the reference's own `DiMPSteepestDescentGN` /
    `filter_layer.apply_filter` modules timed on the GPU box's host cores / on the GPU in stock PyTorch-ROCm.
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("PYTRACKING_REFERENCE_SRC", "/root/reference")
DST = os.path.join(HERE, "_ref", "reference")

PACKAGES = ("ltr", "pytracking")
# directories (relative to the reference root) the path never imports: kept out to keep the snapshot small
SKIP_DIRS = ("ltr/dataset", "ltr/data_specs", "ltr/train_settings", "ltr/actors", "ltr/trainers",
             "pytracking/notebooks", "pytracking/analysis", "pytracking/VOT", "pytracking/util_scripts",
             "pytracking/experiments", "ltr/external")


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


def _wanted(rel):
    rel = rel.replace(os.sep, "/")
    if not rel.endswith(".py"):
        return False
    return not any(rel == d or rel.startswith(d + "/") for d in SKIP_DIRS)


def build(src=SRC, dst=DST):
    if not os.path.isdir(os.path.join(src, "ltr")):
        raise RuntimeError(f"reference tree not found at {src}")
    if os.path.isdir(dst):
        shutil.rmtree(dst)
    manifest = {}
    for pkg in PACKAGES:
        for root, dirs, files in os.walk(os.path.join(src, pkg)):
            dirs[:] = [d for d in dirs if d != "__pycache__"]
            for f in sorted(files):
                full = os.path.join(root, f)
                rel = os.path.relpath(full, src)
                if not _wanted(rel):
                    continue
                out = os.path.join(dst, rel)
                os.makedirs(os.path.dirname(out), exist_ok=True)
                shutil.copyfile(full, out)
                manifest[rel.replace(os.sep, "/")] = _sha(out)
    with open(os.path.join(dst, "MANIFEST.json"), "w") as fh:
        json.dump({"source": "visionml/pytracking, read-only snapshot mounted at /root/reference in the build container",
                   "files": manifest}, fh, indent=0, sort_keys=True)
    return dst, len(manifest)


def available(dst=DST):
    return os.path.isfile(os.path.join(dst, "MANIFEST.json")) and os.path.isdir(os.path.join(dst, "ltr"))


def verify(dst=DST, src=None):
    """Every bundled file hashes to its manifest entry (and, where the reference tree is mounted, to the file it was copied
    from).  Returns the number of files checked."""
    with open(os.path.join(dst, "MANIFEST.json")) as fh:
        files = json.load(fh)["files"]
    for rel, digest in files.items():
        if _sha(os.path.join(dst, rel)) != digest:
            raise AssertionError(f"bundle file differs from its manifest entry: {rel}")
        if src is not None and _sha(os.path.join(src, rel)) != digest:
            raise AssertionError(f"bundle file differs from the reference: {rel}")
    return len(files)


if __name__ == "__main__":
    d, n = build()
    size = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(d) for f in fs)
    print(f"{n} reference files -> {d} ({size / 1e6:.2f} MB); verified {verify(d, SRC)} against {SRC}")
    sys.exit(0)
