"""The reference bundle that travels to the GPU box (oracle/_ref/reference) is the reference, byte for byte."""
import os

import pytest

from oracle import make_ref_bundle as MB


@pytest.mark.skipif(not MB.available(), reason="bundle not built (python -B oracle/make_ref_bundle.py)")
def test_bundle_matches_manifest_and_reference():
    src = MB.SRC if os.path.isdir(os.path.join(MB.SRC, "ltr")) else None
    assert MB.verify(MB.DST, src) > 150
    # the tracker classes and the hot-path modules are in it
    for rel in ("pytracking/tracker/dimp/dimp.py", "pytracking/tracker/atom/atom.py", "pytracking/tracker/tomp/tomp.py",
                "ltr/models/target_classifier/optimizer.py", "ltr/models/layers/filter.py", "pytracking/libs/optimization.py"):
        assert os.path.isfile(os.path.join(MB.DST, rel)), rel


def test_bundle_is_not_tracked_by_git():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, ".gitignore")) as fh:
        assert "oracle/_ref/" in fh.read().split()
    ign = os.path.join(root, ".gpurunignore")
    if os.path.exists(ign):
        with open(ign) as fh:
            assert "oracle/_ref" not in fh.read()
