"""Properties of the compiled gfx950 code that the measured performance depends on, checked on the CPU box (hipcc cross-compiles; no GPU).
Round 3's lesson was that the ISA, not the source, decides where a wave waits: a run-time loop bound, a local array indexed with a
run-time value or one too many live values silently turn into scratch (private memory in HBM) or into a lower occupancy than the
launch geometry assumes.  These are the cheap invariants; the timings themselves live in profiles/."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytracking_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _resources(src, tmp_path_factory):
    """{mangled kernel name: {"vgpr", "scratch", "occupancy", "lds"}} from -Rpass-analysis=kernel-resource-usage."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    sys.path.insert(0, ROOT)
    from pytracking_amd import _lib
    out = tmp_path_factory.mktemp("isa") / (src + ".o")
    cmd = [HIPCC, *[f for f in _lib.HIPCC_FLAGS if f != "-fPIC"], "--cuda-device-only", "-c",
           "-I", CSRC, "-I", os.path.join(ROOT, "include"), os.path.join(CSRC, src + ".hip"), "-o", str(out),
           "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    res, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    assert res, "no kernel-resource remarks in the compiler output"
    return res


@pytest.fixture(scope="module")
def iou(tmp_path_factory):
    return _resources("iou_refine", tmp_path_factory)


@pytest.fixture(scope="module")
def passes(tmp_path_factory):
    return _resources("fast_passes", tmp_path_factory)


def _pick(res, needle):
    hit = {k: v for k, v in res.items() if needle in k}
    assert hit, needle
    return hit


def test_fused_iou_kernels_keep_their_state_in_registers(iou):
    """k_iou_fwd / k_iou_bwd hold two 6 x 6 windows' worth of operands, the weight fragments and the geometry reads: no scratch, and the
    LDS budget (geometry table + staged planes + partial products) has to fit one CU (160 KB)."""
    for name in ("9k_iou_fwd", "9k_iou_bwd", "11k_iou_head2", "11k_iou_final"):
        for k, v in _pick(iou, name).items():
            assert v["scratch"] == 0, (k, v)
            assert v["lds"] <= 160 * 1024, (k, v)
            assert v["vgpr"] <= 256, (k, v)


def test_pass_kernels_have_no_scratch_and_admit_two_workgroups_per_cu(passes):
    """Every instantiation of the two solver passes: no scratch; the correlation runs 400 workgroups of <= 10 waves on 256 CUs, i.e. two
    co-resident workgroups on most CUs -- that needs >= 5 waves per SIMD by registers."""
    corr, adj = _pick(passes, "k_corr2"), _pick(passes, "k_adj2")
    assert len(corr) >= 8 and len(adj) >= 8
    for k, v in {**corr, **adj}.items():
        assert v["scratch"] == 0, (k, v)
    for k, v in corr.items():                                    # NK = 8: the 18 x 18 maps of the headline; larger maps hold more tiles
        assert v["occupancy"] >= (5 if "k_corr2ILi8E" in k else 4), (k, v)
    for k, v in adj.items():
        assert v["occupancy"] >= 2, (k, v)
