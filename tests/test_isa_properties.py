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
    co-resident workgroups on most CUs (profiles/HISTORY.md section 7) -- that needs >= 5 waves per SIMD by registers."""
    corr, adj = _pick(passes, "k_corr2"), _pick(passes, "k_adj2")
    assert len(corr) >= 8 and len(adj) >= 8
    for k, v in {**corr, **adj}.items():
        assert v["scratch"] == 0, (k, v)
    for k, v in corr.items():                                    # NK = 8: the 18 x 18 maps of the headline; larger maps hold more tiles
        assert v["occupancy"] >= (5 if "k_corr2ILi8E" in k else 4), (k, v)
    for k, v in adj.items():
        assert v["occupancy"] >= 2, (k, v)


# ------------------------------------------------------------------------------------------------------
# late-fetched kernel-argument blocks (common.h: pt_late_issue / pt_late_args) stay inside the kernel-argument segment
# ------------------------------------------------------------------------------------------------------
LLVM = "/opt/rocm/lib/llvm/bin"
LATE_SOURCES = ("fast_passes", "sd_solver", "atom_cg", "iou_refine", "tomp")


def _kernel_metadata(src, tmp_path_factory):
    """{mangled name: (kernarg_segment_size, [(offset, size, value_kind), ...])} from the code object's AMDGPU notes."""
    import yaml
    for tool in (HIPCC, os.path.join(LLVM, "clang-offload-bundler"), os.path.join(LLVM, "llvm-readelf")):
        if not os.path.exists(tool):
            pytest.skip(tool + " not available")
    sys.path.insert(0, ROOT)
    from pytracking_amd import _lib
    d = tmp_path_factory.mktemp("meta")
    obj, co = str(d / (src + ".o")), str(d / (src + ".co"))
    r = subprocess.run([HIPCC, *[f for f in _lib.HIPCC_FLAGS if f != "-fPIC"], "--cuda-device-only", "-c", "-I", CSRC, "-I",
                        os.path.join(ROOT, "include"), os.path.join(CSRC, src + ".hip"), "-o", obj], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + obj,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    i = notes.index("amdhsa.kernels")
    doc = notes[notes.rindex("---", 0, i):]
    doc = doc[:doc.index("\n...")]
    out = {}
    for k in yaml.safe_load(doc)["amdhsa.kernels"]:
        out[k[".name"]] = (int(k[".kernarg_segment_size"]), [(int(a[".offset"]), int(a[".size"]), a[".value_kind"]) for a in k[".args"]])
    return out


def _late_fetch_sites(src):
    """[(kernel name, struct type, offset expression)] for every pt_late_issue / pt_late_args call, attributed to the enclosing
    __global__ function by source position."""
    text = open(os.path.join(CSRC, src + ".hip")).read()
    kernels = [(m.start(), m.group(1)) for m in re.finditer(r"__global__[^;{]*?\bvoid\s+(\w+)\s*\(", text)]
    sites = []
    for m in re.finditer(r"pt_late_(?:issue|args)<(\w+)>\(([^;]*)\);", text):
        if text.rfind("\n", 0, m.start()) >= 0 and text[text.rfind("\n", 0, m.start()):m.start()].lstrip().startswith("//"):
            continue
        owner = [name for pos, name in kernels if pos < m.start()]
        assert owner, m.group(0)
        sites.append((owner[-1], m.group(1), m.group(2).strip()))
    return sites


@pytest.mark.parametrize("src", LATE_SOURCES)
def test_late_argument_blocks_lie_inside_the_kernarg_segment(src, tmp_path_factory):
    """The byte offsets handed to pt_late_issue / pt_late_args are written by hand (the scalars in front of the struct, 8-aligned).
    Against the code object's own metadata: every such offset is the `.offset` of a by-value struct parameter of that kernel, and
    offset + size of that parameter ends inside `.kernarg_segment_size` (the fetch reads exactly sizeof(T) bytes -- whole 16-dword
    blocks plus a narrow tail, common.h -- so nothing is read behind the segment; round 3 over-read up to 56 bytes)."""
    meta = _kernel_metadata(src, tmp_path_factory)
    sites = _late_fetch_sites(src)
    assert sites, src
    lv_off = None
    m = re.search(r"constexpr unsigned LV_OFF = (\d+);", open(os.path.join(CSRC, src + ".hip")).read())
    if m:
        lv_off = int(m.group(1))
    checked = 0
    for kernel, typ, expr in sites:
        inst = {k: v for k, v in meta.items() if re.search(r"(?:_GLOBAL__N_1|(?<!\d))%d%s" % (len(kernel), kernel), k)}   # Itanium mangling: <length><name>
        assert inst, (kernel, "no compiled instantiation")
        for name, (seg, args) in inst.items():
            structs = [(o, s) for o, s, kind in args if kind == "by_value" and s > 8]
            if expr.isdigit():
                want = [int(expr)]
            else:                                                  # the IoU kernels index one of two equally sized level blocks
                assert lv_off is not None and "LV_OFF" in expr, expr
                lv = [s for o, s in structs if o == lv_off]
                assert lv, (name, args)
                want = [lv_off, lv_off + lv[0]] if "?" in expr else [lv_off + 2 * lv[0]]
            for off in want:
                hit = [s for o, s in structs if o == off]
                assert hit, (name, typ, off, args)
                assert hit[0] % 4 == 0 and off + hit[0] <= seg, (name, typ, off, hit[0], seg)
                checked += 1
    assert checked >= len(sites)
