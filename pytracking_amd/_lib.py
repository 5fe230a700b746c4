"""Loader + ctypes prototypes for libpt_hot.so (C ABI declared in include/pt_hot.h).

The product path has no fallback: if the shared library is missing or a call fails, a RuntimeError is
raised.  `build_library()` compiles it in-tree with hipcc for gfx950 (cross-compiles without a GPU).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_HOT_LIB", os.path.join(_HERE, "libpt_hot.so"))   # override: experiments only
SOURCES = ["filter_kernels.hip", "fast_passes.hip", "sd_solver.hip", "mf_kernels.hip", "lwl_solver.hip", "atom_cg.hip", "atom_gn.hip", "tomp.hip", "localize.hip", "iou_refine.hip", "prroi.hip", "patch.hip", "frame_full.hip", "api.hip"]
HEADERS = ["common.h", "pt_internal.h", "rbuild.h", "sd_common.h", "mfma_gemm.h", "prroi_dev.h", "localize_dev.h", "frame_mid.h", os.path.join("..", "..", "include", "pt_hot.h")]
# kernarg preload: leading scalar kernel parameters arrive in SGPRs at wave launch (kernels that take them that way only)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-mllvm", "-amdgpu-kernarg-preload-count=14"]

PT_SD_DIMP, PT_SD_DIMP_L2, PT_SD_PRDIMP = 0, 1, 2
PT_ERR_NULL, PT_ERR_SHAPE, PT_ERR_UNSUPPORTED, PT_ERR_WORKSPACE, PT_ERR_LAUNCH = -1, -2, -3, -4, -5   # include/pt_hot.h
PT_ACT_RELU, PT_ACT_BENTPAR = 0, 1
PT_MASK_SIGMOID, PT_MASK_LINEAR = 0, 1

class IouDims(ctypes.Structure):
    """`pt_iou_dims` of include/pt_hot.h."""
    _fields_ = [(n, ctypes.c_int) for n in ("C3", "C4", "I3", "I4", "H3", "W3", "H4", "W4")]


class PatchGeom(ctypes.Structure):
    """`pt_patch_geom` of include/pt_hot.h."""
    _fields_ = [(n, ctypes.c_int) for n in ("df", "os0", "os1", "tl0", "tl1", "crop_h", "crop_w")]


class AugDesc(ctypes.Structure):
    """`pt_aug_desc` of include/pt_hot.h."""
    _fields_ = ([(n, ctypes.c_int) for n in ("kind", "pad_top", "pad_left", "th", "tw", "fs0", "fs1", "tap_off0", "tap_off1",
                                             "reserved")] + [("m", ctypes.c_double * 6)])


PT_AUG_IDENTITY, PT_AUG_FLIP_H, PT_AUG_FLIP_V, PT_AUG_BLUR, PT_AUG_SCALE, PT_AUG_ROTATE = range(6)
PT_AUG_MAX_TAPS = 256


class LocalizeParams(ctypes.Structure):
    """`pt_localize_params` of include/pt_hot.h."""
    _fields_ = ([(n, ctypes.c_double) for n in ("target_not_found_threshold", "uncertain_threshold", "hard_sample_threshold")]
                + [(n, ctypes.c_float) for n in ("distractor_threshold", "hard_negative_threshold", "target_not_found_f32",
                                                 "disp_threshold", "center_r", "center_c", "ratio_r", "ratio_c")]
                + [(n, ctypes.c_float * 8) for n in ("scale", "neigh_r", "neigh_c", "prev_r", "prev_c")])


class LocalizeState(ctypes.Structure):
    """`pt_localize_state` of include/pt_hot.h."""
    _fields_ = ([(n, ctypes.c_double) for n in ("target_not_found_threshold", "uncertain_threshold", "hard_sample_threshold",
                                                "distractor_threshold", "hard_negative_threshold",
                                                "target_neighborhood_scale", "dispalcement_scale")]
                + [(n, ctypes.c_float * 2) for n in ("kernel_size", "img_support_sz", "target_sz", "pos")]
                + [("sample_scales", ctypes.c_float * 8), ("sample_pos", ctypes.c_float * 16)])


PT_LOC_FLAGS = ("normal", "hard_negative", "uncertain", "not_found")      # PT_LOC_* of include/pt_hot.h


class FrameGlue(ctypes.Structure):
    """`pt_frame_glue` of include/pt_hot.h."""
    _fields_ = [("image_sz", ctypes.c_float * 2), ("img_sample_sz", ctypes.c_float * 2), ("target_inside_ratio", ctypes.c_double),
                ("box_jitter_pos", ctypes.c_double), ("box_jitter_sz", ctypes.c_double), ("use_classifier", ctypes.c_int),
                ("num_random", ctypes.c_int), ("rand_u", ctypes.c_float * 60)]


PT_FRAME_HOST_FLOATS = 128


class TompDims(ctypes.Structure):
    """`pt_tomp_dims` of include/pt_hot.h."""
    _fields_ = [(n, ctypes.c_int) for n in ("d_model", "nhead", "dim_ff", "n_enc", "n_dec", "H", "W", "max_res")]


EXPORTS = [
    "pt_strerror", "pt_abi_version",
    "pt_apply_filter_ws_bytes", "pt_apply_filter_f32",
    "pt_feat_transpose_ws_bytes", "pt_feat_transpose_f32",
    "pt_sd_ws_bytes", "pt_sd_solve_f32",
    "pt_atom_cg_ws_bytes", "pt_atom_cg_f32", "pt_atom_gn_ws_bytes", "pt_atom_gn_f32",
    "pt_prroi_fwd_f32", "pt_prroi_bwd_feat_f32", "pt_prroi_bwd_coor_f32",
    "pt_track_frame_ws_bytes", "pt_track_frame_f32", "pt_track_frame_chain_f32", "pt_track_frame_flush_f32",
    "pt_apply_filter_mf_ws_bytes", "pt_apply_filter_mf_f32", "pt_feat_transpose_mf_ws_bytes", "pt_feat_transpose_mf_f32",
    "pt_lwl_ws_bytes", "pt_lwl_gn_solve_f32",
    "pt_tomp_param_floats", "pt_tomp_prepared_floats", "pt_tomp_prepare_f32", "pt_tomp_posenc_f32", "pt_tomp_predict_ws_bytes", "pt_tomp_predict_f32", "pt_tomp_linear_f32",
    "pt_tomp_bbreg_param_floats", "pt_tomp_bbreg_ws_bytes", "pt_tomp_bbreg_f32",
    "pt_clf_head_ws_bytes", "pt_clf_head_f32", "pt_max2d_f32", "pt_localize_f32", "pt_localize_decide_f32",
    "pt_localize_constants_f32", "pt_localize_advanced_f32", "pt_localize_advanced_sync_f32",
    "pt_iou_param_floats", "pt_iou_prepared_floats", "pt_iou_prepare_f32", "pt_iou_refine_ws_bytes", "pt_iou_refine_f32", "pt_iou_refine_sync_f32",
    "pt_track_frame_replay_pass_f32", "pt_sample_patch_f32", "pt_augment_patches_f32", "pt_track_frame_head_ws_bytes", "pt_track_frame_head_f32",
    "pt_track_frame_full_ws_bytes", "pt_track_frame_full_f32", "pt_track_frame_full_launch_f32", "pt_host_buffer_forget",
    "pt_track_frame_full_dyn_bytes", "pt_track_frame_full_dyn_fill_f32", "pt_host_wait_word_f32",
    "pt_sd_solve_batch_f32", "pt_stream_probe_f32",
]


class FramePending(ctypes.Structure):
    """`pt_frame_pending` of include/pt_hot.h."""
    _fields_ = [("iters", ctypes.c_int), ("step_length", ctypes.c_float), ("reg_eps", ctypes.c_float)]


class SdParams(ctypes.Structure):
    """Mirror of `pt_sd_params` (include/pt_hot.h)."""
    _fields_ = [
        ("kind", ctypes.c_int), ("step_length", ctypes.c_float), ("reg", ctypes.c_float),
        ("alpha_eps", ctypes.c_float), ("feat_stride", ctypes.c_float),
        ("num_bins", ctypes.c_int), ("bin_displacement", ctypes.c_float),
        ("label_lut", ctypes.c_void_p), ("mask_lut", ctypes.c_void_p), ("spatial_lut", ctypes.c_void_p),
        ("mask_act", ctypes.c_int), ("score_act", ctypes.c_int), ("act_param", ctypes.c_float),
        ("gauss_sigma", ctypes.c_float), ("hinge_threshold", ctypes.c_float),
        ("uni_weight", ctypes.c_float), ("normalize_label", ctypes.c_int), ("label_shrink", ctypes.c_float),
        ("has_softmax_reg", ctypes.c_int), ("softmax_reg", ctypes.c_float), ("label_threshold", ctypes.c_float),
    ]


class FrameFull(ctypes.Structure):
    """`pt_frame_full` of include/pt_hot.h."""
    _fields_ = ([("sd", ctypes.POINTER(SdParams))]
                + [(n, ctypes.c_void_p) for n in ("filter", "mem_feat", "mem_bb", "sample_weight", "backbone_feat",
                                                  "head_weight_tap_major")]
                + [("norm_scale", ctypes.c_float), ("norm_eps", ctypes.c_float)]
                + [(n, ctypes.c_int) for n in ("slot", "n", "Cin", "C", "H", "W", "K", "num_iter")]
                + [("scores_out", ctypes.c_void_p), ("peak_out", ctypes.c_void_p),
                   ("loc", ctypes.POINTER(LocalizeState)), ("glue", ctypes.POINTER(FrameGlue)), ("iou_dims", ctypes.POINTER(IouDims))]
                + [(n, ctypes.c_void_p) for n in ("iou_params", "iou_prepared", "c3", "c4", "mod3", "mod4")]
                + [("iou_iter", ctypes.c_int), ("relative", ctypes.c_int), ("step_length4", ctypes.c_float * 4),
                   ("step_decay", ctypes.c_float), ("aux_stream", ctypes.c_void_p), ("aux_reordered_update_ok", ctypes.c_int),
                   ("dyn", ctypes.c_void_p)])


def _build_flags():
    return HIPCC_FLAGS + os.environ.get("PT_HOT_CFLAGS", "").split()


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    if "PT_HOT_LIB" in os.environ:                           # an experiment's prebuilt variant is never rebuilt from the current sources
        return False
    stamp = os.path.join(_HERE, "build", "flags.txt")       # a changed PT_HOT_CFLAGS is a rebuild, not a silently stale library
    if os.path.exists(stamp) and open(stamp).read() != " ".join(_build_flags()):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(_HERE, "csrc", s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=False) -> str:
    """hipcc --offload-arch=gfx950 -> pytracking_amd/libpt_hot.so (in-tree, travels with the repo snapshot).
    One object per source under pytracking_amd/build/ (compiled in parallel, only the stale ones unless `force`), then
    one link step."""
    if not force and not needs_build():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = _build_flags()
    bdir = os.path.join(_HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    stamp = os.path.join(bdir, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    hdr_t = max(os.path.getmtime(os.path.join(_HERE, "csrc", h)) for h in HEADERS if os.path.exists(os.path.join(_HERE, "csrc", h)))

    def compile_one(src):
        spath = os.path.join(_HERE, "csrc", src)
        obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), hdr_t):
            return obj, None
        cmd = [hipcc] + flags + ["-c", spath, "-o", obj]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or res.returncode != 0:
            print(" ".join(cmd))
            print(res.stdout, res.stderr)
        return obj, (res.stderr if res.returncode != 0 else None)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    errs = [e for _, e in results if e]
    if errs:
        raise RuntimeError("hipcc failed building libpt_hot.so:\n" + "\n".join(errs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + [o for o, _ in results]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(cmd))
        print(res.stdout, res.stderr)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed linking libpt_hot.so:\n" + res.stderr)
    with open(stamp, "w") as fh:
        fh.write(" ".join(flags))
    return LIB_PATH


_lib = None


def lib():
    """The loaded library (ctypes.CDLL) with argtypes set.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback on the product path)")
    L = ctypes.CDLL(LIB_PATH)
    vp, f, i, l, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_long, ctypes.c_size_t
    L.pt_strerror.restype = ctypes.c_char_p
    L.pt_strerror.argtypes = [i]
    L.pt_abi_version.restype = i
    L.pt_apply_filter_ws_bytes.restype = sz
    L.pt_apply_filter_ws_bytes.argtypes = [i] * 8
    L.pt_apply_filter_f32.restype = i
    L.pt_apply_filter_f32.argtypes = [vp, l, vp, vp] + [i] * 8 + [vp, sz, vp]
    L.pt_feat_transpose_ws_bytes.restype = sz
    L.pt_feat_transpose_ws_bytes.argtypes = [i] * 8
    L.pt_feat_transpose_f32.restype = i
    L.pt_feat_transpose_f32.argtypes = [vp, l, vp, vp] + [i] * 8 + [vp, sz, vp]
    L.pt_sd_ws_bytes.restype = sz
    L.pt_sd_ws_bytes.argtypes = [i] * 5
    L.pt_sd_solve_f32.restype = i
    L.pt_sd_solve_f32.argtypes = [ctypes.POINTER(SdParams), vp, vp, l, vp, vp] + [i] * 6 + [vp, vp, vp, sz, vp]
    vpp = ctypes.POINTER(ctypes.c_void_p)
    L.pt_sd_solve_batch_f32.restype = i
    L.pt_sd_solve_batch_f32.argtypes = [ctypes.POINTER(SdParams), i, vpp, vpp, l, vpp, vpp] + [i] * 6 + [vpp, vpp, vpp, sz, vp, vpp, i]
    L.pt_atom_cg_ws_bytes.restype = sz
    L.pt_atom_cg_ws_bytes.argtypes = [i] * 5
    L.pt_atom_cg_f32.restype = i
    L.pt_atom_cg_f32.argtypes = [vp, vp, l, vp, vp, f, f] + [i] * 7 + [f, vp, vp, sz, vp]
    L.pt_atom_gn_ws_bytes.restype = sz
    L.pt_atom_gn_ws_bytes.argtypes = [i] * 6
    L.pt_atom_gn_f32.restype = i
    L.pt_atom_gn_f32.argtypes = [vp, vp, vp, l, vp, vp, f, f, f] + [i] * 6 + [ctypes.POINTER(ctypes.c_int), i, i, vp, sz, vp]
    for name in ("pt_prroi_fwd_f32", "pt_prroi_bwd_feat_f32"):
        fn = getattr(L, name)
        fn.restype = i
        fn.argtypes = [vp, vp, vp] + [i] * 7 + [f, vp]
    L.pt_prroi_bwd_coor_f32.restype = i
    L.pt_prroi_bwd_coor_f32.argtypes = [vp, vp, vp, vp] + [i] * 7 + [f, vp]
    L.pt_track_frame_ws_bytes.restype = sz
    L.pt_track_frame_ws_bytes.argtypes = [i] * 5
    L.pt_track_frame_f32.restype = i
    L.pt_track_frame_f32.argtypes = [ctypes.POINTER(SdParams), vp, vp, vp, vp, vp] + [i] * 7 + [vp, vp, vp, sz, vp]
    L.pt_track_frame_chain_f32.restype = i
    L.pt_track_frame_chain_f32.argtypes = ([ctypes.POINTER(SdParams), vp, vp, vp, vp, vp] + [i] * 7 +
                                           [vp, vp, vp, sz, ctypes.POINTER(FramePending), i, vp])
    L.pt_track_frame_flush_f32.restype = i
    L.pt_track_frame_flush_f32.argtypes = [ctypes.POINTER(FramePending), vp] + [i] * 5 + [vp, sz, vp]
    L.pt_apply_filter_mf_ws_bytes.restype = sz
    L.pt_apply_filter_mf_ws_bytes.argtypes = [i] * 6
    L.pt_apply_filter_mf_f32.restype = i
    L.pt_apply_filter_mf_f32.argtypes = [vp, l, vp, vp] + [i] * 6 + [vp, sz, vp]
    L.pt_feat_transpose_mf_ws_bytes.restype = sz
    L.pt_feat_transpose_mf_ws_bytes.argtypes = [i] * 6
    L.pt_feat_transpose_mf_f32.restype = i
    L.pt_feat_transpose_mf_f32.argtypes = [vp, l, vp, vp] + [i] * 6 + [vp, sz, vp]
    L.pt_lwl_ws_bytes.restype = sz
    L.pt_lwl_ws_bytes.argtypes = [i] * 6
    L.pt_lwl_gn_solve_f32.restype = i
    L.pt_lwl_gn_solve_f32.argtypes = [vp, vp, l, vp, vp, i, f, f] + [i] * 7 + [vp, vp, vp, sz, vp]
    dp = ctypes.POINTER(TompDims)
    L.pt_tomp_param_floats.restype = sz
    L.pt_tomp_param_floats.argtypes = [dp]
    L.pt_tomp_prepared_floats.restype = sz
    L.pt_tomp_prepared_floats.argtypes = [dp]
    L.pt_tomp_prepare_f32.restype = i
    L.pt_tomp_prepare_f32.argtypes = [dp, vp, vp, vp]
    L.pt_tomp_posenc_f32.restype = i
    L.pt_tomp_posenc_f32.argtypes = [vp, i, i, i, i, vp]
    L.pt_tomp_predict_ws_bytes.restype = sz
    L.pt_tomp_predict_ws_bytes.argtypes = [dp, i, i, i]
    L.pt_tomp_predict_f32.restype = i
    L.pt_tomp_predict_f32.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, vp, vp, vp, sz, vp]
    L.pt_tomp_linear_f32.restype = i
    L.pt_tomp_linear_f32.argtypes = [vp, vp, vp, vp, i, i, i, i, vp]
    L.pt_tomp_bbreg_param_floats.restype = sz
    L.pt_tomp_bbreg_param_floats.argtypes = [i]
    L.pt_tomp_bbreg_ws_bytes.restype = sz
    L.pt_tomp_bbreg_ws_bytes.argtypes = [i, i, i, i]
    L.pt_tomp_bbreg_f32.restype = i
    L.pt_tomp_bbreg_f32.argtypes = [vp, vp, vp, vp, i, i, i, i, vp, sz, vp]
    L.pt_clf_head_ws_bytes.restype = sz
    L.pt_clf_head_ws_bytes.argtypes = [i] * 5
    L.pt_clf_head_f32.restype = i
    L.pt_clf_head_f32.argtypes = [vp, vp, vp, i, i, i, i, i, f, f, vp, sz, vp]
    L.pt_max2d_f32.restype = i
    L.pt_max2d_f32.argtypes = [vp, vp, vp, i, i, i, vp]
    fp = ctypes.POINTER(ctypes.c_float)
    L.pt_localize_f32.restype = i
    L.pt_localize_f32.argtypes = [vp, vp, fp, fp, vp, i, i, i, vp]
    L.pt_localize_decide_f32.restype = i
    L.pt_localize_decide_f32.argtypes = [vp, vp, ctypes.POINTER(LocalizeParams), vp, i, i, i, vp]
    L.pt_localize_constants_f32.restype = i
    L.pt_localize_constants_f32.argtypes = [ctypes.POINTER(LocalizeState), i, i, i, ctypes.POINTER(LocalizeParams)]
    L.pt_localize_advanced_f32.restype = i
    L.pt_localize_advanced_f32.argtypes = [vp, vp, ctypes.POINTER(LocalizeState), vp, i, i, i, vp]
    L.pt_localize_advanced_sync_f32.restype = i
    L.pt_localize_advanced_sync_f32.argtypes = [vp, vp, ctypes.POINTER(LocalizeState), vp, i, i, i, vp]
    ip = ctypes.POINTER(IouDims)
    L.pt_iou_param_floats.restype = sz
    L.pt_iou_param_floats.argtypes = [ip]
    L.pt_iou_prepared_floats.restype = sz
    L.pt_iou_prepared_floats.argtypes = [ip]
    L.pt_iou_prepare_f32.restype = i
    L.pt_iou_prepare_f32.argtypes = [ip, vp, vp, vp]
    L.pt_iou_refine_ws_bytes.restype = sz
    L.pt_iou_refine_ws_bytes.argtypes = [ip, i]
    L.pt_iou_refine_f32.restype = i
    L.pt_iou_refine_f32.argtypes = [ip, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, fp, f, i, i, vp, sz, vp]
    L.pt_iou_refine_sync_f32.restype = i
    L.pt_iou_refine_sync_f32.argtypes = [ip, vp, vp, vp, vp, vp, vp, vp, vp, i, i, fp, f, i, i, vp, sz, vp]
    L.pt_track_frame_replay_pass_f32.restype = i
    L.pt_track_frame_replay_pass_f32.argtypes = [ctypes.POINTER(SdParams), vp, vp, vp, vp] + [i] * 6 + [vp, sz, i, i, vp]
    L.pt_track_frame_head_ws_bytes.restype = sz
    L.pt_track_frame_head_ws_bytes.argtypes = [i] * 6
    L.pt_track_frame_head_f32.restype = i
    L.pt_track_frame_head_f32.argtypes = [ctypes.POINTER(SdParams), vp, vp, vp, vp, vp, vp, f, f] + [i] * 8 + [vp, vp, vp, sz, vp]
    ffp = ctypes.POINTER(FrameFull)
    L.pt_track_frame_full_ws_bytes.restype = sz
    L.pt_track_frame_full_ws_bytes.argtypes = [ffp]
    L.pt_track_frame_full_f32.restype = i
    L.pt_track_frame_full_f32.argtypes = [ffp, vp, vp, sz, vp]
    L.pt_track_frame_full_launch_f32.restype = i
    L.pt_track_frame_full_launch_f32.argtypes = [ffp, vp, vp, sz, vp]
    L.pt_track_frame_full_dyn_bytes.restype = sz
    L.pt_track_frame_full_dyn_bytes.argtypes = []
    L.pt_track_frame_full_dyn_fill_f32.restype = i
    L.pt_track_frame_full_dyn_fill_f32.argtypes = [ffp, ctypes.c_float, vp, vp, sz, vp]
    L.pt_host_wait_word_f32.restype = i
    L.pt_host_wait_word_f32.argtypes = [vp, ctypes.c_float, vp]
    L.pt_stream_probe_f32.restype = i
    L.pt_stream_probe_f32.argtypes = [vp, sz, vp, i, vp]
    L.pt_host_buffer_forget.restype = None
    L.pt_host_buffer_forget.argtypes = [vp]
    L.pt_sample_patch_f32.restype = i
    L.pt_sample_patch_f32.argtypes = [vp, i, i, i, ctypes.POINTER(PatchGeom), i, vp, i, i, vp]
    L.pt_augment_patches_f32.restype = i
    L.pt_augment_patches_f32.argtypes = [vp, i, i, i, ctypes.POINTER(AugDesc), i, ctypes.POINTER(ctypes.c_float), i, vp, i, i, vp]
    _lib = L
    return L


def check(status: int, what: str):
    if status != 0:
        raise RuntimeError(f"{what} failed: {lib().pt_strerror(status).decode()} ({status})")

