"""ctypes binding of include/uvltrack_hip.h (the C ABI of the gfx950 library).

There is deliberately NO fallback: if the shared library is missing or a HIP device is absent the
product path raises -- it never routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libuvltrack_hip.so")

UVL_MAX_LAYERS = 64
UVL_NFAM = 5
UVL_ABI_VERSION = 4      # uvl_version(): bumped whenever a struct of include/uvltrack_hip.h changes size (3: uvl_tuning.res_pre / fin_w, the LayerNorm-free entry points)

# every symbol include/uvltrack_hip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTS = [
    "uvl_last_error", "uvl_version", "uvl_build_toolchain", "uvl_create", "uvl_destroy", "uvl_load_tensor", "uvl_finalize_weights",
    "uvl_workspace_bytes", "uvl_forward_test", "uvl_forward_prompt", "uvl_forward", "uvl_anno2mask", "uvl_decode", "uvl_crop_geometry_of", "uvl_sample_target", "uvl_sample_target_window", "uvl_sample_target_staged", "uvl_grounding_resize", "uvl_normalize_u8", "uvl_graph_capture", "uvl_graph_launch", "uvl_graph_release",
    "uvl_forward_test_profiled", "uvl_profile_count", "uvl_profile_entry", "uvl_profile_entry_weight_bytes", "uvl_debug_set", "uvl_tune_set", "uvl_tuning_init", "uvl_linear_splitk",
    "uvl_linear", "uvl_linear_pk", "uvl_pack_weight", "uvl_attention", "uvl_qkv_project", "uvl_qkv_project_pk", "uvl_layernorm", "uvl_f32_to_bf16", "uvl_fold_conv_bn", "uvl_conv_tower_layer",
    "uvl_fold_ln_linear", "uvl_linear_fin", "uvl_linear_lnf", "uvl_qkv_project_lnf", "uvl_head_end", "uvl_contrast_logits",
]


class NativeLibraryError(RuntimeError):
    pass


class UvlConfig(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("heads", C.c_int32), ("depth", C.c_int32), ("n_fusion_start", C.c_int32),
        ("n_cont", C.c_int32), ("cont_layers", C.c_int32 * UVL_MAX_LAYERS),
        ("template_size", C.c_int32), ("search_size", C.c_int32), ("text_len", C.c_int32), ("head_dim", C.c_int32),
        ("vocab", C.c_int32), ("max_pos", C.c_int32), ("txt_token_mean", C.c_int32), ("cls_tokenize", C.c_int32),
        ("offset_sigmoid", C.c_int32), ("joint_cls", C.c_int32), ("softmax_one", C.c_int32), ("max_batch", C.c_int32),
    ]


class UvlInputs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32),
        ("d_template", C.c_void_p), ("d_search", C.c_void_p), ("d_text_ids", C.c_void_p), ("d_text_mask", C.c_void_p),
        ("d_prompt", C.c_void_p), ("d_flag", C.c_void_p), ("skip_text", C.c_int32), ("reuse_text", C.c_int32),
    ]


class UvlOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "d_search", "d_template", "d_text", "d_vis_token", "d_txt_token", "d_logits", "d_cls_score", "d_cls_score_test",
        "d_bbox_map", "d_pred_boxes", "d_cont_score", "d_argmax")]


class UvlCropGeometry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("crop_sz", "x1", "y1", "x1_pad", "x2_pad", "y1_pad", "y2_pad")] + [("resize_factor", C.c_float)]


TUNING_FIELDS = ("gemm_cfg", "gemm_gm", "gemm_prod", "gemm_big", "gemm_kxcd", "attn_cfg", "sk_k1", "sk_k4", "gemm_pipe", "ring1", "text_cfg", "res_store", "slab_store", "attn_wgs", "gemm_dr", "res_pre", "fin_w", "lnf_w")


class UvlTuning(C.Structure):
    """include/uvltrack_hip.h: uvl_tuning -- overrides of the launch heuristics for tools and tests, every field -1 = heuristic.
    `UvlTuning(gemm_cfg=11)` forces one choice; pass `tuning.ref()` (or None) to the per-kernel entry points."""
    _fields_ = [(n, C.c_int32) for n in TUNING_FIELDS]

    def __init__(self, **kw):
        super().__init__()
        for n in TUNING_FIELDS:
            setattr(self, n, -1)
        for k, v in kw.items():
            if k not in TUNING_FIELDS:
                raise KeyError("unknown tuning field %r" % k)
            setattr(self, k, int(v))

    def ref(self):
        return C.byref(self)


_lib = None


def load():
    """Load the shared library (building is the job of uvltrack_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "HIP library %s is missing -- run `python -m uvltrack_amd.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the product path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64p, f32p = C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_float)
    lib.uvl_last_error.restype = C.c_char_p
    lib.uvl_version.restype = C.c_int
    lib.uvl_build_toolchain.restype = C.c_char_p
    # hand-counted waits / generated K loops are only known-good for one hipcc (uvltrack_amd/build.py): refuse a library built by another
    from . import build as _build
    stamp = (lib.uvl_build_toolchain() or b"").decode()
    if not _build.same_toolchain(stamp):
        msg = ("%s was built by '%s', not by the tested toolchain '%s' (hand-counted waits in the attention / GEMM kernels); rebuild with "
               "it, or set %s=1 and re-run the forced-kernel GPU tests" % (LIB_PATH, stamp, _build.TESTED_HIPCC, _build.OVERRIDE_ENV))
        if os.environ.get(_build.OVERRIDE_ENV) != "1":
            raise NativeLibraryError(msg)
        import warnings
        warnings.warn(msg)
    elif _build.only_hash_differs(stamp):
        import warnings
        warnings.warn("%s was built by '%s': the tested release (%s) with another build hash -- accepted; the hand-counted waits were validated on the tested hash"
                      % (LIB_PATH, stamp, _build.TESTED_HIPCC))
    if lib.uvl_version() != UVL_ABI_VERSION:
        raise NativeLibraryError("%s has ABI version %d, this binding is for %d (uvl_tuning / entry points changed): rebuild with `python -m uvltrack_amd.build --force`"
                                 % (LIB_PATH, lib.uvl_version(), UVL_ABI_VERSION))
    lib.uvl_create.restype = vp
    lib.uvl_create.argtypes = [C.POINTER(UvlConfig)]
    lib.uvl_destroy.argtypes = [vp]
    lib.uvl_destroy.restype = None
    lib.uvl_load_tensor.argtypes = [vp, C.c_char_p, vp, i32, i64p, vp]
    lib.uvl_finalize_weights.argtypes = [vp, vp]
    lib.uvl_workspace_bytes.argtypes = [vp, i32]
    lib.uvl_workspace_bytes.restype = C.c_size_t
    lib.uvl_forward_test.argtypes = [vp, C.POINTER(UvlInputs), C.POINTER(UvlOutputs), vp, C.c_size_t, vp]
    lib.uvl_forward_prompt.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.uvl_forward.argtypes = [vp, C.POINTER(UvlInputs), vp, vp, C.POINTER(UvlOutputs), vp, vp, C.c_size_t, vp]
    lib.uvl_decode.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp]
    lib.uvl_crop_geometry_of.argtypes = [C.POINTER(C.c_float), C.c_float, i32, i32, i32, C.POINTER(UvlCropGeometry)]
    lib.uvl_sample_target.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_float), C.c_float, i32, vp, vp, vp, C.POINTER(UvlCropGeometry), vp]
    lib.uvl_sample_target_window.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float), C.c_float, i32, vp, vp, vp,
                                             C.POINTER(UvlCropGeometry), vp]
    lib.uvl_sample_target_staged.argtypes = [vp, vp, C.c_size_t, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float), C.c_float, i32, vp, vp, vp,
                                             C.POINTER(UvlCropGeometry), vp]
    lib.uvl_grounding_resize.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, C.POINTER(C.c_int32), vp]
    lib.uvl_normalize_u8.argtypes = [vp, i32, i32, vp, vp]
    lib.uvl_graph_capture.argtypes = [vp, C.POINTER(UvlInputs), C.POINTER(UvlOutputs), vp, C.c_size_t]
    lib.uvl_graph_launch.argtypes = [vp, vp]
    lib.uvl_graph_release.argtypes = [vp]
    lib.uvl_forward_test_profiled.argtypes = [vp, C.POINTER(UvlInputs), C.POINTER(UvlOutputs), vp, C.c_size_t, vp, f32p, C.POINTER(C.c_int)]
    lib.uvl_profile_count.argtypes = [vp]
    lib.uvl_profile_entry.argtypes = [vp, i32, C.c_char_p, C.c_char_p, i32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.uvl_profile_entry_weight_bytes.argtypes = [vp, i32, C.POINTER(C.c_double)]
    lib.uvl_debug_set.argtypes = [vp, C.c_char_p, i32]
    tp = C.POINTER(UvlTuning)
    lib.uvl_tune_set.argtypes = [vp, C.c_char_p, i32]
    lib.uvl_tuning_init.argtypes = [tp]
    lib.uvl_tuning_init.restype = None
    lib.uvl_linear_splitk.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, tp, vp]
    lib.uvl_linear.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, tp, vp]
    lib.uvl_linear_pk.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, tp, vp]
    lib.uvl_pack_weight.argtypes = [vp, vp, i32, i32, vp]
    lib.uvl_attention.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, tp, vp]
    lib.uvl_qkv_project.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, C.c_float, tp, vp]
    lib.uvl_qkv_project_pk.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, C.c_float, tp, vp]
    lib.uvl_layernorm.argtypes = [vp, vp, vp, C.c_float, vp, vp, i32, i32, vp]
    lib.uvl_anno2mask.argtypes = [vp, i32, i32, vp, vp]
    lib.uvl_fold_conv_bn.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.uvl_conv_tower_layer.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_int32), i32, i32, vp, vp, vp, vp, tp, vp]
    lib.uvl_f32_to_bf16.argtypes = [vp, vp, C.c_size_t, vp]
    lib.uvl_fold_ln_linear.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.uvl_linear_fin.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, C.c_float, vp, tp, vp]
    lib.uvl_linear_lnf.argtypes = [vp, vp, vp, vp, vp, C.c_float, vp, i32, i32, i32, i32, tp, vp]
    lib.uvl_qkv_project_lnf.argtypes = [vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, i32, i32, i32, i32, C.c_float, tp, vp]
    lib.uvl_contrast_logits.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, i32, i32, i32, vp]
    lib.uvl_head_end.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    _lib = lib
    return lib


def check(rc: int, what: str = "uvl call"):
    if rc < 0:
        raise NativeLibraryError("%s failed (%d): %s" % (what, rc, load().uvl_last_error().decode()))
    return rc


def config_from_spec(spec, max_batch: int = 64) -> UvlConfig:
    c = UvlConfig()
    c.dim, c.heads, c.depth = spec.dim, spec.heads, spec.depth
    c.n_fusion_start = spec.n_bert
    c.n_cont = len(spec.cont_layers)
    for i, l in enumerate(spec.cont_layers):
        c.cont_layers[i] = l
    c.template_size, c.search_size = spec.template_size, spec.search_size
    c.text_len, c.head_dim = spec.text_len, spec.head_dim
    c.vocab, c.max_pos = spec.vocab, spec.max_pos
    c.txt_token_mean = 1 if spec.txt_token_mode == "mean" else 0
    c.cls_tokenize, c.offset_sigmoid = int(spec.cls_tokenize), int(spec.offset_sigmoid)
    c.joint_cls, c.softmax_one = int(spec.joint_cls), int(spec.softmax_one)
    c.max_batch = max_batch
    return c
