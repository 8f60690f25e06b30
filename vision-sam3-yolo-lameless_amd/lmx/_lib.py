"""ctypes loader for liblmx.so (built in-tree by ``csrc/Makefile`` / ``__graft_entry__.build``).

Fails loudly: a missing library is an ImportError, a failed call raises ``LmxError`` with the C side's message.
Nothing here (or anywhere in the package) falls back to a CPU path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LMX_LIB: development builds only (csrc/Makefile `dbg`); the product loads the in-tree liblmx.so
LIB_PATH = os.environ.get("LMX_LIB") or os.path.join(_HERE, "liblmx.so")


class LmxError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("scale", C.c_void_p), ("res", C.c_void_p),
        ("C", C.c_void_p),
        ("lda", C.c_int64), ("ldc", C.c_int64), ("ldr", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("act", C.c_int32), ("out_dtype", C.c_int32), ("a_mode", C.c_int32),
        ("H", C.c_int32), ("W_", C.c_int32), ("Cin", C.c_int32), ("conv_stride", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("res_rows", C.c_int32), ("a_rep", C.c_int32), ("split_k", C.c_int32), ("split_stride", C.c_int64),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
        ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64), ("ldo", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("Tq", C.c_int32), ("Tk", C.c_int32), ("hd", C.c_int32),
        ("scale", C.c_float), ("mode", C.c_int32),
        ("Gh", C.c_int32), ("Gw", C.c_int32), ("ws", C.c_int32), ("q_stride", C.c_int32),
        ("pad_k", C.c_void_p), ("pad_v", C.c_void_p),
        ("rel", C.c_void_p), ("rel_S", C.c_int32),
    ]


# name -> (restype, argtypes); must list every symbol include/lmx.h declares (tests/test_abi.py checks both ways)
_VP, _I, _I64, _F, _D = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
SIGNATURES = {
    "lmx_version": (_I, []),
    "lmx_last_error": (C.c_char_p, []),
    "lmx_device_count": (_I, []),
    "lmx_k_gemm": (_I, [C.POINTER(GemmDesc), _VP]),
    "lmx_dbg_set_gemm2_variant": (None, [_I]),
    "lmx_k_pose_gather": (_I, [_VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _F, _F, _F, _F, _F, _VP, _VP]),
    "lmx_k_pack_bits": (_I, [_VP, _I64, _I, _VP, _VP]),
    "lmx_k_ln_mlp": (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _I, _F, _VP, _VP, _VP, _VP, _VP, _VP]),
    "lmx_k_hiera_attn8": (_I, [_VP, _VP, _I64, _VP, _VP, _F, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _F, _VP]),
    "lmx_k_hiera_attn4": (_I, [_VP, _VP, _I64, _VP, _VP, _I, _I, _I, _I, _I, _F, _VP]),
    "lmx_k_hiera_attn_pool": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _F, _VP]),
    "lmx_k_ln_mlp_img": (_I, [_VP, _I64, _VP, _VP, _I64, _I, _F, _VP, _VP, _VP]),
    "lmx_k_layernorm": (_I, [_VP, _I, _I64, _VP, _VP, _VP, _I, _I64, _I, _I, _F, _I, _VP]),
    "lmx_k_attention": (_I, [C.POINTER(AttnDesc), _VP]),
    "lmx_k_relpos_tables": (_I, [C.POINTER(AttnDesc), _VP, _VP, _I, _VP, _VP]),
    "lmx_k_rope": (_I, [_VP, _I64, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "lmx_k_pil_resize_h": (_I, [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _I, _I, _VP]),
    "lmx_k_pil_resize_v": (_I, [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _I, _VP]),
    "lmx_k_patchify_norm": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I64, _VP, _VP]),
    "lmx_k_assemble_tokens": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "lmx_k_token_mean": (_I, [_VP, _I, _VP, _I, _I, _I, _VP]),
    "lmx_nms_workspace_bytes": (_I64, [_I, _I]),
    "lmx_k_nms": (_I, [_VP, _I, _I, _I, _F, _D, _I, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "lmx_k_letterbox": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _I, _VP]),
    "lmx_k_stem_conv": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "lmx_k_split3": (_I, [_VP, _I64, _I, _VP, _I64, _VP, _I64, _I64, _I, _I, _I, _I64, _VP]),
    "lmx_k_maxpool5_x3": (_I, [_VP, _I64, _VP, _I64, _I, _I, _I, _I, _VP]),
    "lmx_k_stem_conv_x3": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "lmx_k_attention_f32": (_I, [_VP, _I64, _VP, _I64, _VP, _I64, _VP, _I64, _I, _I, _I, _I, _I, _F, _VP]),
    "lmx_k_hyper_mask_f32": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "lmx_k_maxpool5": (_I, [_VP, _I64, _VP, _I64, _I, _I, _I, _I, _VP]),
    "lmx_k_upsample2": (_I, [_VP, _I64, _VP, _I64, _I, _I, _I, _I, _VP]),
    "lmx_k_detect_decode": (_I, [_VP, _I64, _VP, _I, _I, _I, _I, _F, _I, _I, _VP]),
    "lmx_k_scale_boxes": (_I, [_VP, _I, _F, _F, _F, _F, _F, _VP]),
    "lmx_k_im2col_u8": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I64, _VP]),
    "lmx_k_maxpool2": (_I, [_VP, _I64, _VP, _I64, _I, _I, _I, _I, _I, _VP]),
    "lmx_k_cast_f32_f16": (_I, [_VP, _I64, _VP, _I64, _I64, _I, _VP]),
    "lmx_k_add_bcast": (_I, [_VP, _I, _I64, _VP, _I64, _I, _VP, _I, _I64, _I64, _I, _VP]),
    "lmx_h_mask_features": (_I, [_VP, _I, _I, _VP]),
    "lmx_h_iou_matrix": (_I, [_VP, _I, _VP, _I, _VP]),
    "lmx_h_assign": (_I, [_VP, _I, _I, _VP, _VP]),
    "lmx_contour_workspace_bytes": (_I64, [_I, _I, _I]),
    "lmx_k_contour_features": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP]),
    "lmx_k_prompt_box": (_I, [_VP, _I64, _VP, _I, _D, _D, _F, _VP, _VP, _I, _VP]),
    "lmx_k_hyper_mask": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP]),
    "lmx_k_mask_post": (_I, [_VP, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP, _VP]),
}

_lib = None


def load():
    """Load liblmx.so and bind every declared symbol.  Raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  lmx has no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=os.RTLD_NOW)  # resolve every symbol now: a broken build fails here, not mid-run
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().lmx_last_error().decode("utf-8", "replace")
        raise LmxError(f"{what} failed (rc={rc}): {msg}")
