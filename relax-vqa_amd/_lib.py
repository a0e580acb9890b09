"""ctypes binding of librelax_hip.so (include/relax_hip.h).  No fallback: if the
HIP library is missing or fails to load, importing the engine raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RELAX_HIP_LIB") or os.path.join(_HERE, "csrc", "librelax_hip.so")   # override: another build of it

c_u8p = C.POINTER(C.c_uint8)
c_i32p = C.POINTER(C.c_int32)
c_u32p = C.POINTER(C.c_uint32)
c_f32p = C.POINTER(C.c_float)
c_vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/relax_hip.h one to one
PROTOTYPES = {
    "relax_abi_version": (C.c_int, []),
    "relax_create": (C.c_int, [C.c_int, C.POINTER(c_vp)]),
    "relax_destroy": (C.c_int, [c_vp]),
    "relax_last_error": (C.c_char_p, [c_vp]),
    "relax_reserve": (C.c_int, [c_vp, C.c_int]),
    "relax_set_option": (C.c_int, [c_vp, C.c_char_p, C.c_int]),
    "relax_get_option": (C.c_int, [c_vp, C.c_char_p, C.POINTER(C.c_int)]),
    "relax_load_resnet50": (C.c_int, [c_vp, C.POINTER(c_vp), C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int]),
    "relax_load_vit": (C.c_int, [c_vp, C.POINTER(c_vp), C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int,
                                 C.c_int, C.c_int, C.c_int]),
    "relax_fragment_pairs": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                       c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "relax_fragment_image": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                       c_vp, c_vp, c_vp, c_vp, c_vp]),
    "relax_gather_patches": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int, C.c_int, C.c_int, c_vp, c_vp, c_vp, c_vp]),
    "relax_merge_fragments": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int64, c_vp]),
    "relax_optical_flow": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int, C.c_int, C.c_int, c_vp, c_vp, c_vp]),
    "relax_flow_to_rgb": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_vp, c_vp]),
    "relax_resize_frames": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int, C.c_int, C.c_int, c_vp, c_vp, c_vp]),
    "relax_resnet50_features": (C.c_int, [c_vp, c_vp, C.c_int, c_vp, c_vp, C.POINTER(c_vp), c_vp]),
    "relax_resnet50_clip_features": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, c_vp, c_vp, c_vp]),
    "relax_vit_features": (C.c_int, [c_vp, c_vp, C.c_int, c_vp, c_vp, c_vp]),
    "relax_load_mlp_head": (C.c_int, [c_vp, C.POINTER(c_vp), C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int,
                                      c_vp, c_vp, c_vp, C.c_int]),
    "relax_mlp_head": (C.c_int, [c_vp, c_vp, C.c_int, c_vp, c_vp]),
    "relax_op_gemm": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "relax_op_conv2d_nhwc": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp] + [C.c_int] * 10 + [c_vp]),
    "relax_op_layernorm": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_float, c_vp]),
    "relax_op_attention": (C.c_int, [c_vp, c_vp, c_vp, C.c_int, C.c_int, c_vp]),
    "relax_op_bn_relu_maxpool": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "relax_op_gap": (C.c_int, [c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int64, c_vp]),
    "relax_op_token_stats": (C.c_int, [c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_vp]),
    "relax_copy_bytes": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, c_vp]),
    "relax_segment_mean": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int, C.c_int, c_vp, C.c_int, c_vp, C.c_int64, C.c_int, c_vp]),
    "relax_profile_enable": (C.c_int, [c_vp, C.c_int]),
    "relax_profile_read": (C.c_int, [c_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_int64)]),
}

_lib = None


def load():
    """dlopen librelax_hip.so and attach prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C relax-vqa_amd/csrc).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.relax_abi_version() != 1:
        raise RuntimeError(f"librelax_hip.so ABI {lib.relax_abi_version()} != 1")
    _lib = lib
    return lib
