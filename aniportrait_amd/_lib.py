"""ctypes binding of libaniportrait_hip.so (the C ABI declared in include/aniportrait_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, this module
raises.  `load()` only dlopens and checks the exported symbols (works without a GPU);
compute entry points need a gfx950 device.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ANIP_LIB: an experiment build of the same ABI (aniportrait_amd/build.py --out=...); the product is the default path
LIB_PATH = os.environ.get("ANIP_LIB") or os.path.join(HERE, "lib", "libaniportrait_hip.so")
ABI_VERSION = 15

c_void_p, c_int, c_int64, c_float = C.c_void_p, C.c_int, C.c_int64, C.c_float


class GemmParams(C.Structure):
    """mirror of `struct anip_gemm_params`"""
    _fields_ = [
        ("A", c_void_p), ("lda", c_int64),
        ("A2", c_void_p), ("lda2", c_int64), ("K1", c_int),
        ("W", c_void_p), ("ldw", c_int64),
        ("out", c_void_p), ("ldo", c_int64),
        ("out_f32", c_int),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("alpha", c_float),
        ("bias", c_void_p),
        ("rowbias", c_void_p), ("rows_per_group", c_int64), ("ld_rowbias", c_int64),
        ("residual", c_void_p), ("ldr", c_int64),
        ("act", c_int),
        ("batch", c_int), ("strideA", c_int64), ("strideW", c_int64), ("strideO", c_int64),
        ("conv", c_int), ("Nimg", c_int), ("Hin", c_int), ("Win", c_int), ("Cin", c_int),
        ("Hout", c_int), ("Wout", c_int), ("stride", c_int), ("pad", c_int), ("upsample", c_int),
        ("trans_out", c_int),
        ("workspace", c_void_p), ("workspace_bytes", c_int64),
        ("head_dim", c_int),
    ]


# name -> (restype, argtypes); must list every symbol declared in include/aniportrait_hip.h
SIGNATURES = {
    "anip_version": (c_int, []),
    "anip_last_error": (C.c_char_p, []),
    "anip_device_info": (c_int, [C.c_char_p, c_int, C.POINTER(c_int)]),
    "anip_groupnorm_ws_floats": (c_int64, [c_int, c_int64, c_int, c_int]),
    "anip_groupnorm_single_launch": (c_int, [c_int, c_int64, c_int, c_int]),
    "anip_groupnorm": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                               c_int, c_float, c_int, c_void_p, c_void_p]),
    "anip_groupnorm_frames": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                      c_int, c_float, c_int, c_int, c_void_p, c_void_p]),
    "anip_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_int64,
                               c_int, c_void_p]),
    "anip_gemm_workspace_bytes": (c_int64, [C.POINTER(GemmParams)]),
    "anip_gemm": (c_int, [C.POINTER(GemmParams), c_void_p]),
    "anip_ffn_geglu": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                               c_void_p]),
    "anip_ffn_geglu_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_int64, c_int, c_void_p]),
    "anip_conv_direct": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_int, c_int, c_void_p]),
    "anip_batchnorm_ws_floats": (c_int64, [c_int64, c_int]),
    "anip_batchnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float,
                               c_int, c_void_p, c_void_p]),
    "anip_ref_attention": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                   c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                   c_float, c_int64, c_int64, c_void_p]),
    "anip_ref_attention_ex": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                      c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                      c_float, c_int64, c_int64, c_int, c_void_p]),
    "anip_temporal_attention": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "anip_rowgemm320_supported": (c_int, [c_int64, c_int, c_int64]),
    "anip_ln_qkv_projection": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                       c_int64, c_int64, c_int, c_int, c_void_p]),
    "anip_groupnorm_scale_shift": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_float,
                                           c_void_p, c_void_p]),
    "anip_affine_linear320": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "anip_temporal_qkv_attention_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "anip_temporal_qkv_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                            c_int, c_float, c_float, c_void_p]),
    "anip_softmax_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "anip_linear_small": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "anip_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "anip_window_accumulate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64,
                                       c_void_p]),
    "anip_cfg_ddim_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_float, c_float,
                                   c_float, c_float, c_float, c_void_p]),
    "anip_ncfhw_to_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int64, c_void_p]),
    "anip_nhwc_to_ncfhw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_float, c_float, c_int,
                                   c_void_p]),
    "anip_u8_to_f16": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p]),
    "anip_f16_to_u8": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p]),
    "anip_profile_enable": (c_int, [c_int]),
    "anip_profile_collect": (c_int, [c_int, C.POINTER(c_int64), C.POINTER(C.c_double)]),
    "anip_profile_collect_records": (c_int, [c_int, C.POINTER(c_int64), C.POINTER(C.c_double), c_int64,
                                             C.POINTER(c_int), C.POINTER(c_float), C.POINTER(c_int64)]),
    "anip_profile_kernel_name": (C.c_char_p, [c_int]),
}
N_KERNEL_IDS = 12  # ANIP_K_COUNT

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """dlopen the library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64; import it FIRST so that this library's HIP symbols resolve to
    # the runtime torch already initialised (two HIP runtimes in one process do not share devices,
    # streams or allocations: launches then fail with "no ROCm-capable device is detected")
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -m aniportrait_amd.build` (hipcc, gfx950). "
            "There is no CPU or PyTorch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.anip_version()
    if v != ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().anip_last_error()
        raise HipLibraryError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
