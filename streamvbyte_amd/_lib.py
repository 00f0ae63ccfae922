"""Loader for libstreamvbyte.so.2.  Fails loudly: there is no fallback path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVB_LIB_PATH") or os.path.join(HERE, "libstreamvbyte.so.2")

u8p, u32p, i32p, u64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_int32, C.c_uint64))
vp = C.c_void_p

class MultiShard(C.Structure):
    """svbmulti_shard_t of include/streamvbyte_multi.h."""
    _fields_ = [("d_values", vp), ("d_value_off", vp), ("d_bytes", vp), ("d_byte_off", vp), ("d_counts", vp),
                ("d_prev", vp), ("nstreams", C.c_uint32), ("d_sizes_all", vp), ("d_offsets_all", vp)]


# name -> (restype, argtypes); mirrors include/*.h one to one
SIGNATURES = {
    # include/streamvbyte.h
    "streamvbyte_encode": (C.c_size_t, [vp, C.c_uint32, vp]),
    "streamvbyte_encode_0124": (C.c_size_t, [vp, C.c_uint32, vp]),
    "streamvbyte_compressedbytes": (C.c_size_t, [vp, C.c_uint32]),
    "streamvbyte_compressedbytes_0124": (C.c_size_t, [vp, C.c_uint32]),
    "streamvbyte_decode": (C.c_size_t, [vp, vp, C.c_uint32]),
    "streamvbyte_decode_0124": (C.c_size_t, [vp, vp, C.c_uint32]),
    "streamvbyte_validate_stream": (C.c_bool, [vp, C.c_size_t, C.c_uint32]),
    "streamvbyte_validate_stream_0124": (C.c_bool, [vp, C.c_size_t, C.c_uint32]),
    # include/streamvbytedelta.h
    "streamvbyte_delta_encode": (C.c_size_t, [vp, C.c_uint32, vp, C.c_uint32]),
    "streamvbyte_delta_decode": (C.c_size_t, [vp, vp, C.c_uint32, C.c_uint32]),
    # include/streamvbyte_zigzag.h
    "zigzag_encode": (None, [vp, vp, C.c_size_t]),
    "zigzag_decode": (None, [vp, vp, C.c_size_t]),
    "zigzag_delta_encode": (None, [vp, vp, C.c_size_t, C.c_int32]),
    "zigzag_delta_decode": (None, [vp, vp, C.c_size_t, C.c_int32]),
    # include/streamvbyte_gpu.h
    "svbgpu_strerror": (C.c_char_p, [C.c_int]),
    "svbgpu_device_ok": (C.c_int, []),
    "svbgpu_build_has": (C.c_int, [C.c_char_p]),
    "svbgpu_workspace_bytes": (C.c_size_t, [C.c_uint32]),
    "svbgpu_workspace_init": (C.c_int, [vp, C.c_size_t, vp]),
    "svbgpu_workspace_release": (None, [vp]),
    "svbgpu_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "svbgpu_check_status": (C.c_int, [vp, vp]),
    "svbgpu_decode": (C.c_int, [C.c_int, C.c_uint, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp, vp]),
    "svbgpu_decode_plan": (C.c_int, [C.c_int, vp, C.c_uint32, vp, C.c_size_t, vp]),
    "svbgpu_decode_run": (C.c_int, [C.c_int, C.c_uint, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp, vp]),
    "svbgpu_decode_static": (C.c_int, [C.c_int, C.c_uint, vp, vp, C.c_uint32, C.c_uint32, vp, vp]),
    "svbgpu_static_status": (C.c_int, [vp, vp]),
    "svbgpu_encode": (C.c_int, [C.c_int, C.c_uint, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_size_t, vp, vp]),
    "svbgpu_quantize_encode": (C.c_int, [C.c_int, vp, C.c_uint32, C.c_float, vp, vp, C.c_size_t, vp, vp]),
    "svbgpu_decode_dequantize": (C.c_int, [C.c_int, vp, vp, C.c_uint32, C.c_float, C.c_int, vp, C.c_size_t, vp, vp]),
    "svbgpu_decode_intersect": (C.c_int, [C.c_int, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, vp, vp, C.c_size_t,
                                          vp, vp]),
    "svbgpu_intersect_workspace_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "svbgpu_intersect_streams": (C.c_int, [C.c_int, vp, C.c_uint32, C.c_uint32, C.c_int, vp, C.c_uint32, C.c_uint32,
                                           vp, vp, vp, C.c_size_t, vp]),
    "svbgpu_compressedbytes_batch": (C.c_int, [C.c_int, C.c_uint, vp, vp, vp, vp, C.c_uint32, vp, vp]),
    "svbgpu_intersect_batch": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp, vp, vp]),
    "svbgpu_compressedbytes": (C.c_int, [C.c_int, C.c_uint, vp, C.c_uint32, C.c_uint32, vp, vp]),
    "svbgpu_validate": (C.c_int, [C.c_int, vp, C.c_uint64, C.c_uint32, vp, vp, C.c_size_t, vp]),
    "svbgpu_zigzag_encode": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "svbgpu_zigzag_decode": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "svbgpu_zigzag_delta_encode": (C.c_int, [vp, vp, C.c_size_t, C.c_int32, vp]),
    "svbgpu_zigzag_delta_decode": (C.c_int, [vp, vp, C.c_size_t, C.c_int32, vp, C.c_size_t, vp]),
    "svbgpu_encode64": (C.c_int, [C.c_uint, vp, C.c_uint32, vp, C.c_uint64, vp, C.c_size_t, vp, vp]),
    "svbgpu_decode64": (C.c_int, [C.c_uint, vp, vp, C.c_uint32, C.c_uint64, vp, C.c_size_t, vp, vp]),
    "svbgpu_encode64_batch": (C.c_int, [C.c_uint, vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp]),
    "svbgpu_decode64_batch": (C.c_int, [C.c_uint, vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp]),
    "svbgpu_encode_batch": (C.c_int, [C.c_int, C.c_uint, vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp]),
    "svbgpu_decode_batch": (C.c_int, [C.c_int, C.c_uint, vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp]),
    # include/streamvbyte_multi.h
    "svbgpu_partition_streams": (C.c_int, [vp, C.c_uint64, C.c_int, vp]),
    "svbmulti_create": (C.c_int, [vp, C.c_int, vp]),
    "svbmulti_destroy": (None, [vp]),
    "svbmulti_ndev": (C.c_int, [vp]),
    "svbmulti_stream": (vp, [vp, C.c_int]),
    "svbmulti_last_error": (C.c_char_p, []),
    "svbmulti_encode_batch": (C.c_int, [vp, C.c_int, C.c_uint, vp]),
    "svbmulti_decode_batch": (C.c_int, [vp, C.c_int, C.c_uint, vp]),
    "svbmulti_synchronize": (C.c_int, [vp]),
    # include/streamvbyte_frame.h
    "svbgpu_validate_batch": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_uint32, vp, vp]),
    "svbframe_max_bytes": (C.c_size_t, [C.c_uint32]),
    "svbframe_max_bytes_many": (C.c_size_t, [C.c_size_t, C.c_uint32]),
    "svbframe_peek": (C.c_int, [vp, C.c_size_t, vp]),
    "svbframe_scan": (C.c_int, [vp, C.c_size_t, vp, vp]),
    "svbframe_encode_many": (C.c_int, [vp, C.c_size_t, C.c_uint32, C.c_int, C.c_uint, vp, C.c_size_t, vp]),
    "svbframe_decode_many": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, vp]),
    "svbframe_lists_max_bytes": (C.c_size_t, [C.c_size_t, C.c_size_t, C.c_uint32]),
    "svbframe_encode_lists": (C.c_int, [vp, vp, C.c_size_t, C.c_uint32, C.c_int, C.c_uint, vp, C.c_size_t, vp]),
    "svbframe_scan_lists": (C.c_int, [vp, C.c_size_t, vp, vp, vp]),
    "svbframe_decode_lists": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, vp]),
    "svbframe_alloc_pinned": (vp, [C.c_size_t]),
    "svbframe_free_pinned": (None, [vp]),
}


class FrameInfo(C.Structure):
    """svbframe_info of include/streamvbyte_frame.h"""
    _fields_ = [("count", C.c_uint32), ("prev", C.c_uint32), ("variant", C.c_uint8), ("flags", C.c_uint8),
                ("stream_bytes", C.c_uint64), ("frame_bytes", C.c_uint64)]

_lib = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64.so.7.  Two HIP runtimes in one
    process leave the one initialised second without a device, so where torch
    is installed but not imported yet its copy is loaded first; ours then
    resolves to it by soname, exactly as it does when torch was imported
    first.  Without torch the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Return the ctypes handle, building nothing and substituting nothing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m streamvbyte_amd.build` "
                "(hipcc, gfx950). streamvbyte_amd has no CPU implementation to fall back to.")
        _share_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
