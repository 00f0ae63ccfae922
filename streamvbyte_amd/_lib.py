"""Loader for libstreamvbyte.so.2.  Fails loudly: there is no fallback path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVB_LIB_PATH") or os.path.join(HERE, "libstreamvbyte.so.2")

u8p, u32p, i32p, u64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_int32, C.c_uint64))
vp = C.c_void_p

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
    "svbgpu_workspace_bytes": (C.c_size_t, [C.c_uint32]),
    "svbgpu_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "svbgpu_check_status": (C.c_int, [vp, vp]),
    "svbgpu_decode": (C.c_int, [C.c_int, C.c_uint, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp, vp]),
    "svbgpu_decode_plan": (C.c_int, [C.c_int, vp, C.c_uint32, vp, C.c_size_t, vp]),
    "svbgpu_decode_run": (C.c_int, [C.c_int, C.c_uint, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp, vp]),
    "svbgpu_encode": (C.c_int, [C.c_int, C.c_uint, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_size_t, vp, vp]),
    "svbgpu_compressedbytes": (C.c_int, [C.c_int, C.c_uint, vp, C.c_uint32, C.c_uint32, vp, vp]),
    "svbgpu_validate": (C.c_int, [C.c_int, vp, C.c_uint64, C.c_uint32, vp, vp, C.c_size_t, vp]),
    "svbgpu_zigzag_encode": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "svbgpu_zigzag_decode": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "svbgpu_zigzag_delta_encode": (C.c_int, [vp, vp, C.c_size_t, C.c_int32, vp]),
    "svbgpu_zigzag_delta_decode": (C.c_int, [vp, vp, C.c_size_t, C.c_int32, vp, C.c_size_t, vp]),
    "svbgpu_encode_batch": (C.c_int, [C.c_int, C.c_uint, vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp]),
    "svbgpu_decode_batch": (C.c_int, [C.c_int, C.c_uint, vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp]),
}

_lib = None


def load():
    """Return the ctypes handle, building nothing and substituting nothing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m streamvbyte_amd.build` "
                "(hipcc, gfx950). streamvbyte_amd has no CPU implementation to fall back to.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
