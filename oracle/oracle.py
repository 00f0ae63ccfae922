"""ctypes front-ends for the two CPU checkers (test infrastructure only).

`Oracle()`     -> the scalar restatement, oracle/libsvb_oracle.so (svb_oracle.c)
`Reference()`  -> the reference library compiled from its own sources into
                  oracle/_ref/libstreamvbyte_ref.so by oracle/Makefile
                  (raises FileNotFoundError when that build is absent)

Both expose the same numpy-level methods so tests can swap one for the other.
Nothing in the product package imports this module.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
V1234, V0124 = 0, 1

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)


def _p(a, t):
    return a.ctypes.data_as(t)


def max_compressedbytes(n):
    """include/streamvbyte.h:33-39 of the reference."""
    return (n + 3) // 4 + 4 * n + 16


class _Base:
    def encode(self, variant, values, delta=False, prev=0):
        values = np.ascontiguousarray(values, dtype=np.uint32)
        n = values.size
        out = np.zeros(max_compressedbytes(n), dtype=np.uint8)
        size = self._encode(variant, values, n, out, delta, prev)
        return out[:size].copy()

    def decode(self, variant, stream, n, delta=False, prev=0):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        padded = np.zeros(stream.size + 16, dtype=np.uint8)  # STREAMVBYTE_PADDING
        padded[: stream.size] = stream
        out = np.zeros(n, dtype=np.uint32)
        used = self._decode(variant, padded, out, n, delta, prev)
        return out, used


class Oracle(_Base):
    kind = "port"

    def __init__(self):
        path = os.path.join(HERE, "libsvb_oracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle`)")
        L = self.lib = C.CDLL(path)
        L.svbo_encode.restype = C.c_size_t
        L.svbo_encode.argtypes = [C.c_int, _u32p, C.c_uint32, _u8p, C.c_int, C.c_uint32]
        L.svbo_decode.restype = C.c_size_t
        L.svbo_decode.argtypes = [C.c_int, _u8p, _u32p, C.c_uint32, C.c_int, C.c_uint32]
        L.svbo_compressedbytes.restype = C.c_size_t
        L.svbo_compressedbytes.argtypes = [C.c_int, _u32p, C.c_uint32]
        L.svbo_validate.restype = C.c_int
        L.svbo_validate.argtypes = [C.c_int, _u8p, C.c_size_t, C.c_uint32]
        _u64p = C.POINTER(C.c_uint64)
        L.svbo_encode64.restype = C.c_size_t
        L.svbo_encode64.argtypes = [_u64p, C.c_uint32, _u8p]
        L.svbo_decode64.restype = C.c_size_t
        L.svbo_decode64.argtypes = [_u8p, _u64p, C.c_uint32]
        for nm, a, b in (("svbo_zigzag_encode", _i32p, _u32p), ("svbo_zigzag_decode", _u32p, _i32p)):
            getattr(L, nm).restype = None
            getattr(L, nm).argtypes = [a, b, C.c_size_t]
        for nm, a, b in (("svbo_zigzag_delta_encode", _i32p, _u32p),
                         ("svbo_zigzag_delta_decode", _u32p, _i32p)):
            getattr(L, nm).restype = None
            getattr(L, nm).argtypes = [a, b, C.c_size_t, C.c_int32]

    def _encode(self, variant, values, n, out, delta, prev):
        return self.lib.svbo_encode(variant, _p(values, _u32p), n, _p(out, _u8p), int(delta), prev)

    def _decode(self, variant, stream, out, n, delta, prev):
        return self.lib.svbo_decode(variant, _p(stream, _u8p), _p(out, _u32p), n, int(delta), prev)

    def encode64(self, values):
        """64-bit values, 1248 variant (no reference counterpart: parity unpinned)"""
        values = np.ascontiguousarray(values, dtype=np.uint64)
        out = np.zeros((values.size + 3) // 4 + 8 * values.size + 32, dtype=np.uint8)
        size = self.lib.svbo_encode64(_p(values, C.POINTER(C.c_uint64)), values.size, _p(out, _u8p))
        return out[:size].copy()

    def decode64(self, stream, n):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        padded = np.zeros(stream.size + 32, dtype=np.uint8)
        padded[: stream.size] = stream
        out = np.zeros(n, dtype=np.uint64)
        used = self.lib.svbo_decode64(_p(padded, _u8p), _p(out, C.POINTER(C.c_uint64)), n)
        return out, used

    def compressedbytes(self, variant, values):
        values = np.ascontiguousarray(values, dtype=np.uint32)
        return self.lib.svbo_compressedbytes(variant, _p(values, _u32p), values.size)

    def validate(self, variant, stream, n, inlen=None):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        buf = np.zeros(max(stream.size, (n + 3) // 4) + 16, dtype=np.uint8)
        buf[: stream.size] = stream
        inlen = stream.size if inlen is None else inlen
        return bool(self.lib.svbo_validate(variant, _p(buf, _u8p), inlen, n))

    def zigzag_encode(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        o = np.empty(a.size, dtype=np.uint32)
        self.lib.svbo_zigzag_encode(_p(a, _i32p), _p(o, _u32p), a.size)
        return o

    def zigzag_decode(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        o = np.empty(a.size, dtype=np.int32)
        self.lib.svbo_zigzag_decode(_p(a, _u32p), _p(o, _i32p), a.size)
        return o

    def zigzag_delta_encode(self, a, prev=0):
        a = np.ascontiguousarray(a, dtype=np.int32)
        o = np.empty(a.size, dtype=np.uint32)
        self.lib.svbo_zigzag_delta_encode(_p(a, _i32p), _p(o, _u32p), a.size, prev)
        return o

    def zigzag_delta_decode(self, a, prev=0):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        o = np.empty(a.size, dtype=np.int32)
        self.lib.svbo_zigzag_delta_decode(_p(a, _u32p), _p(o, _i32p), a.size, prev)
        return o


class Reference(_Base):
    """The reference's 14 exported functions (include/streamvbyte.h:23-79,
    include/streamvbytedelta.h:19,29, include/streamvbyte_zigzag.h:14-29)."""
    kind = "reference"

    def __init__(self):
        path = os.path.join(HERE, "_ref", "libstreamvbyte_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = self.lib = C.CDLL(path)
        for nm in ("streamvbyte_encode", "streamvbyte_encode_0124"):
            getattr(L, nm).restype = C.c_size_t
            getattr(L, nm).argtypes = [_u32p, C.c_uint32, _u8p]
        for nm in ("streamvbyte_decode", "streamvbyte_decode_0124"):
            getattr(L, nm).restype = C.c_size_t
            getattr(L, nm).argtypes = [_u8p, _u32p, C.c_uint32]
        for nm in ("streamvbyte_compressedbytes", "streamvbyte_compressedbytes_0124"):
            getattr(L, nm).restype = C.c_size_t
            getattr(L, nm).argtypes = [_u32p, C.c_uint32]
        for nm in ("streamvbyte_validate_stream", "streamvbyte_validate_stream_0124"):
            getattr(L, nm).restype = C.c_bool
            getattr(L, nm).argtypes = [_u8p, C.c_size_t, C.c_uint32]
        L.streamvbyte_delta_encode.restype = C.c_size_t
        L.streamvbyte_delta_encode.argtypes = [_u32p, C.c_uint32, _u8p, C.c_uint32]
        L.streamvbyte_delta_decode.restype = C.c_size_t
        L.streamvbyte_delta_decode.argtypes = [_u8p, _u32p, C.c_uint32, C.c_uint32]
        L.zigzag_encode.restype = None
        L.zigzag_encode.argtypes = [_i32p, _u32p, C.c_size_t]
        L.zigzag_decode.restype = None
        L.zigzag_decode.argtypes = [_u32p, _i32p, C.c_size_t]
        L.zigzag_delta_encode.restype = None
        L.zigzag_delta_encode.argtypes = [_i32p, _u32p, C.c_size_t, C.c_int32]
        L.zigzag_delta_decode.restype = None
        L.zigzag_delta_decode.argtypes = [_u32p, _i32p, C.c_size_t, C.c_int32]

    def _encode(self, variant, values, n, out, delta, prev):
        if delta:
            assert variant == V1234, "the reference has no delta_0124"
            return self.lib.streamvbyte_delta_encode(_p(values, _u32p), n, _p(out, _u8p), prev)
        f = self.lib.streamvbyte_encode if variant == V1234 else self.lib.streamvbyte_encode_0124
        return f(_p(values, _u32p), n, _p(out, _u8p))

    def _decode(self, variant, stream, out, n, delta, prev):
        if delta:
            assert variant == V1234
            return self.lib.streamvbyte_delta_decode(_p(stream, _u8p), _p(out, _u32p), n, prev)
        f = self.lib.streamvbyte_decode if variant == V1234 else self.lib.streamvbyte_decode_0124
        return f(_p(stream, _u8p), _p(out, _u32p), n)

    def compressedbytes(self, variant, values):
        values = np.ascontiguousarray(values, dtype=np.uint32)
        f = (self.lib.streamvbyte_compressedbytes if variant == V1234
             else self.lib.streamvbyte_compressedbytes_0124)
        return f(_p(values, _u32p), values.size)

    def validate(self, variant, stream, n, inlen=None):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        buf = np.zeros(max(stream.size, (n + 3) // 4) + 16, dtype=np.uint8)
        buf[: stream.size] = stream
        inlen = stream.size if inlen is None else inlen
        f = (self.lib.streamvbyte_validate_stream if variant == V1234
             else self.lib.streamvbyte_validate_stream_0124)
        return bool(f(_p(buf, _u8p), inlen, n))

    def zigzag_encode(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        o = np.empty(a.size, dtype=np.uint32)
        self.lib.zigzag_encode(_p(a, _i32p), _p(o, _u32p), a.size)
        return o

    def zigzag_decode(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        o = np.empty(a.size, dtype=np.int32)
        self.lib.zigzag_decode(_p(a, _u32p), _p(o, _i32p), a.size)
        return o

    def zigzag_delta_encode(self, a, prev=0):
        a = np.ascontiguousarray(a, dtype=np.int32)
        o = np.empty(a.size, dtype=np.uint32)
        self.lib.zigzag_delta_encode(_p(a, _i32p), _p(o, _u32p), a.size, prev)
        return o

    def zigzag_delta_decode(self, a, prev=0):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        o = np.empty(a.size, dtype=np.int32)
        self.lib.zigzag_delta_decode(_p(a, _u32p), _p(o, _i32p), a.size, prev)
        return o
