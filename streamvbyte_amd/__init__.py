"""streamvbyte_amd -- StreamVByte codec on MI355X (gfx950).

Thin Python face of libstreamvbyte.so.2.  Two layers, both calling straight
through the C ABI:

* the reference's own functions under their own names (`streamvbyte_encode`,
  `streamvbyte_decode`, `streamvbyte_delta_encode`, ... -- reference
  include/streamvbyte.h, streamvbytedelta.h, streamvbyte_zigzag.h), taking
  numpy arrays (host memory) and returning what the C function returns;
* `DeviceCodec`, the device-pointer API of include/streamvbyte_gpu.h on torch
  CUDA(HIP) tensors: no host copies, asynchronous on torch's current stream.

The HIP library is required; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from ._lib import load, LIB_PATH, SIGNATURES  # noqa: F401

V1234, V0124 = 0, 1
DELTA, ZIGZAG = 1, 2
STREAMVBYTE_PADDING = 16


def streamvbyte_max_compressedbytes(length):
    """reference include/streamvbyte.h:33-39"""
    return (length + 3) // 4 + 4 * length + STREAMVBYTE_PADDING


def _np(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _ptr(a):
    return a.ctypes.data if a.size else None


# ---- host-pointer API, reference names ---------------------------------------
def _encode(fn, values, *extra):
    v = _np(values, np.uint32)
    out = np.empty(streamvbyte_max_compressedbytes(v.size), dtype=np.uint8)
    n = fn(_ptr(v), v.size, out.ctypes.data, *extra)
    return out[:n].copy()


def streamvbyte_encode(values):
    return _encode(load().streamvbyte_encode, values)


def streamvbyte_encode_0124(values):
    return _encode(load().streamvbyte_encode_0124, values)


def streamvbyte_delta_encode(values, prev=0):
    return _encode(load().streamvbyte_delta_encode, values, prev)


def _decode(fn, stream, length, *extra):
    s = _np(stream, np.uint8)
    padded = np.zeros(s.size + STREAMVBYTE_PADDING, dtype=np.uint8)
    padded[: s.size] = s
    out = np.empty(length, dtype=np.uint32)
    used = fn(padded.ctypes.data, _ptr(out), length, *extra)
    return out, used


def streamvbyte_decode(stream, length):
    return _decode(load().streamvbyte_decode, stream, length)


def streamvbyte_decode_0124(stream, length):
    return _decode(load().streamvbyte_decode_0124, stream, length)


def streamvbyte_delta_decode(stream, length, prev=0):
    return _decode(load().streamvbyte_delta_decode, stream, length, prev)


def streamvbyte_compressedbytes(values):
    v = _np(values, np.uint32)
    return load().streamvbyte_compressedbytes(_ptr(v), v.size)


def streamvbyte_compressedbytes_0124(values):
    v = _np(values, np.uint32)
    return load().streamvbyte_compressedbytes_0124(_ptr(v), v.size)


def streamvbyte_validate_stream(stream, out_length, in_length=None):
    s = _np(stream, np.uint8)
    return bool(load().streamvbyte_validate_stream(_ptr(s), s.size if in_length is None else in_length, out_length))


def streamvbyte_validate_stream_0124(stream, out_length, in_length=None):
    s = _np(stream, np.uint8)
    return bool(load().streamvbyte_validate_stream_0124(_ptr(s), s.size if in_length is None else in_length, out_length))


def zigzag_encode(values):
    v = _np(values, np.int32)
    out = np.empty(v.size, dtype=np.uint32)
    load().zigzag_encode(_ptr(v), _ptr(out), v.size)
    return out


def zigzag_decode(values):
    v = _np(values, np.uint32)
    out = np.empty(v.size, dtype=np.int32)
    load().zigzag_decode(_ptr(v), _ptr(out), v.size)
    return out


def zigzag_delta_encode(values, prev=0):
    v = _np(values, np.int32)
    out = np.empty(v.size, dtype=np.uint32)
    load().zigzag_delta_encode(_ptr(v), _ptr(out), v.size, prev)
    return out


def zigzag_delta_decode(values, prev=0):
    v = _np(values, np.uint32)
    out = np.empty(v.size, dtype=np.int32)
    load().zigzag_delta_decode(_ptr(v), _ptr(out), v.size, prev)
    return out


class SvbError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise SvbError(f"{what}: {load().svbgpu_strerror(rc).decode()} ({rc})")


# ---- frames (include/streamvbyte_frame.h), host memory -------------------------
FRAME_ERRORS = {-11: "buffer ends inside a frame", -12: "bad frame header", -13: "output buffer too small",
                -14: "a payload fails streamvbyte_validate_stream"}


def _check_frame(rc, what):
    if rc in FRAME_ERRORS:
        raise SvbError(f"{what}: {FRAME_ERRORS[rc]} ({rc})")
    _check(rc, what)


def frame_peek(frames, offset=0):
    """Header of the frame at `offset` of a uint8 array -> FrameInfo."""
    from ._lib import FrameInfo
    f = _np(frames, np.uint8)
    info = FrameInfo()
    _check_frame(load().svbframe_peek(f.ctypes.data + offset, f.size - offset, C.addressof(info)), "svbframe_peek")
    return info


def frame_scan(frames):
    """(number of frames, number of values) of consecutive frames."""
    f = _np(frames, np.uint8)
    nf, nv = C.c_size_t(0), C.c_size_t(0)
    _check_frame(load().svbframe_scan(_ptr(f), f.size, C.addressof(nf), C.addressof(nv)), "svbframe_scan")
    return nf.value, nv.value


def frame_encode_many(values, frame_values, variant=V1234, flags=0):
    """values -> consecutive frames of at most frame_values values (uint8 array)."""
    lib = load()
    v = _np(values, np.uint32)
    cap = lib.svbframe_max_bytes_many(v.size, frame_values)
    out = np.empty(cap, dtype=np.uint8)
    written = C.c_size_t(0)
    _check_frame(lib.svbframe_encode_many(_ptr(v), v.size, frame_values, variant, flags, _ptr(out), cap,
                                          C.addressof(written)), "svbframe_encode_many")
    return out[:written.value].copy()


def frame_decode_many(frames):
    """consecutive frames -> uint32 array of all their values."""
    lib = load()
    f = _np(frames, np.uint8)
    n = C.c_size_t(0)
    _check_frame(lib.svbframe_decode_many(_ptr(f), f.size, None, 0, C.addressof(n)), "svbframe_decode_many")
    out = np.empty(n.value, dtype=np.uint32)
    _check_frame(lib.svbframe_decode_many(_ptr(f), f.size, _ptr(out), out.size, C.addressof(n)), "svbframe_decode_many")
    return out


def frame_scan_lists(frames):
    """(frames, lists, values) of consecutive list frames."""
    f = _np(frames, np.uint8)
    nf, nl, nv = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    _check_frame(load().svbframe_scan_lists(_ptr(f), f.size, C.addressof(nf), C.addressof(nl), C.addressof(nv)),
                 "svbframe_scan_lists")
    return nf.value, nl.value, nv.value


def frame_encode_lists(values, list_start, frame_values, variant=V1234, flags=0):
    """Lists values[list_start[i]:list_start[i+1]] -> consecutive list frames (uint8 array)."""
    lib = load()
    v = _np(values, np.uint32)
    ls = _np(list_start, np.uint64)
    n_lists = max(ls.size - 1, 0)
    n_values = int(ls[-1] - ls[0]) if ls.size else 0
    cap = lib.svbframe_lists_max_bytes(n_values, n_lists, frame_values)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    written = C.c_size_t(0)
    _check_frame(lib.svbframe_encode_lists(_ptr(v), _ptr(ls), n_lists, frame_values, variant, flags, _ptr(out), cap,
                                           C.addressof(written)), "svbframe_encode_lists")
    return out[:written.value].copy()


def frame_decode_lists(frames):
    """consecutive list frames -> (uint32 values, uint64 list_start with one entry more than lists)."""
    lib = load()
    f = _np(frames, np.uint8)
    nl, nv = C.c_size_t(0), C.c_size_t(0)
    _check_frame(lib.svbframe_decode_lists(_ptr(f), f.size, None, 0, None, 0, C.addressof(nl), C.addressof(nv)),
                 "svbframe_decode_lists")
    out = np.empty(max(nv.value, 1), dtype=np.uint32)
    ls = np.zeros(nl.value + 1, dtype=np.uint64)
    _check_frame(lib.svbframe_decode_lists(_ptr(f), f.size, _ptr(out), nv.value, _ptr(ls), nl.value, C.addressof(nl),
                                           C.addressof(nv)), "svbframe_decode_lists")
    return out[:nv.value], ls


class DeviceCodec:
    """Device-pointer API on torch tensors (include/streamvbyte_gpu.h).

    Holds the workspace and the 8-byte result words; every call is enqueued on
    torch's current stream of `device` and returns without synchronising.
    """

    def __init__(self, max_count, device="cuda:0", zero=True):
        """zero=False skips the two fills of a fresh workspace and result block (every call clears what
        it needs itself); for codecs made per call, whose memory comes from torch's caching allocator."""
        import torch
        self.torch = torch
        self.lib = load()
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            if not self.lib.svbgpu_device_ok():
                raise SvbError("no usable gfx950 device")
        self.max_count = int(max_count)
        self.ws_bytes = self.lib.svbgpu_workspace_bytes(self.max_count)
        make = torch.zeros if zero else torch.empty
        self.ws = make(self.ws_bytes, dtype=torch.uint8, device=self.device)
        assert self.ws.data_ptr() % 256 == 0
        self.res = make(8, dtype=torch.int64, device=self.device)
        # a codec that keeps its workspace has it initialised once, so that its calls skip the clearing
        # launch (svbgpu_workspace_init); a per-call codec (zero=False) leaves that to every call
        self._registered = False
        if zero:
            with torch.cuda.device(self.device):
                _check(self.lib.svbgpu_workspace_init(self.ws.data_ptr(), self.ws_bytes, self._stream()),
                       "svbgpu_workspace_init")
            self._registered = True

    def __del__(self):
        try:
            if getattr(self, "_registered", False):
                self.lib.svbgpu_workspace_release(self.ws.data_ptr())
        except Exception:
            pass

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def _res_ptr(self, i):
        return self.res.data_ptr() + 8 * i

    def encode(self, values, out=None, variant=V1234, flags=0, prev=0):
        """values: uint32/int32 tensor; returns (out_bytes, size_tensor[1])."""
        torch = self.torch
        n = values.numel()
        assert n <= self.max_count and values.is_contiguous()
        if out is None:
            out = torch.empty(streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_encode(variant, flags, values.data_ptr(), n, out.data_ptr(), prev & 0xFFFFFFFF,
                                          self.ws.data_ptr(), self.ws_bytes, self._res_ptr(0), self._stream()),
                   "svbgpu_encode")
        return out, self.res[0:1]

    def decode(self, stream, count, out=None, variant=V1234, flags=0, prev=0):
        """stream: uint8 tensor readable 16 B past the stream; returns (values, consumed_tensor[1])."""
        torch = self.torch
        assert count <= self.max_count and stream.is_contiguous()
        if out is None:
            out = torch.empty(count, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_decode(variant, flags, stream.data_ptr(), out.data_ptr(), count, prev & 0xFFFFFFFF,
                                          self.ws.data_ptr(), self.ws_bytes, self._res_ptr(1), self._stream()),
                   "svbgpu_decode")
        return out, self.res[1:2]

    def encode64(self, values, out=None, flags=0, prev=0):
        """int64 tensor (the uint64 bits) -> 1248 stream; returns (out_bytes, size_tensor[1])."""
        torch = self.torch
        n = values.numel()
        assert n <= self.max_count and values.is_contiguous() and values.dtype == torch.int64
        if out is None:
            out = torch.empty((n + 3) // 4 + 8 * n + 32, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_encode64(flags, values.data_ptr(), n, out.data_ptr(), prev & 0xFFFFFFFFFFFFFFFF,
                                            self.ws.data_ptr(), self.ws_bytes, self._res_ptr(0), self._stream()),
                   "svbgpu_encode64")
        return out, self.res[0:1]

    def decode64(self, stream, count, out=None, flags=0, prev=0):
        """1248 stream (readable 32 B past its end) -> int64 tensor; returns (values, consumed_tensor[1])."""
        torch = self.torch
        assert count <= self.max_count and stream.is_contiguous()
        if out is None:
            out = torch.empty(count, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_decode64(flags, stream.data_ptr(), out.data_ptr(), count, prev & 0xFFFFFFFFFFFFFFFF,
                                            self.ws.data_ptr(), self.ws_bytes, self._res_ptr(1), self._stream()),
                   "svbgpu_decode64")
        return out, self.res[1:2]

    def quantize_encode(self, values, scale, out=None, variant=V0124):
        """float32 tensor -> stream of zigzag(round(values * scale)); returns (out_bytes, size_tensor[1])."""
        torch = self.torch
        n = values.numel()
        assert n <= self.max_count and values.is_contiguous() and values.dtype == torch.float32
        if out is None:
            out = torch.empty(streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_quantize_encode(variant, values.data_ptr(), n, scale, out.data_ptr(),
                                                   self.ws.data_ptr(), self.ws_bytes, self._res_ptr(0), self._stream()),
                   "svbgpu_quantize_encode")
        return out, self.res[0:1]

    def decode_dequantize(self, stream, count, scale, out=None, accumulate=False, variant=V0124):
        """stream -> float32 tensor (int value * scale), optionally added to `out`."""
        torch = self.torch
        assert count <= self.max_count and stream.is_contiguous()
        if out is None:
            assert not accumulate
            out = torch.empty(count, dtype=torch.float32, device=self.device)
        assert out.dtype == torch.float32 and out.is_contiguous()
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_decode_dequantize(variant, stream.data_ptr(), out.data_ptr(), count, scale,
                                                     int(accumulate), self.ws.data_ptr(), self.ws_bytes,
                                                     self._res_ptr(1), self._stream()), "svbgpu_decode_dequantize")
        return out, self.res[1:2]

    def decode_intersect(self, stream, count, sorted_values, out=None, variant=V1234, prev=0):
        """delta-coded stream x ascending uint32/int32-bits tensor -> (matches buffer, n_matches_tensor[1],
        consumed_tensor[1]); the first n_matches entries of the buffer are the stream's values found in
        sorted_values, in stream order."""
        torch = self.torch
        assert count <= self.max_count and stream.is_contiguous() and sorted_values.is_contiguous()
        if out is None:
            out = torch.empty(max(count, 1), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_decode_intersect(variant, stream.data_ptr(), count, prev & 0xFFFFFFFF,
                                                    sorted_values.data_ptr(), sorted_values.numel(), out.data_ptr(),
                                                    self._res_ptr(4), self.ws.data_ptr(), self.ws_bytes,
                                                    self._res_ptr(1), self._stream()), "svbgpu_decode_intersect")
        return out, self.res[4:5], self.res[1:2]

    def intersect_streams(self, stream_a, count_a, stream_b, count_b, out=None, variant_a=V1234, prev_a=0,
                          variant_b=V1234, prev_b=0):
        """Two delta-coded streams (B ascending) -> (matches buffer, n_matches_tensor[1]): the values of A
        that occur in B, in A's order; neither stream is written out decoded."""
        torch = self.torch
        assert stream_a.is_contiguous() and stream_b.is_contiguous()
        need = self.lib.svbgpu_intersect_workspace_bytes(count_a, count_b)
        ws = getattr(self, "_sx_ws", None)
        if ws is None or ws.numel() < need:
            ws = self._sx_ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty(max(count_a, 1), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_intersect_streams(variant_a, stream_a.data_ptr(), count_a, prev_a & 0xFFFFFFFF,
                                                     variant_b, stream_b.data_ptr(), count_b, prev_b & 0xFFFFFFFF,
                                                     out.data_ptr(), self._res_ptr(4), ws.data_ptr(), ws.numel(),
                                                     self._stream()), "svbgpu_intersect_streams")
        return out, self.res[4:5]

    def intersect_batch(self, bytes_a, byte_off_a, counts_a, prevs_a, bytes_b, byte_off_b, counts_b, prevs_b,
                        out, out_off, variant=V1234, n_out=None):
        """Pairs of delta-coded lists (device tensors: uint8 bytes, int64 offsets, int32 counts, int32 seeds or
        None) -> (out, n_out int32[npairs]): pair i's common values at out[out_off[i]:], in A_i's order."""
        torch = self.torch
        npairs = counts_a.numel()
        if n_out is None:
            n_out = torch.empty(max(npairs, 1), dtype=torch.int32, device=self.device)
        ptr = lambda t: 0 if t is None else t.data_ptr()
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_intersect_batch(variant, ptr(bytes_a), ptr(byte_off_a), ptr(counts_a), ptr(prevs_a),
                                                   ptr(bytes_b), ptr(byte_off_b), ptr(counts_b), ptr(prevs_b), npairs,
                                                   ptr(out), ptr(out_off), ptr(n_out), self._stream()),
                   "svbgpu_intersect_batch")
        return out, n_out[:npairs]

    def decode_plan(self, stream, count, variant=V1234):
        with self.torch.cuda.device(self.device):
            _check(self.lib.svbgpu_decode_plan(variant, stream.data_ptr(), count, self.ws.data_ptr(),
                                               self.ws_bytes, self._stream()), "svbgpu_decode_plan")

    def decode_run(self, stream, count, out, variant=V1234, flags=0, prev=0):
        with self.torch.cuda.device(self.device):
            _check(self.lib.svbgpu_decode_run(variant, flags, stream.data_ptr(), out.data_ptr(), count,
                                              prev & 0xFFFFFFFF, self.ws.data_ptr(), self.ws_bytes,
                                              self._res_ptr(1), self._stream()), "svbgpu_decode_run")
        return out, self.res[1:2]

    def compressedbytes(self, values, variant=V1234, flags=0, prev=0):
        with self.torch.cuda.device(self.device):
            _check(self.lib.svbgpu_compressedbytes(variant, flags, values.data_ptr(), values.numel(),
                                                   prev & 0xFFFFFFFF, self._res_ptr(2), self._stream()),
                   "svbgpu_compressedbytes")
        return self.res[2:3]

    def validate(self, stream, in_length, count, variant=V1234):
        with self.torch.cuda.device(self.device):
            _check(self.lib.svbgpu_validate(variant, stream.data_ptr(), in_length, count, self._res_ptr(3),
                                            self.ws.data_ptr(), self.ws_bytes, self._stream()), "svbgpu_validate")
        return self.res[3:4]

    def zigzag(self, values, mode, prev=0, out=None):
        """mode: 'encode' | 'decode' | 'delta_encode' | 'delta_decode'"""
        torch = self.torch
        n = values.numel()
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device=self.device)
        st = self._stream()
        with torch.cuda.device(self.device):
            if mode == "encode":
                rc = self.lib.svbgpu_zigzag_encode(values.data_ptr(), out.data_ptr(), n, st)
            elif mode == "decode":
                rc = self.lib.svbgpu_zigzag_decode(values.data_ptr(), out.data_ptr(), n, st)
            elif mode == "delta_encode":
                rc = self.lib.svbgpu_zigzag_delta_encode(values.data_ptr(), out.data_ptr(), n, prev, st)
            elif mode == "delta_decode":
                rc = self.lib.svbgpu_zigzag_delta_decode(values.data_ptr(), out.data_ptr(), n, prev,
                                                         self.ws.data_ptr(), self.ws_bytes, st)
            else:
                raise ValueError(mode)
        _check(rc, "svbgpu_zigzag_" + mode)
        return out

    def encode_batch(self, values, value_off, byte_buf, byte_off, counts, prevs=None, variant=V1234, flags=0,
                     sizes=None):
        torch = self.torch
        ns = counts.numel()
        if sizes is None:
            sizes = torch.empty(ns, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_encode_batch(variant, flags, values.data_ptr(), value_off.data_ptr(),
                                                byte_buf.data_ptr(), byte_off.data_ptr(), counts.data_ptr(),
                                                prevs.data_ptr() if prevs is not None else None, ns,
                                                sizes.data_ptr(), self._stream()), "svbgpu_encode_batch")
        return sizes

    def compressedbytes_batch(self, values, value_off, counts, prevs=None, variant=V1234, flags=0, sizes=None):
        """Exact encoded size of every stream of a batch (int64 tensor), without encoding."""
        torch = self.torch
        ns = counts.numel()
        if sizes is None:
            sizes = torch.empty(ns, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_compressedbytes_batch(variant, flags, values.data_ptr(), value_off.data_ptr(),
                                                         counts.data_ptr(),
                                                         prevs.data_ptr() if prevs is not None else None, ns,
                                                         sizes.data_ptr(), self._stream()),
                   "svbgpu_compressedbytes_batch")
        return sizes

    def decode_batch(self, byte_buf, byte_off, values, value_off, counts, prevs=None, variant=V1234, flags=0,
                     sizes=None):
        torch = self.torch
        ns = counts.numel()
        if sizes is None:
            sizes = torch.empty(ns, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_decode_batch(variant, flags, byte_buf.data_ptr(), byte_off.data_ptr(),
                                                values.data_ptr(), value_off.data_ptr(), counts.data_ptr(),
                                                prevs.data_ptr() if prevs is not None else None, ns,
                                                sizes.data_ptr(), self._stream()), "svbgpu_decode_batch")
        return sizes

    def encode64_batch(self, values, value_off, byte_buf, byte_off, counts, prevs=None, flags=0, sizes=None):
        """svbgpu_encode64_batch: int64 `values`, one 1248 stream per entry of `counts`; prevs int64 or None."""
        torch = self.torch
        ns = counts.numel()
        if sizes is None:
            sizes = torch.empty(ns, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_encode64_batch(flags, values.data_ptr(), value_off.data_ptr(), byte_buf.data_ptr(),
                                                  byte_off.data_ptr(), counts.data_ptr(),
                                                  prevs.data_ptr() if prevs is not None else None, ns,
                                                  sizes.data_ptr(), self._stream()), "svbgpu_encode64_batch")
        return sizes

    def decode64_batch(self, byte_buf, byte_off, values, value_off, counts, prevs=None, flags=0, sizes=None):
        torch = self.torch
        ns = counts.numel()
        if sizes is None:
            sizes = torch.empty(ns, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_decode64_batch(flags, byte_buf.data_ptr(), byte_off.data_ptr(), values.data_ptr(),
                                                  value_off.data_ptr(), counts.data_ptr(),
                                                  prevs.data_ptr() if prevs is not None else None, ns,
                                                  sizes.data_ptr(), self._stream()), "svbgpu_decode64_batch")
        return sizes

    def validate_batch(self, byte_buf, byte_off, lengths, counts, variant=V1234):
        """Per-stream streamvbyte_validate_stream: int32 tensor of 1 / 0."""
        torch = self.torch
        ns = counts.numel()
        ok = torch.empty(ns, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.svbgpu_validate_batch(variant, byte_buf.data_ptr(), byte_off.data_ptr(), lengths.data_ptr(),
                                                  counts.data_ptr(), ns, ok.data_ptr(), self._stream()),
                   "svbgpu_validate_batch")
        return ok

    def check(self):
        """Synchronise and raise if a device scan gave up (never expected)."""
        with self.torch.cuda.device(self.device):
            _check(self.lib.svbgpu_check_status(self.ws.data_ptr(), self._stream()), "svbgpu_check_status")
            sx = getattr(self, "_sx_ws", None)  # the workspace of intersect_streams, if it was ever used
            if sx is not None:
                _check(self.lib.svbgpu_check_status(sx.data_ptr(), self._stream()), "svbgpu_check_status")
