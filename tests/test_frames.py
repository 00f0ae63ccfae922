"""Frames (include/streamvbyte_frame.h): header handling on the CPU, the
pipelined encode / decode of frame sequences on the GPU against the oracle."""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle import V1234, V0124
from util import GENERATORS

MAGIC = 0x46425653


def header(count, stream_bytes, variant=0, flags=0, prev=0, magic=MAGIC, version=1):
    return struct.pack("<IBBBBIIQQ", magic, version, variant, flags, 0, count, prev, stream_bytes, 0)


def test_header_rules_on_cpu():
    """svbframe_peek / svbframe_scan / size bounds need no GPU."""
    import streamvbyte_amd as svb
    lib = svb.load()
    assert lib.svbframe_max_bytes(0) == 32 + 16
    assert lib.svbframe_max_bytes(5) == 32 + ((2 + 20 + 16 + 15) // 16) * 16
    assert lib.svbframe_max_bytes_many(10, 4) == 2 * lib.svbframe_max_bytes(4) + lib.svbframe_max_bytes(2)
    good = np.frombuffer(header(5, 9) + bytes(16), dtype=np.uint8)
    info = svb.frame_peek(good)
    assert (info.count, info.stream_bytes, info.frame_bytes, info.variant, info.flags) == (5, 9, 48, 0, 0)
    assert svb.frame_scan(np.concatenate([good, good])) == (2, 10)
    # the last frame may come without padding
    assert svb.frame_peek(good[:41]).frame_bytes == 41
    for bad, code in ((good[:40], -11), (good[:20], -11),
                      (np.frombuffer(header(5, 9, magic=0x12345678) + bytes(16), dtype=np.uint8), -12),
                      (np.frombuffer(header(5, 9, version=2) + bytes(16), dtype=np.uint8), -12),
                      (np.frombuffer(header(5, 9, variant=2) + bytes(16), dtype=np.uint8), -12),
                      (np.frombuffer(header(5, 9, flags=4) + bytes(16), dtype=np.uint8), -12),
                      (np.frombuffer(header(5, 1) + bytes(16), dtype=np.uint8), -12),    # shorter than its keys
                      (np.frombuffer(header(5, 23) + bytes(32), dtype=np.uint8), -12),   # longer than 4 B per value
                      (np.frombuffer(header(5, 9, prev=7) + bytes(16), dtype=np.uint8), -12)):  # prev without delta
        with pytest.raises(svb.SvbError) as e:
            svb.frame_peek(bad)
        assert f"({code})" in str(e.value)
    assert svb.frame_scan(np.zeros(0, dtype=np.uint8)) == (0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("variant,flags", [(V1234, 0), (V0124, 0), (V1234, 1), (V0124, 2), (V1234, 3)])
def test_frames_match_oracle(oracle, variant, flags):
    import streamvbyte_amd as svb
    rng = np.random.default_rng(31 + variant + 4 * flags)
    for n, fv in ((1, 4), (4096, 1000), (100003, 16384), (300000, 65536), (70000, 70000)):
        # narrow deltas take the two-pass differential decoder, wide ones the look-back decoder
        v = GENERATORS["postings" if (flags & 1) and n != 100003 else "width_mix"](rng, n)
        frames = svb.frame_encode_many(v, fv, variant=variant, flags=flags)
        assert svb.frame_scan(frames) == ((n + fv - 1) // fv, n)
        off = 0
        for i in range((n + fv - 1) // fv):
            info = svb.frame_peek(frames, off)
            chunk = v[i * fv:(i + 1) * fv]
            prev = int(v[i * fv - 1]) if (flags & 1) and i else 0
            assert (info.count, info.variant, info.flags, info.prev) == (chunk.size, variant, flags, prev)
            if flags & 2:
                sprev = prev - (1 << 32) if prev >= (1 << 31) else prev
                pre = oracle.zigzag_delta_encode(chunk.view(np.int32), sprev) if flags & 1 else oracle.zigzag_encode(chunk.view(np.int32))
                expect = oracle.encode(variant, pre)
            else:
                expect = oracle.encode(variant, chunk, bool(flags & 1), prev)
            assert info.stream_bytes == expect.size
            assert np.array_equal(frames[off + 32: off + 32 + expect.size], expect), (n, fv, i)
            assert not frames[off + 32 + expect.size: off + info.frame_bytes].any()  # zero padding
            off += info.frame_bytes
        assert off == frames.size
        assert np.array_equal(svb.frame_decode_many(frames), v)


@pytest.mark.gpu
def test_frames_reject_bad_input(oracle):
    import streamvbyte_amd as svb
    v = GENERATORS["width_mix"](np.random.default_rng(5), 50000)
    frames = svb.frame_encode_many(v, 20000)
    # a payload whose keys promise more data than the header says: caught on the device, never decoded
    bad = frames.copy()
    info = svb.frame_peek(bad, 0)
    second = int(info.frame_bytes)
    bad[second + 32: second + 32 + 100] = 0xFF
    with pytest.raises(svb.SvbError) as e:
        svb.frame_decode_many(bad)
    assert "(-14)" in str(e.value)
    with pytest.raises(svb.SvbError) as e:
        svb.frame_decode_many(frames[: frames.size - 40])
    assert "(-11)" in str(e.value)
    lib = svb.load()
    n = C.c_size_t(0)
    out = np.empty(10, dtype=np.uint32)
    assert lib.svbframe_decode_many(frames.ctypes.data, frames.size, out.ctypes.data, out.size, C.addressof(n)) == -13
    assert n.value == 50000
    # pinned buffers work the same
    p = lib.svbframe_alloc_pinned(frames.size)
    assert p
    C.memmove(p, frames.ctypes.data, frames.size)
    q = lib.svbframe_alloc_pinned(4 * v.size)
    assert lib.svbframe_decode_many(p, frames.size, q, v.size, C.addressof(n)) == 0
    got = np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_uint32)), shape=(v.size,)).copy()
    assert np.array_equal(got, v)
    lib.svbframe_free_pinned(p)
    lib.svbframe_free_pinned(q)
