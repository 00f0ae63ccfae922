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


# ---- list frames: many lists per frame behind a directory ---------------------------
LIST_MAGIC = 0x4C425653


def pad16(b):
    return b + bytes(-len(b) % 16)


def list_frame(counts, sizes, payload, variant=0, flags=0, magic=LIST_MAGIC, version=1, n_values=None, payload_bytes=None):
    n_values = sum(counts) if n_values is None else n_values
    payload_bytes = len(payload) if payload_bytes is None else payload_bytes
    head = struct.pack("<IBBBBIIQQ", magic, version, variant, flags, 0, len(counts), 0, n_values, payload_bytes)
    return (head + pad16(np.asarray(counts, dtype="<u4").tobytes()) + pad16(np.asarray(sizes, dtype="<u4").tobytes())
            + pad16(payload))


def expected_list_frames(oracle, v, list_start, frame_values, variant, flags):
    """The frames svbframe_encode_lists must write, built from the oracle's streams."""
    def stream(chunk):
        if flags & 2:
            s = chunk.view(np.int32)
            pre = oracle.zigzag_delta_encode(s, 0) if flags & 1 else oracle.zigzag_encode(s)
            return oracle.encode(variant, pre).tobytes()
        return oracle.encode(variant, chunk, bool(flags & 1), 0).tobytes()
    out, cur_c, cur_s, cur_p = b"", [], [], b""
    for i in range(len(list_start) - 1):
        chunk = v[int(list_start[i]):int(list_start[i + 1])]
        if cur_c and sum(cur_c) + chunk.size > frame_values:
            out += list_frame(cur_c, cur_s, cur_p, variant, flags)
            cur_c, cur_s, cur_p = [], [], b""
        b = stream(chunk) if chunk.size else b""
        cur_c.append(chunk.size)
        cur_s.append(len(b))
        cur_p += b
    return out + list_frame(cur_c, cur_s, cur_p, variant, flags)


def test_list_header_rules_on_cpu():
    import streamvbyte_amd as svb
    lib = svb.load()
    assert lib.svbframe_lists_max_bytes(0, 0, 100) == 0
    good = np.frombuffer(list_frame([5, 0, 2], [9, 0, 3], bytes(12)), dtype=np.uint8)
    assert good.size == 32 + 16 + 16 + 16
    assert svb.frame_scan_lists(np.concatenate([good, good])) == (2, 6, 14)
    assert svb.frame_scan_lists(good[:32 + 32 + 12]) == (1, 3, 7)  # last frame without its padding
    for bad, code in ((good[:32 + 32 + 11], -11), (good[:20], -11),
                      (list_frame([5], [9], bytes(9), magic=0x46425653), -12),
                      (list_frame([5], [9], bytes(9), version=3), -12),
                      (list_frame([5], [9], bytes(9), variant=2), -12),
                      (list_frame([5], [9], bytes(9), flags=8), -12),
                      (list_frame([1], [9], bytes(9)), -12)):      # more payload than one value can take
        with pytest.raises(svb.SvbError) as e:
            svb.frame_scan_lists(np.frombuffer(bytes(bad), dtype=np.uint8))
        assert f"({code})" in str(e.value)
    # directories that do not add up are refused before any device work (so this runs without a GPU)
    nl, nv = C.c_size_t(0), C.c_size_t(0)
    vals = np.empty(64, dtype=np.uint32)
    ls = np.empty(8, dtype=np.uint64)
    for frame in (list_frame([5, 2], [9, 3], bytes(12), n_values=8),        # counts do not sum to the header's
                  list_frame([5, 2], [9, 3], bytes(13)),                    # sizes do not sum to the payload
                  list_frame([5, 2], [1, 11], bytes(12)),                   # a size below its key bytes
                  list_frame([5, 2], [2, 10], bytes(12))):                  # a size beyond 4 bytes per value
        f = np.frombuffer(frame, dtype=np.uint8)
        assert lib.svbframe_decode_lists(f.ctypes.data, f.size, vals.ctypes.data, vals.size, ls.ctypes.data, 7,
                                         C.addressof(nl), C.addressof(nv)) == -12
    # frames of empty lists decode on the host alone
    f = np.frombuffer(list_frame([0, 0, 0], [0, 0, 0], b""), dtype=np.uint8)
    v, start = svb.frame_decode_lists(f)
    assert v.size == 0 and start.tolist() == [0, 0, 0, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("variant,flags", [(V1234, 0), (V0124, 0), (V1234, 1), (V0124, 2), (V0124, 3)])
def test_list_frames_match_oracle(oracle, variant, flags):
    import streamvbyte_amd as svb
    rng = np.random.default_rng(77 + variant + 4 * flags)
    for n_lists, longest, fv in ((1, 5, 100), (40, 300, 2000), (3000, 700, 50000), (25, 9000, 6000)):
        counts = np.floor(np.exp(rng.uniform(0, np.log(longest + 1), n_lists))).astype(np.int64)
        counts[rng.integers(0, n_lists, max(1, n_lists // 10))] = 0       # empty lists
        if n_lists > 1:
            counts[n_lists // 2] = longest
        start = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        n = int(start[-1])
        v = GENERATORS["width_mix"](rng, max(n, 1))[:n]
        if flags & 1:   # sorted lists with mostly narrow gaps, each starting near zero
            for i in range(n_lists):
                a, b = int(start[i]), int(start[i + 1])
                v[a:b] = np.cumsum(v[a:b] >> 12, dtype=np.uint64).astype(np.uint32)
        frames = svb.frame_encode_lists(v, start, fv, variant=variant, flags=flags)
        expect = np.frombuffer(expected_list_frames(oracle, v, start, fv, variant, flags), dtype=np.uint8)
        assert frames.size == expect.size and np.array_equal(frames, expect), (n_lists, longest, fv)
        nf, nl, nv = svb.frame_scan_lists(frames)
        assert (nl, nv) == (n_lists, n)
        got, got_start = svb.frame_decode_lists(frames)
        assert np.array_equal(got_start, start)
        assert np.array_equal(got, v)


@pytest.mark.gpu
def test_list_frames_reject_bad_input(oracle):
    import streamvbyte_amd as svb
    lib = svb.load()
    rng = np.random.default_rng(9)
    counts = rng.integers(1, 600, 500)
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    v = GENERATORS["width_mix"](rng, int(start[-1]))
    frames = svb.frame_encode_lists(v, start, 40000)
    nf, nl, nv = svb.frame_scan_lists(frames)
    assert nf > 2
    # keys of one list in the second frame promise other lengths than its directory entry
    hdr = struct.unpack_from("<IBBBBIIQQ", frames, 0)
    second = 32 + 2 * ((4 * hdr[5] + 15) // 16 * 16) + (hdr[8] + 15) // 16 * 16
    hdr2 = struct.unpack_from("<IBBBBIIQQ", frames, second)
    payload2 = second + 32 + 2 * ((4 * hdr2[5] + 15) // 16 * 16)
    bad = frames.copy()
    bad[payload2] ^= 0x03
    with pytest.raises(svb.SvbError) as e:
        svb.frame_decode_lists(bad)
    assert "(-14)" in str(e.value)
    with pytest.raises(svb.SvbError) as e:
        svb.frame_decode_lists(frames[: frames.size - 40])
    assert "(-11)" in str(e.value)
    # capacity errors
    n_l, n_v = C.c_size_t(0), C.c_size_t(0)
    out = np.empty(10, dtype=np.uint32)
    ls = np.empty(nl + 1, dtype=np.uint64)
    assert lib.svbframe_decode_lists(frames.ctypes.data, frames.size, out.ctypes.data, out.size, ls.ctypes.data, nl,
                                     C.addressof(n_l), C.addressof(n_v)) == -13
    assert (n_l.value, n_v.value) == (nl, nv)
    small = np.empty(100, dtype=np.uint8)
    w = C.c_size_t(0)
    assert lib.svbframe_encode_lists(v.ctypes.data, start.ctypes.data, 500, 40000, 0, 0, small.ctypes.data, small.size,
                                     C.addressof(w)) == -13
    # list_start must not decrease
    wrong = start.copy()
    wrong[3] = wrong[5]
    wrong[4] = wrong[2]
    big = np.empty(lib.svbframe_lists_max_bytes(v.size, 500, 40000), dtype=np.uint8)
    assert lib.svbframe_encode_lists(v.ctypes.data, wrong.ctypes.data, 500, 40000, 0, 0, big.ctypes.data, big.size,
                                     C.addressof(w)) == -1
    # pinned buffers work the same
    p = lib.svbframe_alloc_pinned(frames.size)
    q = lib.svbframe_alloc_pinned(4 * v.size)
    assert p and q
    C.memmove(p, frames.ctypes.data, frames.size)
    assert lib.svbframe_decode_lists(p, frames.size, q, v.size, ls.ctypes.data, nl, C.addressof(n_l), C.addressof(n_v)) == 0
    got = np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_uint32)), shape=(v.size,)).copy()
    assert np.array_equal(got, v) and np.array_equal(ls, start)
    lib.svbframe_free_pinned(p)
    lib.svbframe_free_pinned(q)
