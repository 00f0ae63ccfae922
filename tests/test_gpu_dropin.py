"""The reference's own unit tests (tests/unit.c of lemire/streamvbyte) re-expressed
against the drop-in host-pointer API of libstreamvbyte.so.2."""
import ctypes

import numpy as np
import pytest

from oracle import V1234, V0124
from util import GENERATORS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def svb():
    """Every call of this module goes to the kernels: the cutoff below which the reference-named
    functions run on the host codec is switched off (tests/test_host_codec.py covers that codec)."""
    import torch
    assert torch.cuda.is_available()
    import streamvbyte_amd
    lib = streamvbyte_amd.load()
    assert lib.svbgpu_set_option(b"host_cpu_cutoff", 0) == 0
    yield streamvbyte_amd
    lib.svbgpu_set_option(b"host_cpu_cutoff", 1 << 16)


def test_small_host_calls_take_the_host_codec_by_default(svb, oracle):
    """With the default cutoff a small host-pointer call is answered without the GPU (same bytes),
    and a device pointer goes to the kernels whatever its size."""
    import time
    import torch
    lib = svb.load()
    v = GENERATORS["width_mix"](np.random.default_rng(2), 1000)
    expect = oracle.encode(V1234, v)
    out = np.zeros(svb.streamvbyte_max_compressedbytes(v.size), dtype=np.uint8)
    try:
        lib.svbgpu_set_option(b"host_cpu_cutoff", 1 << 16)
        lib.streamvbyte_encode(v.ctypes.data, v.size, out.ctypes.data)  # warm
        t0 = time.perf_counter()
        for _ in range(50):
            n = lib.streamvbyte_encode(v.ctypes.data, v.size, out.ctypes.data)
        host_us = (time.perf_counter() - t0) / 50 * 1e6
        assert n == expect.size and np.array_equal(out[:n], expect)
        lib.svbgpu_set_option(b"host_cpu_cutoff", 0)
        lib.streamvbyte_encode(v.ctypes.data, v.size, out.ctypes.data)
        t0 = time.perf_counter()
        for _ in range(50):
            n = lib.streamvbyte_encode(v.ctypes.data, v.size, out.ctypes.data)
        gpu_us = (time.perf_counter() - t0) / 50 * 1e6
        assert n == expect.size and np.array_equal(out[:n], expect)
        assert host_us < gpu_us, (host_us, gpu_us)  # that is what the cutoff is for
        lib.svbgpu_set_option(b"host_cpu_cutoff", 1 << 16)
        d_v = torch.from_numpy(v.view(np.int32)).cuda()
        d_out = torch.zeros(out.size, dtype=torch.uint8, device="cuda:0")
        n = lib.streamvbyte_encode(d_v.data_ptr(), v.size, d_out.data_ptr())
        assert n == expect.size and np.array_equal(d_out[:n].cpu().numpy(), expect)
    finally:
        lib.svbgpu_set_option(b"host_cpu_cutoff", 0)


def test_issue42(svb, golden):
    """reference tests/unit.c:310-2783"""
    arrays, _, stream = golden
    out, used = svb.streamvbyte_decode(stream, 29159)
    assert used == 36494
    assert np.array_equal(out, arrays["issue42_decoded"])
    again = svb.streamvbyte_encode(out)
    assert again.size == 36494 and np.array_equal(again, stream)
    out2, used2 = svb.streamvbyte_decode(again, 29159)
    assert used2 == 36494 and np.array_equal(out2, out)


def test_issue69(svb, golden):
    """reference tests/unit.c:2786-2805"""
    arrays, _, _ = golden
    s = svb.streamvbyte_encode(arrays["issue69_values"])
    assert np.array_equal(s, arrays["issue69_stream"])
    out, used = svb.streamvbyte_decode(s, 22)
    assert used == s.size and np.array_equal(out, arrays["issue69_values"])


def test_zigzag(svb, golden):
    """reference tests/unit.c:22-87"""
    arrays, _, _ = golden
    assert np.array_equal(svb.zigzag_encode(arrays["zigzag_fix_in"]), arrays["zigzag_fix_out"])
    assert np.array_equal(svb.zigzag_decode(arrays["zigzag_fix_out"]), arrays["zigzag_fix_in"])
    rng = np.random.default_rng(5)
    d = (rng.integers(0, 2**31, 4096) - rng.integers(0, 2**31, 4096)).astype(np.int32)
    assert np.array_equal(svb.zigzag_decode(svb.zigzag_encode(d)), d)
    assert np.array_equal(svb.zigzag_delta_decode(svb.zigzag_delta_encode(d, 0), 0), d)


def test_basictests(svb, oracle):
    """reference tests/unit.c:90-185: lengths 0..128 then doubling to 4096,
    gaps 3^k, misaligned compressed buffer, all three codecs."""
    lib = svb.load()
    rng = np.random.default_rng(11)
    lengths = list(range(0, 129, 7)) + [127, 128, 256, 512, 1024, 2048, 4096]
    N = 4096
    raw = np.zeros(svb.streamvbyte_max_compressedbytes(N) + 3, dtype=np.uint8)
    comp = raw[3:]  # misaligned on purpose (tests/unit.c:93-96)
    for length in lengths:
        gap = 1
        while gap <= 387420489:
            d = np.cumsum(gap - 1 + rng.integers(0, 8, size=max(length, 1)), dtype=np.uint64).astype(np.uint32)[:length]
            recov = np.zeros(max(length, 1), dtype=np.uint32)
            for enc, val, dec, variant in (
                    (lib.streamvbyte_encode, lib.streamvbyte_validate_stream, lib.streamvbyte_decode, 0),
                    (lib.streamvbyte_encode_0124, lib.streamvbyte_validate_stream_0124, lib.streamvbyte_decode_0124, 1)):
                size = enc(d.ctypes.data, length, comp.ctypes.data)
                assert size == oracle.compressedbytes(variant, d)
                assert np.array_equal(comp[:size], oracle.encode(variant, d))
                assert val(comp.ctypes.data, size, length)
                used = dec(comp.ctypes.data, recov.ctypes.data, length)
                assert used == size and np.array_equal(recov[:length], d)
            gap *= 9
        gap = 1
        while gap <= 531441:
            d = (gap * np.arange(length, dtype=np.uint64)).astype(np.uint32)
            recov = np.zeros(max(length, 1), dtype=np.uint32)
            size = lib.streamvbyte_delta_encode(d.ctypes.data, length, comp.ctypes.data, 0)
            used = lib.streamvbyte_delta_decode(comp.ctypes.data, recov.ctypes.data, length, 0)
            assert size == used and np.array_equal(recov[:length], d)
            gap *= 9


def test_aqrit_patterns(svb, oracle):
    """reference tests/unit.c:187-252: every 0/1 pattern of 16 bytes as 4 ints
    (as one long stream plus a sample of quads as length-4 streams)."""
    bits = (np.arange(0x10000, dtype=np.uint32)[:, None] >> np.arange(16)) & 1
    allv = bits.astype(np.uint8).view(np.uint32).reshape(-1)
    for enc, dec, val, variant in (
            (svb.streamvbyte_encode, svb.streamvbyte_decode, svb.streamvbyte_validate_stream, 0),
            (svb.streamvbyte_encode_0124, svb.streamvbyte_decode_0124, svb.streamvbyte_validate_stream_0124, 1)):
        s = enc(allv)
        assert np.array_equal(s, oracle.encode(variant, allv))
        assert val(s, allv.size)
        out, used = dec(s, allv.size)
        assert used == s.size and np.array_equal(out, allv)
        for i in range(0, 0x10000, 1021):
            q = allv[4 * i: 4 * i + 4]
            s = enc(q)
            assert np.array_equal(s, oracle.encode(variant, q))
            out, used = dec(s, 4)
            assert used == s.size and np.array_equal(out, q)


def test_compressedbytes_known_answers(svb):
    """reference tests/unit.c:254-308"""
    for vals, a, b in (([1] * 4, 5, 5), ([260] * 4, 9, 9), ([2**32 - 1] * 4, 17, 17), ([0] * 4, 5, 1), ([], 0, 0)):
        v = np.array(vals, dtype=np.uint32)
        assert svb.streamvbyte_compressedbytes(v) == a
        assert svb.streamvbyte_compressedbytes_0124(v) == b


def test_perf_c_shape(svb, golden):
    """reference tests/perf.c:52-53: glibc rand() data, known compressed size"""
    _, scal, _ = golden
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    n = scal["perf_c"]["n"]
    d = np.empty(n, dtype=np.uint32)
    for k in range(n):
        r = libc.rand()
        s = libc.rand()
        d[k] = r >> (31 & s)
    if int(np.bitwise_xor.reduce(d)) != scal["perf_c"]["xor"]:
        pytest.skip("libc rand() sequence differs from the one the fixture was made with")
    s = svb.streamvbyte_encode(d)
    assert s.size == scal["perf_c"]["bytes"]
    out, used = svb.streamvbyte_decode(s, n)
    assert used == s.size and np.array_equal(out, d)


def test_device_pointers_through_reference_names(svb, oracle):
    """The 14 reference symbols also accept hipMalloc'ed pointers."""
    import torch
    lib = svb.load()
    rng = np.random.default_rng(3)
    v = (rng.integers(0, 2**32, 50000, dtype=np.uint64) >> rng.integers(0, 32, 50000).astype(np.uint64)).astype(np.uint32)
    expect = oracle.encode(0, v)
    d_v = torch.from_numpy(v.view(np.int32)).cuda()
    d_out = torch.zeros(svb.streamvbyte_max_compressedbytes(v.size), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    size = lib.streamvbyte_encode(d_v.data_ptr(), v.size, d_out.data_ptr())
    assert size == expect.size and np.array_equal(d_out[:size].cpu().numpy(), expect)
    d_back = torch.zeros_like(d_v)
    used = lib.streamvbyte_decode(d_out.data_ptr(), d_back.data_ptr(), v.size)
    assert used == size and np.array_equal(d_back.cpu().numpy().view(np.uint32), v)


def test_device_pointer_calls_wait_for_the_callers_stream(svb, oracle):
    """The reference-named functions run on a stream of the library's own; with
    device pointers they first wait for whatever the caller queued on the device.
    The input is still being produced (a chain of passes over a large tensor on
    torch's stream) and the output is still being overwritten when the calls are
    made, and no synchronisation is done here."""
    import torch
    lib = svb.load()
    n = 1 << 24
    rng = np.random.default_rng(11)
    v = (rng.integers(0, 2**32, n, dtype=np.uint64) >> rng.integers(0, 32, n).astype(np.uint64)).astype(np.uint32)
    expect = oracle.encode(0, v)
    base = torch.from_numpy(v.view(np.int32)).cuda()
    d_out = torch.empty(svb.streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device="cuda:0")
    d_back = torch.empty_like(base)
    torch.cuda.synchronize()
    d_v = base ^ 0x5A5A5A5A
    for _ in range(20):  # ~20 passes over 64 MiB still queued when the encoder is called
        d_v = d_v ^ 0x0F0F0F0F
        d_v = d_v ^ 0x0F0F0F0F
    d_v = d_v ^ 0x5A5A5A5A
    for _ in range(20):
        d_out.fill_(0xEE)
    size = lib.streamvbyte_encode(d_v.data_ptr(), n, d_out.data_ptr())
    assert size == expect.size
    for _ in range(20):
        d_back.fill_(-1)
    used = lib.streamvbyte_decode(d_out.data_ptr(), d_back.data_ptr(), n)
    assert used == size
    assert np.array_equal(d_out[:size].cpu().numpy(), expect)
    assert np.array_equal(d_back.cpu().numpy().view(np.uint32), v)


def test_concurrent_callers(svb, oracle):
    """The reference API is re-entrant (pure functions of their arguments): eight
    threads calling every host-pointer function at once on their own buffers get
    the answers they would get alone.  ctypes releases the GIL during the calls."""
    import threading
    errors = []

    def worker(t):
        try:
            rng = np.random.default_rng(900 + t)
            for i in range(12):
                n = int(rng.integers(1, 60000))
                v = GENERATORS[["width_mix", "sparse", "postings", "perf_c"][(t + i) % 4]](rng, n)
                e = svb.streamvbyte_encode(v)
                assert np.array_equal(e, oracle.encode(V1234, v)), ("encode", t, i)
                out, used = svb.streamvbyte_decode(e, n)
                assert used == e.size and np.array_equal(out, v), ("decode", t, i)
                assert svb.streamvbyte_validate_stream(e, n) and not svb.streamvbyte_validate_stream(e[:-1], n)
                assert svb.streamvbyte_compressedbytes(v) == e.size
                d = svb.streamvbyte_delta_encode(v, prev=t)
                assert np.array_equal(d, oracle.encode(V1234, v, True, t)), ("delta", t, i)
                assert np.array_equal(svb.streamvbyte_delta_decode(d, n, prev=t)[0], v)
                z = svb.streamvbyte_encode_0124(v)
                assert np.array_equal(z, oracle.encode(V0124, v)) and np.array_equal(svb.streamvbyte_decode_0124(z, n)[0], v)
                s = v.view(np.int32)
                assert np.array_equal(svb.zigzag_delta_decode(svb.zigzag_delta_encode(s, -t), -t), s)
                f = svb.frame_encode_many(v, 7000, flags=svb.DELTA)
                assert np.array_equal(svb.frame_decode_many(f), v), ("frames", t, i)
        except BaseException as exc:  # noqa: BLE001 - reported by the main thread
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]


def test_zigzag_delta_decode_in_pieces(svb, oracle):
    """Arrays longer than one device scan (2^31 values by default) go in pieces seeded with the
    previous piece's last value; with a small piece size the same path runs on a small array."""
    lib = svb.load()
    rng = np.random.default_rng(3)
    v = rng.integers(-(1 << 31), 1 << 31, size=5003, dtype=np.int64).astype(np.int32)
    zz = oracle.zigzag_delta_encode(v, -77)
    assert lib.svbgpu_set_option(b"zigzag_piece", 1000) == 0
    try:
        assert np.array_equal(svb.zigzag_delta_decode(zz, -77), v)
    finally:
        lib.svbgpu_set_option(b"zigzag_piece", 2**31 - 1)
