"""The library's own host codec (csrc/svb_cpu.h): what the 14 reference-named
functions run for small host-pointer calls and in a process without a gfx950
device.  Checked here, on the CPU, against the oracle and the golden vectors;
on a box with a GPU the cutoff is raised so that the same path is taken.
The oracle is the checker only: the library never links or loads it."""
import numpy as np
import pytest

from oracle import V1234, V0124
from util import GENERATORS


@pytest.fixture(scope="module")
def svb():
    import streamvbyte_amd
    lib = streamvbyte_amd.load()
    assert lib.svbgpu_set_option(b"host_cpu_cutoff", 0x7FFFFFFF) == 0  # everything below 2^31 values on the host
    yield streamvbyte_amd
    lib.svbgpu_set_option(b"host_cpu_cutoff", 1 << 16)


SIZES = [0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 4099, 100003]


@pytest.mark.parametrize("kind", ["width_mix", "sparse", "postings", "full", "zeros", "max", "perf_c"])
def test_all_entry_points_against_the_oracle(svb, oracle, kind):
    rng = np.random.default_rng(42)
    for n in SIZES:
        v = GENERATORS[kind](rng, n)
        for variant, enc, dec, size, valid in (
                (V1234, svb.streamvbyte_encode, svb.streamvbyte_decode, svb.streamvbyte_compressedbytes,
                 svb.streamvbyte_validate_stream),
                (V0124, svb.streamvbyte_encode_0124, svb.streamvbyte_decode_0124, svb.streamvbyte_compressedbytes_0124,
                 svb.streamvbyte_validate_stream_0124)):
            expect = oracle.encode(variant, v)
            got = enc(v)
            assert np.array_equal(got, expect), (kind, n, variant)
            back, used = dec(expect, n)
            assert used == expect.size and np.array_equal(back, v), (kind, n, variant)
            assert size(v) == expect.size
            assert valid(expect, n) == oracle.validate(variant, expect, n)
            if expect.size:
                assert valid(expect[:-1], n) == oracle.validate(variant, expect[:-1], n)
                assert valid(expect, n + 1) == oracle.validate(variant, expect, n + 1)
        for prev in (0, 0xDEADBEEF):
            expect = oracle.encode(V1234, v, True, prev)
            assert np.array_equal(svb.streamvbyte_delta_encode(v, prev), expect), (kind, n, prev)
            back, used = svb.streamvbyte_delta_decode(expect, n, prev)
            assert used == expect.size and np.array_equal(back, v), (kind, n, prev)
        s = v.view(np.int32)
        assert np.array_equal(svb.zigzag_encode(s), oracle.zigzag_encode(s))
        assert np.array_equal(svb.zigzag_decode(v), oracle.zigzag_decode(v))
        assert np.array_equal(svb.zigzag_delta_encode(s, -77), oracle.zigzag_delta_encode(s, -77))
        assert np.array_equal(svb.zigzag_delta_decode(v, -77), oracle.zigzag_delta_decode(v, -77))


def test_golden_vectors(svb, golden):
    """reference tests/unit.c:310-2805 (issue42, issue69), :60-87 (zigzag fixtures), :254-308 (sizes),
    README.md:228-242"""
    arrays, scal, stream = golden
    out, used = svb.streamvbyte_decode(stream, 29159)
    assert used == 36494 and np.array_equal(out, arrays["issue42_decoded"])
    assert np.array_equal(svb.streamvbyte_encode(out), stream)
    assert np.array_equal(svb.streamvbyte_encode(arrays["issue69_values"]), arrays["issue69_stream"])
    assert np.array_equal(svb.zigzag_encode(arrays["zigzag_fix_in"]), arrays["zigzag_fix_out"])
    assert np.array_equal(svb.zigzag_decode(arrays["zigzag_fix_out"]), arrays["zigzag_fix_in"])
    assert svb.streamvbyte_compressedbytes(np.array([1, 1, 1, 1], dtype=np.uint32)) == 5
    assert svb.streamvbyte_compressedbytes(np.full(4, 260, dtype=np.uint32)) == 9
    assert svb.streamvbyte_compressedbytes(np.full(4, 0xFFFFFFFF, dtype=np.uint32)) == 17
    assert svb.streamvbyte_compressedbytes_0124(np.zeros(4, dtype=np.uint32)) == 1
    readme = svb.streamvbyte_encode(np.arange(0, 800, 100, dtype=np.uint32))
    assert readme.tobytes().hex() == "405500" + "64c82c019001f4015802bc02"


def test_every_byte_pattern_quad(svb, oracle):
    """reference tests/unit.c:187-252 (aqrittests): all 65 536 combinations of zero / non-zero bytes in four
    values, every one through both variants; here on the host codec, all of them."""
    idx = np.arange(65536, dtype=np.uint32)
    quads = np.zeros((65536, 4), dtype=np.uint32)
    for b in range(16):
        quads[:, b // 4] |= ((idx >> b) & 1) << (8 * (b % 4))
    flat = quads.reshape(-1)
    for variant, enc, dec in ((V1234, svb.streamvbyte_encode, svb.streamvbyte_decode),
                              (V0124, svb.streamvbyte_encode_0124, svb.streamvbyte_decode_0124)):
        # one long stream holds every quad as one group: group boundaries are key boundaries
        expect = oracle.encode(variant, flat)
        assert np.array_equal(enc(flat), expect)
        back, used = dec(expect, flat.size)
        assert used == expect.size and np.array_equal(back, flat)
        for q in quads[::257]:  # and a sample as 4-value streams of their own (the scalar tail path)
            e = enc(q)
            assert np.array_equal(e, oracle.encode(variant, q))
            assert np.array_equal(dec(e, 4)[0], q)


def test_without_a_device_nothing_aborts(svb):
    """A process without a gfx950 device gets answers, not abort(): this test file runs on such a box
    in the CPU tier.  With a device present the same calls are simply below the (raised) cutoff."""
    v = np.arange(100000, dtype=np.uint32) * 2654435761 % (1 << 20)
    v = v.astype(np.uint32)
    e = svb.streamvbyte_encode(v)
    assert svb.streamvbyte_validate_stream(e, v.size)
    assert np.array_equal(svb.streamvbyte_decode(e, v.size)[0], v)
