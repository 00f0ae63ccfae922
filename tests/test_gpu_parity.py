"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the
golden vectors.  Bit-exact for every byte and every integer."""
import numpy as np
import pytest

from oracle import V1234, V0124
from util import GENERATORS, dev_stream, to_dev, to_np

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097,
         16383, 16384, 16385, 65535, 65537, 100003, 262144 + 7, 1 << 20]


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def codec(torch_mod):
    import streamvbyte_amd as svb
    return svb.DeviceCodec(max_count=1 << 25)


def gen(kind, n, seed):
    return GENERATORS[kind](np.random.default_rng(seed), n)


@pytest.mark.parametrize("variant", [V1234, V0124])
@pytest.mark.parametrize("kind", ["width_mix", "sparse", "zeros", "max", "perf_c"])
def test_decode_matches_oracle(torch_mod, codec, oracle, variant, kind):
    for i, n in enumerate(SIZES):
        v = gen(kind, n, 1000 + i)
        s = oracle.encode(variant, v)
        for misalign in ((0,) if i % 3 else (0, 3)):
            d_in = dev_stream(torch_mod, s, misalign=misalign)
            out, used = codec.decode(d_in, n, variant=variant)
            assert int(used.item()) == s.size, (kind, n, misalign)
            got = to_np(out, np.uint32)
            assert np.array_equal(got, v), (kind, n, misalign, np.flatnonzero(got != v)[:5])


@pytest.mark.parametrize("variant", [V1234, V0124])
@pytest.mark.parametrize("kind", ["width_mix", "sparse", "zeros", "max", "perf_c"])
def test_encode_matches_oracle(torch_mod, codec, oracle, variant, kind):
    for i, n in enumerate(SIZES):
        v = gen(kind, n, 2000 + i)
        expect = oracle.encode(variant, v)
        d_v = to_dev(torch_mod, v)
        buf = torch_mod.full((3 + expect.size + 64 + 4 * n // 2 + 16,), 0xAB, dtype=torch_mod.uint8, device="cuda:0")
        import streamvbyte_amd as svb
        buf = torch_mod.full((3 + svb.streamvbyte_max_compressedbytes(n),), 0xAB, dtype=torch_mod.uint8, device="cuda:0")
        mis = 3 if i % 2 else 0
        out, size = codec.encode(d_v, out=buf[mis:], variant=variant)
        codec.check()
        assert int(size.item()) == expect.size, (kind, n)
        got = buf[mis: mis + expect.size].cpu().numpy()
        assert np.array_equal(got, expect), (kind, n, np.flatnonzero(got != expect)[:5])
        if mis:
            assert (buf[:mis].cpu().numpy() == 0xAB).all()  # nothing written in front of out
        assert (buf[mis + expect.size:].cpu().numpy() == 0xAB).all(), (kind, n)  # nor behind the stream


@pytest.mark.parametrize("kind", ["postings", "full", "width_mix"])
def test_delta_matches_oracle(torch_mod, codec, oracle, kind):
    from streamvbyte_amd import DELTA
    for i, n in enumerate(SIZES):
        v = gen(kind, n, 3000 + i)
        prev = [0, 7, 0xDEADBEEF][i % 3]
        s = oracle.encode(V1234, v, True, prev)
        out, size = codec.encode(to_dev(torch_mod, v), variant=V1234, flags=DELTA, prev=prev)
        codec.check()
        assert int(size.item()) == s.size
        assert np.array_equal(out[: s.size].cpu().numpy(), s), (kind, n)
        for lookback, waves in ((0, 16), (1, 16), (1, 8), (1, 4)):  # two-pass and single-pass look-back decoders
            codec.lib.svbgpu_set_option(b"delta_decode_lookback", lookback)
            codec.lib.svbgpu_set_option(b"scan_waves", waves)
            dec, used = codec.decode(dev_stream(torch_mod, s), n, variant=V1234, flags=DELTA, prev=prev)
            codec.check()
            codec.lib.svbgpu_set_option(b"delta_decode_lookback", 1)
            codec.lib.svbgpu_set_option(b"scan_waves", 16)
            assert int(used.item()) == s.size
            assert np.array_equal(to_np(dec, np.uint32), v), (kind, n, lookback)


def test_zigzag_fused_and_plain(torch_mod, codec, oracle):
    from streamvbyte_amd import ZIGZAG, DELTA
    for i, n in enumerate(SIZES):
        g = gen("sparse", n, 4000 + i).view(np.int32)
        d_g = to_dev(torch_mod, g)
        zz = oracle.zigzag_encode(g)
        assert np.array_equal(to_np(codec.zigzag(d_g, "encode"), np.uint32), zz)
        assert np.array_equal(to_np(codec.zigzag(to_dev(torch_mod, zz), "decode"), np.int32), g)
        prev = [0, -5, 2**31 - 1][i % 3]
        zzd = oracle.zigzag_delta_encode(g, prev)
        assert np.array_equal(to_np(codec.zigzag(d_g, "delta_encode", prev), np.uint32), zzd)
        back = codec.zigzag(to_dev(torch_mod, zzd), "delta_decode", prev)
        codec.check()
        assert np.array_equal(to_np(back, np.int32), g)
        # fused: zigzag inside the 0124 codec pass == zigzag_encode then encode_0124
        expect = oracle.encode(V0124, zz)
        out, size = codec.encode(d_g, variant=V0124, flags=ZIGZAG)
        assert int(size.item()) == expect.size
        assert np.array_equal(out[: expect.size].cpu().numpy(), expect)
        dec, used = codec.decode(dev_stream(torch_mod, expect), n, variant=V0124, flags=ZIGZAG)
        assert int(used.item()) == expect.size
        assert np.array_equal(to_np(dec, np.int32), g)
        # fused signed delta: zigzag_delta_encode then encode
        expect = oracle.encode(V1234, zzd)
        out, size = codec.encode(d_g, variant=V1234, flags=ZIGZAG | DELTA, prev=prev)
        codec.check()
        assert int(size.item()) == expect.size
        assert np.array_equal(out[: expect.size].cpu().numpy(), expect)
        dec, used = codec.decode(dev_stream(torch_mod, expect), n, variant=V1234, flags=ZIGZAG | DELTA, prev=prev)
        codec.check()
        assert np.array_equal(to_np(dec, np.int32), g)


def test_encoder_tile_sizes(torch_mod, codec, oracle):
    """Every tile shape of the tile-at-a-time encoder (wave tiles per wave, waves per workgroup, a workgroup per
    tile or persistent workgroups) gives the same bytes."""
    from streamvbyte_amd import DELTA
    opt = codec.lib.svbgpu_set_option
    try:
        assert opt(b"encode_pipeline", 0) == 0
        for i, n in enumerate([1, 4097, 16385, 32769, 65537, 262144 + 7, (1 << 20) + 3]):
            v = gen("width_mix", n, 6000 + i)
            for variant, flags, prev in ((V1234, 0, 0), (V0124, 0, 0), (V1234, DELTA, 99)):
                expect = oracle.encode(variant, v, bool(flags), prev)
                for qw, waves, fresh in ((1, 8, 1), (2, 8, 1), (4, 4, 1), (4, 8, 1), (4, 16, 1), (4, 4, 0), (4, 8, 0), (4, 16, 0)):
                    opt(b"encode_tile", qw)
                    opt(b"encode_waves", waves)
                    opt(b"encode_fresh", fresh)
                    out, size = codec.encode(to_dev(torch_mod, v), variant=variant, flags=flags, prev=prev)
                    codec.check()
                    assert int(size.item()) == expect.size, (n, variant, flags, qw, waves, fresh)
                    assert np.array_equal(out[: expect.size].cpu().numpy(), expect), (n, variant, flags, qw, waves, fresh)
    finally:
        opt(b"encode_tile", 4)
        opt(b"encode_waves", 8)
        opt(b"encode_fresh", 1)
        opt(b"encode_pipeline", 3)


def test_encoder_chosen_by_size(torch_mod, oracle):
    """The default picks an encoder by the stream's size: one size from each band of the rule (pipelined, 8 waves
    with a workgroup per tile, 16 waves, pipelined again for plain input), plain and differential, against the oracle."""
    import streamvbyte_amd as svb
    from streamvbyte_amd import DELTA
    n_max = (1 << 25) + 4099
    codec = svb.DeviceCodec(max_count=n_max)
    for n in ((1 << 23) - 5, (1 << 23) + 4097, (1 << 24) + 33, n_max):
        v = gen("width_mix", n, 6600 + (n & 0xFF))
        d_v = to_dev(torch_mod, v)
        for variant, flags, prev in ((V1234, 0, 0), (V0124, 0, 0), (V1234, DELTA, 99)):
            expect = oracle.encode(variant, v, bool(flags), prev)
            out, size = codec.encode(d_v, variant=variant, flags=flags, prev=prev)
            codec.check()
            assert int(size.item()) == expect.size, (n, variant, flags)
            assert np.array_equal(out[: expect.size].cpu().numpy(), expect), (n, variant, flags)


def test_cache_policies_and_early_ticket(torch_mod, codec, oracle):
    """Non-temporal loads / stores (forced on and off, whatever the size) and the encoder's early ticket
    change no byte and no integer."""
    from streamvbyte_amd import DELTA, ZIGZAG
    lib = codec.lib
    try:
        for i, n in enumerate([1, 16385, 262144 + 7, (1 << 21) + 3]):
            v = gen("width_mix", n, 6500 + i)
            d_v = to_dev(torch_mod, v)
            for nt in (0, 1, 2, 3):
                assert lib.svbgpu_set_option(b"nt_force", nt) == 0
                for early in (0, 1):
                    assert lib.svbgpu_set_option(b"encode_ticket_early", early) == 0
                    for variant, flags, prev in ((V1234, 0, 0), (V0124, ZIGZAG, 0), (V1234, DELTA, 99), (V0124, 0, 0)):
                        src = oracle.zigzag_encode(v.view(np.int32)) if flags & ZIGZAG else v
                        expect = oracle.encode(variant, src, bool(flags & DELTA), prev)
                        out, size = codec.encode(d_v, variant=variant, flags=flags, prev=prev)
                        codec.check()
                        assert int(size.item()) == expect.size, (n, nt, early, variant, flags)
                        assert np.array_equal(out[: expect.size].cpu().numpy(), expect), (n, nt, early, variant, flags)
                        if early == 0:
                            dec, used = codec.decode(out, n, variant=variant, flags=flags, prev=prev)
                            codec.check()
                            assert int(used.item()) == expect.size
                            assert np.array_equal(to_np(dec, np.uint32), v), (n, nt, variant, flags)
                assert int(codec.compressedbytes(d_v).item()) == oracle.encode(V1234, v).size
                zz = codec.zigzag(d_v, "encode")
                assert np.array_equal(to_np(zz, np.uint32), oracle.zigzag_encode(v.view(np.int32)))
                back = codec.zigzag(zz, "delta_decode", 5)
                assert np.array_equal(to_np(back, np.int32), oracle.zigzag_delta_decode(to_np(zz, np.uint32), 5))
    finally:
        lib.svbgpu_set_option(b"nt_force", -1)
        lib.svbgpu_set_option(b"encode_ticket_early", -1)


def test_differential_encode_through_every_encoder(torch_mod, codec, oracle):
    """encode_pipeline 0 / 1 / 2: differential input through the tile-at-a-time kernel and through the
    pipelined one gives the oracle's bytes (dense gaps, mixed gaps, zigzag, both variants, prev != 0)."""
    from streamvbyte_amd import DELTA, ZIGZAG
    lib = codec.lib
    try:
        for i, n in enumerate([5, 16385, 65536 + 3, (1 << 20) + 7, (1 << 22) + 1]):
            dense = np.cumsum(np.random.default_rng(7100 + i).integers(0, 32, n, dtype=np.uint64)).astype(np.uint32)
            for v in (dense, gen("width_mix", n, 7200 + i)):
                d_v = to_dev(torch_mod, v)
                for mode in (0, 1, 2):
                    assert lib.svbgpu_set_option(b"encode_pipeline", mode) == 0
                    for variant, flags, prev in ((V1234, DELTA, 77), (V0124, DELTA, 0), (V1234, DELTA | ZIGZAG, 5)):
                        src = oracle.zigzag_delta_encode(v.view(np.int32), prev) if flags & ZIGZAG else v
                        expect = oracle.encode(variant, src, not (flags & ZIGZAG), prev)
                        out, size = codec.encode(d_v, variant=variant, flags=flags, prev=prev)
                        codec.check()
                        assert int(size.item()) == expect.size, (n, mode, variant, flags)
                        assert np.array_equal(out[: expect.size].cpu().numpy(), expect), (n, mode, variant, flags)
    finally:
        lib.svbgpu_set_option(b"encode_pipeline", 3)


def test_size_and_validate(torch_mod, codec, oracle):
    for i, n in enumerate(SIZES):
        v = gen("perf_c", n, 5000 + i)
        d_v = to_dev(torch_mod, v)
        for variant in (V1234, V0124):
            s = oracle.encode(variant, v)
            assert int(codec.compressedbytes(d_v, variant=variant).item()) == s.size
            d_s = dev_stream(torch_mod, s)
            for inlen, cnt in ((s.size, n), (s.size + 1, n), (s.size - 1, n), (s.size, n + 1), (s.size, n - 1),
                               (0, n), (s.size, 0), (0, 0), ((n + 3) // 4 - 1, n)):
                if inlen < 0 or cnt < 0:
                    continue
                expect = oracle.validate(variant, s, cnt, inlen) if (cnt + 3) // 4 <= s.size + 16 else False
                got = bool(codec.validate(d_s, inlen, cnt, variant=variant).item() & 0xFFFFFFFF)
                assert got == expect, (n, variant, inlen, cnt)


def test_golden_vectors(torch_mod, codec, golden):
    """issue42 / issue69 / README vector (reference tests/unit.c:310-2805, README.md:228-242)
    and the reference-generated seeded cases, straight from tests/golden/."""
    from streamvbyte_amd import DELTA
    arrays, scal, issue42 = golden
    out, used = codec.decode(dev_stream(torch_mod, issue42), 29159)
    assert int(used.item()) == 36494
    assert np.array_equal(to_np(out, np.uint32), arrays["issue42_decoded"])
    enc, size = codec.encode(out)
    codec.check()
    assert int(size.item()) == 36494 and np.array_equal(enc[:36494].cpu().numpy(), issue42)
    for vals, stream in ((arrays["issue69_values"], arrays["issue69_stream"]),
                         (arrays["readme_values"], arrays["readme_stream"])):
        enc, size = codec.encode(to_dev(torch_mod, vals))
        assert np.array_equal(enc[: int(size.item())].cpu().numpy(), stream)
    for c in scal["cases"]:
        i, n, prev = c["id"], c["n"], c["prev"]
        if n == 0:
            continue
        v = arrays[f"case{i}_values"]
        d_v = to_dev(torch_mod, v)
        for variant, key, flags, p in ((V1234, "s1234", 0, 0), (V0124, "s0124", 0, 0), (V1234, "sdelta", DELTA, prev)):
            s = arrays[f"case{i}_{key}"]
            enc, size = codec.encode(d_v, variant=variant, flags=flags, prev=p)
            codec.check()
            assert int(size.item()) == s.size and np.array_equal(enc[: s.size].cpu().numpy(), s), (i, key)
            dec, used = codec.decode(dev_stream(torch_mod, s), n, variant=variant, flags=flags, prev=p)
            codec.check()
            assert int(used.item()) == s.size and np.array_equal(to_np(dec, np.uint32), v), (i, key)


def test_batched_streams(torch_mod, codec, oracle):
    from streamvbyte_amd import DELTA, streamvbyte_max_compressedbytes
    torch = torch_mod
    rng = np.random.default_rng(99)
    ns = 3000
    counts = np.floor(np.exp(rng.uniform(0, np.log(4096), size=ns))).astype(np.uint32)
    counts[:8] = [0, 1, 3, 4, 5, 1024, 1025, 5000]
    prevs = rng.integers(0, 2**32, size=ns, dtype=np.uint64).astype(np.uint32)
    voff = np.concatenate([[0], np.cumsum(counts.astype(np.uint64))]).astype(np.uint64)
    cap = np.array([streamvbyte_max_compressedbytes(int(c)) for c in counts], dtype=np.uint64)
    boff = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
    values = np.concatenate([np.cumsum(rng.integers(0, 300, size=int(c)), dtype=np.uint64).astype(np.uint32)
                             if k % 2 else GENERATORS["perf_c"](rng, int(c)) for k, c in enumerate(counts)])
    d_values = to_dev(torch, values)
    d_voff = torch.from_numpy(voff[:-1].astype(np.int64)).cuda()
    d_boff = torch.from_numpy(boff[:-1].astype(np.int64)).cuda()
    d_counts = to_dev(torch, counts)
    d_prevs = to_dev(torch, prevs)
    for variant, flags in ((V1234, 0), (V0124, 0), (V1234, DELTA)):
        buf = torch.zeros(int(boff[-1]) + 16, dtype=torch.uint8, device="cuda:0")
        sizes = codec.encode_batch(d_values, d_voff, buf, d_boff, d_counts, d_prevs, variant=variant, flags=flags)
        h_sizes = sizes.cpu().numpy()
        h_buf = buf.cpu().numpy()
        for s in range(ns):
            v = values[int(voff[s]): int(voff[s + 1])]
            expect = oracle.encode(variant, v, bool(flags & DELTA), int(prevs[s]))
            assert h_sizes[s] == expect.size, (s, counts[s])
            assert np.array_equal(h_buf[int(boff[s]): int(boff[s]) + expect.size], expect), (s, counts[s])
        # the size pre-pass of the batch (streamvbyte_compressedbytes stream by stream) agrees without encoding
        cb = codec.compressedbytes_batch(d_values, d_voff, d_counts, d_prevs, variant=variant, flags=flags)
        assert np.array_equal(cb.cpu().numpy(), h_sizes)
        back = torch.zeros_like(d_values)
        used = codec.decode_batch(buf, d_boff, back, d_voff, d_counts, d_prevs, variant=variant, flags=flags)
        assert np.array_equal(used.cpu().numpy(), h_sizes)
        assert np.array_equal(to_np(back, np.uint32), values)


def _expected_stream(oracle, variant, flags, v_u32, prev):
    """What the reference's functions chained on the CPU give for this flag set:
    [zigzag_delta_encode | zigzag_encode | delta] then encode / encode_0124."""
    from streamvbyte_amd import DELTA, ZIGZAG
    if flags & ZIGZAG:
        signed = v_u32.view(np.int32)
        sprev = prev - (1 << 32) if prev >= (1 << 31) else prev
        pre = oracle.zigzag_delta_encode(signed, sprev) if flags & DELTA else oracle.zigzag_encode(signed)
        return oracle.encode(variant, pre)
    return oracle.encode(variant, v_u32, bool(flags & DELTA), prev)


@pytest.mark.parametrize("variant", [V1234, V0124])
@pytest.mark.parametrize("flagbits", [0, 1, 2, 3])
def test_every_variant_and_flag_combination(torch_mod, codec, oracle, variant, flagbits):
    """All 8 combinations of {1234, 0124} x {plain, delta, zigzag, zigzag+delta}: the fused
    kernels equal the reference's functions applied one after the other."""
    from streamvbyte_amd import DELTA, ZIGZAG
    flags = (DELTA if flagbits & 1 else 0) | (ZIGZAG if flagbits & 2 else 0)
    for i, n in enumerate([1, 4, 5, 255, 1024, 1025, 4097, 16384, 16385, 70001, 300007]):
        v = gen(["sparse", "width_mix", "postings"][i % 3], n, 7000 + 16 * flagbits + i)
        prev = [0, 0xFFFFFFF0, 12345][i % 3] if flags & DELTA else 0
        expect = _expected_stream(oracle, variant, flags, v, prev)
        out, size = codec.encode(to_dev(torch_mod, v), variant=variant, flags=flags, prev=prev)
        codec.check()
        assert int(size.item()) == expect.size, (n,)
        assert np.array_equal(out[: expect.size].cpu().numpy(), expect), (n,)
        assert int(codec.compressedbytes(to_dev(torch_mod, v), variant=variant, flags=flags, prev=prev).item()) == expect.size
        dec, used = codec.decode(dev_stream(torch_mod, expect), n, variant=variant, flags=flags, prev=prev)
        codec.check()
        assert int(used.item()) == expect.size
        assert np.array_equal(to_np(dec, np.uint32), v), (n,)


@pytest.mark.parametrize("variant", [V1234, V0124])
@pytest.mark.parametrize("flagbits", [1, 2, 3])
def test_batched_flag_combinations(torch_mod, codec, oracle, variant, flagbits):
    from streamvbyte_amd import DELTA, ZIGZAG, streamvbyte_max_compressedbytes
    torch = torch_mod
    flags = (DELTA if flagbits & 1 else 0) | (ZIGZAG if flagbits & 2 else 0)
    rng = np.random.default_rng(500 + flagbits)
    ns = 400
    counts = np.floor(np.exp(rng.uniform(0, np.log(3000), size=ns))).astype(np.uint32)
    counts[:6] = [0, 1, 255, 256, 257, 2049]
    prevs = rng.integers(0, 2**32, size=ns, dtype=np.uint64).astype(np.uint32)
    voff = np.concatenate([[0], np.cumsum(counts.astype(np.uint64))]).astype(np.uint64)
    cap = np.array([streamvbyte_max_compressedbytes(int(c)) for c in counts], dtype=np.uint64)
    boff = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
    values = np.concatenate([GENERATORS["sparse" if k % 2 else "width_mix"](rng, int(c)) for k, c in enumerate(counts)])
    d_values = to_dev(torch, values)
    d_voff = torch.from_numpy(voff[:-1].astype(np.int64)).cuda()
    d_boff = torch.from_numpy(boff[:-1].astype(np.int64)).cuda()
    d_counts, d_prevs = to_dev(torch, counts), to_dev(torch, prevs)
    buf = torch.zeros(int(boff[-1]) + 16, dtype=torch.uint8, device="cuda:0")
    sizes = codec.encode_batch(d_values, d_voff, buf, d_boff, d_counts, d_prevs, variant=variant, flags=flags)
    h_sizes, h_buf = sizes.cpu().numpy(), buf.cpu().numpy()
    for s in range(ns):
        v = values[int(voff[s]): int(voff[s + 1])]
        expect = _expected_stream(oracle, variant, flags, v, int(prevs[s]) if flags & DELTA else 0)
        assert h_sizes[s] == expect.size, (s, counts[s])
        assert np.array_equal(h_buf[int(boff[s]): int(boff[s]) + expect.size], expect), (s, counts[s])
    cb = codec.compressedbytes_batch(d_values, d_voff, d_counts, d_prevs, variant=variant, flags=flags)
    assert np.array_equal(cb.cpu().numpy(), h_sizes)
    cb = codec.compressedbytes_batch(d_values, d_voff, d_counts, None, variant=variant, flags=flags)   # seeds default to 0
    if flags & DELTA:
        for s in range(0, ns, 7):
            v = values[int(voff[s]): int(voff[s + 1])]
            assert int(cb[s].item()) == _expected_stream(oracle, variant, flags, v, 0).size, (s,)
    else:
        assert np.array_equal(cb.cpu().numpy(), h_sizes)
    back = torch.zeros_like(d_values)
    used = codec.decode_batch(buf, d_boff, back, d_voff, d_counts, d_prevs, variant=variant, flags=flags)
    assert np.array_equal(used.cpu().numpy(), h_sizes)
    assert np.array_equal(to_np(back, np.uint32), values)


def test_batched_cache_policies(torch_mod, codec, oracle):
    """Batched calls under every forced cache policy, and with the choice left to the device at a
    threshold (0 and 1 MiB) that this batch crosses: the same bytes and integers."""
    from streamvbyte_amd import DELTA, ZIGZAG, streamvbyte_max_compressedbytes
    torch = torch_mod
    lib = codec.lib
    rng = np.random.default_rng(4242)
    ns = 1500
    counts = np.floor(np.exp(rng.uniform(0, np.log(4096), size=ns))).astype(np.uint32)
    counts[:7] = [0, 1, 256, 257, 1024, 1025, 9000]
    counts[-1] = 0  # the device's guess reads the last directory entry
    prevs = rng.integers(0, 2**32, size=ns, dtype=np.uint64).astype(np.uint32)
    voff = np.concatenate([[0], np.cumsum(counts.astype(np.uint64))]).astype(np.uint64)
    cap = np.array([streamvbyte_max_compressedbytes(int(c)) for c in counts], dtype=np.uint64)
    boff = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
    values = np.concatenate([GENERATORS["postings" if k % 2 else "width_mix"](rng, int(c)) for k, c in enumerate(counts)])
    assert values.size * 4 > (2 << 20)
    d_values = to_dev(torch, values)
    d_voff = torch.from_numpy(voff[:-1].astype(np.int64)).cuda()
    d_boff = torch.from_numpy(boff[:-1].astype(np.int64)).cuda()
    d_counts, d_prevs = to_dev(torch, counts), to_dev(torch, prevs)
    try:
        for variant, flags in ((V1234, DELTA), (V0124, ZIGZAG), (V1234, 0)):
            expect = [_expected_stream(oracle, variant, flags, values[int(voff[s]): int(voff[s + 1])],
                                       int(prevs[s]) if flags & DELTA else 0) for s in range(ns)]
            for force, mib in ((0, 128), (1, 128), (2, 128), (3, 128), (-1, 0), (-1, 1), (-1, 128)):
                assert lib.svbgpu_set_option(b"nt_force", force) == 0
                assert lib.svbgpu_set_option(b"nt_threshold_mib", mib) == 0
                buf = torch.zeros(int(boff[-1]) + 16, dtype=torch.uint8, device="cuda:0")
                sizes = codec.encode_batch(d_values, d_voff, buf, d_boff, d_counts, d_prevs, variant=variant, flags=flags)
                h_sizes, h_buf = sizes.cpu().numpy(), buf.cpu().numpy()
                for s in range(ns):
                    assert h_sizes[s] == expect[s].size, (s, counts[s], force, mib)
                    assert np.array_equal(h_buf[int(boff[s]): int(boff[s]) + expect[s].size], expect[s]), (s, counts[s], force, mib)
                back = torch.zeros_like(d_values)
                used = codec.decode_batch(buf, d_boff, back, d_voff, d_counts, d_prevs, variant=variant, flags=flags)
                assert np.array_equal(used.cpu().numpy(), h_sizes), (force, mib)
                assert np.array_equal(to_np(back, np.uint32), values), (force, mib)
    finally:
        lib.svbgpu_set_option(b"nt_force", -1)
        lib.svbgpu_set_option(b"nt_threshold_mib", 128)


def test_batched_decode_any_start_alignment(torch_mod, codec, oracle):
    """Lists whose values start at every distance from a 64-byte boundary, with gaps of sentinel values between
    them: the decoder (which stores the whole tiles of a long list as aligned quarters, displaced through LDS, and
    the values that hang over one tile later) writes every value of every list and nothing else.  Lengths
    around the tile borders, exact multiples of a tile included."""
    from streamvbyte_amd import DELTA, ZIGZAG, streamvbyte_max_compressedbytes
    torch = torch_mod
    rng = np.random.default_rng(9091)
    lens = [1024, 2048, 1025, 3077, 257, 1023, 4096, 1, 300, 2047, 5120, 1039]
    counts, voff, pos = [], [], 0
    for a in range(16):
        for k, n in enumerate(lens):
            gap = 16 + (a - pos) % 16 if k == 0 else int(rng.integers(0, 3)) * 16 + int(rng.integers(1, 4)) * (k % 2)
            pos += gap
            if k == 0:
                assert pos % 16 == a
            counts.append(n)
            voff.append(pos)
            pos += n
    total = pos + 64
    counts = np.array(counts, dtype=np.uint32)
    voff = np.array(voff, dtype=np.uint64)
    ns = counts.size
    assert {int(v) % 16 for v, c in zip(voff, counts) if c >= 1024} == set(range(16))
    prevs = rng.integers(0, 2**32, size=ns, dtype=np.uint64).astype(np.uint32)
    cap = np.array([streamvbyte_max_compressedbytes(int(c)) for c in counts], dtype=np.uint64)
    boff = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
    SENT = np.uint32(0xA5A5A5A5)
    values = np.full(total, SENT, dtype=np.uint32)
    for s in range(ns):
        values[int(voff[s]): int(voff[s]) + int(counts[s])] = GENERATORS["postings" if s % 3 else "width_mix"](rng, int(counts[s]))
    d_values = to_dev(torch, values)
    d_voff = torch.from_numpy(voff.astype(np.int64)).cuda()
    d_boff = torch.from_numpy(boff[:-1].astype(np.int64)).cuda()
    d_counts, d_prevs = to_dev(torch, counts), to_dev(torch, prevs)
    for variant, flags in ((V1234, DELTA), (V1234, 0), (V0124, ZIGZAG | DELTA)):
        buf = torch.zeros(int(boff[-1]) + 16, dtype=torch.uint8, device="cuda:0")
        sizes = codec.encode_batch(d_values, d_voff, buf, d_boff, d_counts, d_prevs, variant=variant, flags=flags)
        h_sizes, h_buf = sizes.cpu().numpy(), buf.cpu().numpy()
        for s in range(ns):
            e = _expected_stream(oracle, variant, flags, values[int(voff[s]): int(voff[s]) + int(counts[s])],
                                 int(prevs[s]) if flags & DELTA else 0)
            assert h_sizes[s] == e.size and np.array_equal(h_buf[int(boff[s]): int(boff[s]) + e.size], e), (s, counts[s])
        back = torch.full((total,), int(SENT) - (1 << 32), dtype=torch.int32, device="cuda:0")
        used = codec.decode_batch(buf, d_boff, back, d_voff, d_counts, d_prevs, variant=variant, flags=flags)
        assert np.array_equal(used.cpu().numpy(), h_sizes)
        assert np.array_equal(to_np(back, np.uint32), values), (variant, flags)  # the lists AND the gaps between them


def test_batched_directory_in_any_order(torch_mod, codec, oracle):
    """The directory of a batch need not follow the memory layout: streams listed in reverse and in
    shuffled order (the device's guess at the batch's size, taken from the first and last entry, is then
    wrong or even 'negative') give the same bytes per stream; a batch of one stream too."""
    from streamvbyte_amd import DELTA, streamvbyte_max_compressedbytes
    torch = torch_mod
    lib = codec.lib
    rng = np.random.default_rng(777)
    ns = 600
    counts = np.floor(np.exp(rng.uniform(0, np.log(4096), size=ns))).astype(np.uint32)
    prevs = rng.integers(0, 2**32, size=ns, dtype=np.uint64).astype(np.uint32)
    voff = np.concatenate([[0], np.cumsum(counts.astype(np.uint64))]).astype(np.uint64)
    cap = np.array([streamvbyte_max_compressedbytes(int(c)) for c in counts], dtype=np.uint64)
    boff = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
    values = np.concatenate([GENERATORS["postings"](rng, int(c)) for c in counts])
    expect = [oracle.encode(V1234, values[int(voff[s]): int(voff[s + 1])], True, int(prevs[s])) for s in range(ns)]
    d_values = to_dev(torch, values)
    try:
        assert lib.svbgpu_set_option(b"nt_threshold_mib", 0) == 0  # the guess decides something at this size
        for order in (np.arange(ns)[::-1].copy(), rng.permutation(ns), np.array([ns // 2])):
            d_voff = torch.from_numpy(voff[:-1][order].astype(np.int64)).cuda()
            d_boff = torch.from_numpy(boff[:-1][order].astype(np.int64)).cuda()
            d_counts, d_prevs = to_dev(torch, counts[order]), to_dev(torch, prevs[order])
            buf = torch.zeros(int(boff[-1]) + 16, dtype=torch.uint8, device="cuda:0")
            sizes = codec.encode_batch(d_values, d_voff, buf, d_boff, d_counts, d_prevs, variant=V1234, flags=DELTA)
            h_sizes, h_buf = sizes.cpu().numpy(), buf.cpu().numpy()
            for k, s in enumerate(order):
                assert h_sizes[k] == expect[s].size, (k, s)
                assert np.array_equal(h_buf[int(boff[s]): int(boff[s]) + expect[s].size], expect[s]), (k, s)
            back = torch.zeros_like(d_values)
            used = codec.decode_batch(buf, d_boff, back, d_voff, d_counts, d_prevs, variant=V1234, flags=DELTA)
            assert np.array_equal(used.cpu().numpy(), h_sizes)
            h_back = to_np(back, np.uint32)
            for s in order:
                assert np.array_equal(h_back[int(voff[s]): int(voff[s + 1])], values[int(voff[s]): int(voff[s + 1])]), s
    finally:
        lib.svbgpu_set_option(b"nt_threshold_mib", 128)


def test_calls_capture_into_a_hip_graph(torch_mod, codec, oracle):
    """The device API allocates nothing and never synchronises, so a sequence of
    calls records into a HIP graph and replays on new data in the same buffers."""
    from streamvbyte_amd import DELTA
    torch = torch_mod
    n = 50001
    v0 = gen("width_mix", n, 1)
    d_v = to_dev(torch, v0)
    buf = torch.zeros(5 * n + 64, dtype=torch.uint8, device="cuda:0")
    back = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            _, size = codec.encode(d_v, out=buf, flags=DELTA, prev=5)
            ok = codec.validate(buf, 0, n)  # in_length is an argument, not device data: a wrong one must say 0
            _, used = codec.decode(buf, n, out=back, flags=DELTA, prev=5)
    for seed in (2, 3):
        v = gen("width_mix", n, seed)
        d_v.copy_(to_dev(torch, v))
        graph.replay()
        torch.cuda.synchronize()
        expect = oracle.encode(V1234, v, True, 5)
        assert int(size.item()) == expect.size and int(used.item()) == expect.size
        assert np.array_equal(buf[: expect.size].cpu().numpy(), expect)
        assert np.array_equal(to_np(back, np.uint32), v)
        assert int(ok.item()) == 0
    codec.check()


def test_device_api_errors_and_empty_inputs(torch_mod, codec):
    """Error codes instead of faults: bad workspace, bad arguments; count == 0 everywhere."""
    torch = torch_mod
    lib = codec.lib
    st = torch.cuda.current_stream().cuda_stream
    v = torch.arange(5000, dtype=torch.int32, device="cuda:0")
    buf = torch.zeros(5 * 5000 + 64, dtype=torch.uint8, device="cuda:0")
    res = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    ws, wsb = codec.ws.data_ptr(), codec.ws_bytes
    need = lib.svbgpu_workspace_bytes(5000)
    # workspace: missing, misaligned, too small
    assert lib.svbgpu_encode(0, 0, v.data_ptr(), 5000, buf.data_ptr(), 0, None, 0, res.data_ptr(), st) == -3
    assert lib.svbgpu_encode(0, 0, v.data_ptr(), 5000, buf.data_ptr(), 0, ws + 8, wsb - 8, res.data_ptr(), st) == -3
    assert lib.svbgpu_encode(0, 0, v.data_ptr(), 5000, buf.data_ptr(), 0, ws, need - 256, res.data_ptr(), st) == -3
    assert lib.svbgpu_decode(0, 0, buf.data_ptr(), v.data_ptr(), 5000, 0, ws, need - 256, res.data_ptr(), st) == -3
    # arguments
    assert lib.svbgpu_encode(2, 0, v.data_ptr(), 5000, buf.data_ptr(), 0, ws, wsb, res.data_ptr(), st) == -1
    assert lib.svbgpu_encode(0, 4, v.data_ptr(), 5000, buf.data_ptr(), 0, ws, wsb, res.data_ptr(), st) == -1
    assert lib.svbgpu_encode(0, 0, None, 5000, buf.data_ptr(), 0, ws, wsb, res.data_ptr(), st) == -1
    assert lib.svbgpu_decode(0, 0, None, v.data_ptr(), 5000, 0, ws, wsb, res.data_ptr(), st) == -1
    assert lib.svbgpu_set_option(b"encode_tile", 3) == -1 and lib.svbgpu_set_option(b"no_such_knob", 1) == -1
    assert lib.svbgpu_encode_batch(0, 0, None, None, None, None, None, None, 5, None, st) == -1
    # count == 0: nothing is touched but the result words
    res.fill_(77)
    assert lib.svbgpu_encode(0, 0, None, 0, None, 0, None, 0, res.data_ptr(), st) == 0
    assert lib.svbgpu_decode(0, 0, None, None, 0, 0, None, 0, res.data_ptr() + 8, st) == 0
    assert lib.svbgpu_compressedbytes(0, 0, None, 0, 0, res.data_ptr() + 16, st) == 0
    assert lib.svbgpu_encode_batch(0, 0, None, None, None, None, None, None, 0, None, st) == 0
    torch.cuda.synchronize()
    assert res[:3].tolist() == [0, 0, 0]
    ok = codec.validate(buf, 0, 0)
    assert int(ok.item()) == 1
    assert int(codec.validate(buf, 5, 0).item()) == 0 and int(codec.validate(buf, 0, 5).item()) == 0
    # still healthy afterwards
    out, size = codec.encode(v)
    dec, used = codec.decode(out, 5000)
    assert torch.equal(dec, v) and int(used.item()) == int(size.item())


@pytest.mark.parametrize("variant", [V1234, V0124])
def test_all_small_wave_tiles(torch_mod, codec, oracle, variant):
    """Wave tiles (1024 values) whose keys are all zero take a shortcut in the decoders
    and encoders; mix them with ordinary tiles, with tiles that miss the condition
    by one value, and with a short last tile."""
    from streamvbyte_amd import DELTA, ZIGZAG
    rng = np.random.default_rng(77)
    hi = 256 if variant == V1234 else 1  # values whose code is 0
    for n in (1024, 4096 + 7, 20 * 1024, 20 * 1024 + 1000, 70000):
        v = GENERATORS["width_mix"](rng, n)
        for t in range(0, n, 1024):
            kind = (t // 1024) % 4
            if kind in (0, 1):  # all small
                v[t:t + 1024] = rng.integers(0, hi, size=min(1024, n - t), dtype=np.uint32)
            if kind == 1 and t + 1024 <= n:  # ... except one value somewhere
                v[t + int(rng.integers(0, 1024))] = 0x12345
        for flags, prev in ((0, 0), (DELTA, 3), (ZIGZAG, 0)):
            src = np.cumsum(v.astype(np.uint64)).astype(np.uint32) if flags == DELTA else v
            if flags == DELTA:
                src = (src + np.uint32(3)).astype(np.uint32)  # deltas against prev = 3 are exactly v
            expect = _expected_stream(oracle, variant, flags, src, prev)
            out, size = codec.encode(to_dev(torch_mod, src), variant=variant, flags=flags, prev=prev)
            codec.check()
            assert int(size.item()) == expect.size, (n, flags)
            assert np.array_equal(out[: expect.size].cpu().numpy(), expect), (n, flags)
            for lookback in ((0, 1) if flags == DELTA else (0,)):
                codec.lib.svbgpu_set_option(b"delta_decode_lookback", lookback)
                dec, used = codec.decode(dev_stream(torch_mod, expect, misalign=n % 3), n, variant=variant, flags=flags, prev=prev)
                codec.check()
                codec.lib.svbgpu_set_option(b"delta_decode_lookback", 1)
                assert int(used.item()) == expect.size
                assert np.array_equal(to_np(dec, np.uint32), src), (n, flags, lookback)


def test_decoders_write_exactly_count_values(torch_mod, codec, oracle):
    """The reference never writes past out[n-1] (src/streamvbyte_decode.c:74-86); neither
    may any decoder here, nor in front of out."""
    from streamvbyte_amd import DELTA, ZIGZAG
    torch = torch_mod
    guard = 64
    for n in (1, 3, 5, 1023, 1025, 4099, 16385, 70001):
        v = gen("width_mix", n, 40 + n)
        for variant, flags in ((V1234, 0), (V0124, 0), (V1234, DELTA), (V0124, ZIGZAG), (V1234, DELTA | ZIGZAG)):
            s = _expected_stream(oracle, variant, flags, v, 9 if flags & DELTA else 0)
            for lookback in ((0, 1) if flags & DELTA else (0,)):
                big = torch.full((n + 2 * guard,), 0x5A5A5A5A, dtype=torch.int32, device="cuda:0")
                codec.lib.svbgpu_set_option(b"delta_decode_lookback", lookback)
                codec.decode(dev_stream(torch, s), n, out=big[guard:guard + n], variant=variant, flags=flags,
                             prev=9 if flags & DELTA else 0)
                codec.lib.svbgpu_set_option(b"delta_decode_lookback", 1)
                h = big.cpu().numpy().view(np.uint32)
                assert np.array_equal(h[guard:guard + n], v), (n, variant, flags)
                assert (h[:guard] == 0x5A5A5A5A).all() and (h[guard + n:] == 0x5A5A5A5A).all(), (n, variant, flags)
        # zigzag_delta_decode and the plain maps
        zz = oracle.zigzag_delta_encode(v.view(np.int32), -4)
        big = torch.full((n + 2 * guard,), 0x5A5A5A5A, dtype=torch.int32, device="cuda:0")
        codec.zigzag(to_dev(torch, zz), "delta_decode", -4, out=big[guard:guard + n])
        h = big.cpu().numpy().view(np.uint32)
        assert np.array_equal(h[guard:guard + n], v)
        assert (h[:guard] == 0x5A5A5A5A).all() and (h[guard + n:] == 0x5A5A5A5A).all(), n


@pytest.mark.parametrize("variant", [V0124, V1234])
def test_quantized_floats_fused(torch_mod, codec, oracle, variant):
    """quantise -> zigzag -> encode and decode -> un-zigzag -> dequantise(-accumulate) in one
    pass each equal the same steps done one after the other (numpy float32 + oracle)."""
    torch = torch_mod
    rng = np.random.default_rng(8)
    for n in (1, 5, 1023, 4096, 16384 + 3, 100003, 300000):
        x = (rng.standard_normal(n) * rng.choice([0.3, 3.0, 300.0])).astype(np.float32)
        x[rng.random(n) < 0.3] = 0.0
        scale = np.float32(rng.choice([1.0, 7.5, 0.125]))
        q = np.rint(x * scale).astype(np.int32)  # float32 product, round half to even: what the kernel does
        expect = oracle.encode(variant, oracle.zigzag_encode(q))
        d_x = torch.from_numpy(x).cuda()
        out, size = codec.quantize_encode(d_x, float(scale), variant=variant)
        codec.check()
        assert int(size.item()) == expect.size, (n,)
        assert np.array_equal(out[: expect.size].cpu().numpy(), expect), (n,)
        step = np.float32(1.0) / scale
        want = q.astype(np.float32) * step
        dec, used = codec.decode_dequantize(dev_stream(torch, expect), n, float(step), variant=variant)
        assert int(used.item()) == expect.size
        assert np.array_equal(dec.cpu().numpy(), want), (n,)
        acc0 = rng.standard_normal(n).astype(np.float32)
        acc = torch.from_numpy(acc0.copy()).cuda()
        codec.decode_dequantize(dev_stream(torch, expect), n, float(step), out=acc, accumulate=True, variant=variant)
        assert np.array_equal(acc.cpu().numpy(), acc0 + want), (n,)


@pytest.mark.parametrize("variant", [V1234, V0124])
def test_decode_intersect(torch_mod, codec, oracle, variant):
    """Differential decode fused with an intersection = decode (oracle) then numpy isin,
    for sparse, similar and dense sorted arrays, repeats, extremes and streams that wrap."""
    torch = torch_mod
    rng = np.random.default_rng(90 + variant)
    cases = []
    for n in (1, 3, 1023, 1025, 4096, 4097, 16384, 16385, 65537, 300001, 1 << 20):
        gaps = rng.integers(0, 40, n, dtype=np.uint64)          # ascending, with repeats (gap 0)
        cases.append((np.cumsum(gaps).astype(np.uint32), 0))
    wide = np.cumsum(rng.integers(1, 3000, 200000, dtype=np.uint64)).astype(np.uint32)
    cases.append((wide, 0))
    cases.append(((np.cumsum(rng.integers(0, 40, 70000, dtype=np.uint64)) + 0xFFFF0000).astype(np.uint32), 0))  # wraps
    cases.append((GENERATORS["width_mix"](rng, 20000), 5))       # values that jump about
    for v, prev in cases:
        n = v.size
        s = oracle.encode(variant, v, True, prev)
        top = int(v.max()) + 50
        sets = [rng.integers(0, top, max(1, n // 50), dtype=np.uint64),       # much sparser than the stream
                rng.integers(0, top, n, dtype=np.uint64),                     # about as dense
                rng.integers(0, top, min(20 * n, 1 << 21), dtype=np.uint64),  # much denser
                np.array([0, 0xFFFFFFFF], dtype=np.uint64),
                np.zeros(0, dtype=np.uint64),
                v[:: 3].astype(np.uint64)]                                     # every third value certainly hits
        for b in sets:
            b = np.unique(b.astype(np.uint32))
            expect = v[np.isin(v, b)]
            d_b = to_dev(torch, b) if b.size else torch.empty(0, dtype=torch.int32, device="cuda:0")
            out, n_out, used = codec.decode_intersect(dev_stream(torch, s), n, d_b, variant=variant, prev=prev)
            codec.check()
            assert int(used.item()) == s.size
            assert int(n_out.item()) == expect.size, (n, b.size)
            assert np.array_equal(to_np(out[: expect.size], np.uint32), expect), (n, b.size)
    # empty stream
    out, n_out, used = codec.decode_intersect(dev_stream(torch, np.zeros(0, dtype=np.uint8)), 0, to_dev(torch, wide),
                                              variant=variant)
    assert int(n_out.item()) == 0 and int(used.item()) == 0


@pytest.mark.parametrize("variant_a,variant_b", [(V1234, V1234), (V0124, V1234), (V1234, V0124)])
def test_intersect_streams(torch_mod, codec, oracle, variant_a, variant_b):
    """Two delta-coded streams intersected on the device = both decoded by the oracle, then numpy isin:
    B sparser, as dense, denser than A; repeats in both; seeds; B of one value, of exactly one wave tile;
    A that wraps or jumps about (B must ascend, A need not)."""
    torch = torch_mod
    rng = np.random.default_rng(190 + 2 * variant_a + variant_b)

    def ascending(n, gap, start=0):
        return (np.cumsum(rng.integers(0, gap, n, dtype=np.uint64)) + start).astype(np.uint32)

    cases = []
    for na in (1, 5, 1023, 1024, 1025, 2048, 2049, 8192, 8193, 70001, 300001):
        for ratio in (0.02, 1.0, 17.0):
            nb = max(1, int(na * ratio))
            gap_a = 40
            gap_b = max(2, int(gap_a / ratio))
            cases.append((ascending(na, gap_a), 0, ascending(nb, gap_b), 0))
    # wide gaps (no bitmap), seeds on both sides, B starting far above / below A
    cases.append((ascending(100000, 3000, 77), 77, ascending(90000, 3300, 5), 5))
    cases.append((ascending(50000, 40, 1 << 20), 1 << 20, ascending(50000, 40), 0))
    cases.append((ascending(50000, 40), 0, ascending(50000, 40, 1 << 21), 1 << 21))
    cases.append((ascending(5000, 40), 0, np.array([4321], dtype=np.uint32), 0))
    cases.append((ascending(5000, 3), 0, ascending(1024, 9), 0))
    # A wraps past 2^32 / jumps about; B ascends
    cases.append(((np.cumsum(rng.integers(0, 40, 70000, dtype=np.uint64)) + 0xFFFF0000).astype(np.uint32), 0,
                  np.unique(rng.integers(0, 1 << 32, 400000, dtype=np.uint64).astype(np.uint32)), 0))
    cases.append((GENERATORS["width_mix"](rng, 20000), 9,
                  np.unique(GENERATORS["width_mix"](rng, 60000)), 0))
    for a, prev_a, b, prev_b in cases:
        b = np.sort(b)
        if b.size > 3:
            b[b.size // 2] = b[b.size // 2 - 1]        # a repeat in B
        sa = oracle.encode(variant_a, a, True, prev_a)
        sb = oracle.encode(variant_b, b, True, prev_b)
        expect = a[np.isin(a, b)]
        out, n_out = codec.intersect_streams(dev_stream(torch, sa), a.size, dev_stream(torch, sb), b.size,
                                             variant_a=variant_a, prev_a=prev_a, variant_b=variant_b, prev_b=prev_b)
        assert int(n_out.item()) == expect.size, (a.size, b.size)
        assert np.array_equal(to_np(out[: expect.size], np.uint32), expect), (a.size, b.size)
    # every value of A in B; an empty side
    a = ascending(40000, 50)
    sa = oracle.encode(variant_a, a, True, 0)
    sb = oracle.encode(variant_b, a, True, 0)
    out, n_out = codec.intersect_streams(dev_stream(torch, sa), a.size, dev_stream(torch, sb), a.size,
                                         variant_a=variant_a, variant_b=variant_b)
    assert int(n_out.item()) == a.size and np.array_equal(to_np(out[: a.size], np.uint32), a)
    out, n_out = codec.intersect_streams(dev_stream(torch, sa), a.size, dev_stream(torch, np.zeros(0, dtype=np.uint8)), 0,
                                         variant_a=variant_a, variant_b=variant_b)
    assert int(n_out.item()) == 0


@pytest.mark.parametrize("variant", [V1234, V0124])
def test_intersect_batch(torch_mod, codec, oracle, variant):
    """svbgpu_intersect_batch pair by pair = the oracle's decode of both lists, then numpy isin: empty lists,
    lists of one value, of exactly one and two wave tiles, long ones, B far denser / sparser than A, B wholly
    below or above A, seeds, repeats, and a few A that step back (B must ascend, A need not)."""
    torch = torch_mod
    rng = np.random.default_rng(300 + variant)
    sizes = [0, 1, 3, 1024, 1025, 2048, 2049, 4096, 5000, 20000, 70000]
    pairs = []
    for i in range(160):
        na = int(rng.choice(sizes)) if rng.random() < 0.4 else int(rng.integers(0, 6000))
        nb = int(rng.choice(sizes)) if rng.random() < 0.4 else int(rng.integers(0, 6000))
        gap_a = int(rng.choice([2, 30, 400, 70000]))
        gap_b = int(rng.choice([2, 30, 400, 70000]))
        pa = int(rng.integers(0, 1 << 20)) if rng.random() < 0.3 else 0
        pb = int(rng.integers(0, 1 << 20)) if rng.random() < 0.3 else 0
        a = (np.cumsum(rng.integers(0, gap_a, na, dtype=np.uint64)) + pa).astype(np.uint32)
        b = (np.cumsum(rng.integers(0, gap_b, nb, dtype=np.uint64)) + pb)
        b = b[b < (1 << 32)].astype(np.uint32)
        kind = i % 8
        if kind == 0 and na > 10:                      # A steps back in the middle
            a = np.concatenate([a[na // 2:], a[: na // 2]])
        elif kind == 1 and nb:                         # B wholly above A
            b = np.unique((b.astype(np.uint64) + (int(a.max()) + 5 if na else 0)) % (1 << 32)).astype(np.uint32)
            pb = 0
        elif kind == 2 and na and nb:                  # some of A's own values in B for sure
            b = np.unique(np.concatenate([b, a[:: 7]]))
            pb = 0
        if b.size and pb > int(b[0]):
            pb = int(b[0])
        pairs.append((a, pa, np.sort(b), pb))
    sa = [oracle.encode(variant, a, True, pa) if a.size else np.zeros(0, np.uint8) for a, pa, _, _ in pairs]
    sb = [oracle.encode(variant, b, True, pb) if b.size else np.zeros(0, np.uint8) for _, _, b, pb in pairs]

    def pack(streams):
        off = np.zeros(len(streams) + 1, dtype=np.int64)
        off[1:] = np.cumsum([x.size for x in streams])
        return dev_stream(torch, np.concatenate(streams)), torch.from_numpy(off[:-1].copy()).cuda()
    bytes_a, off_a = pack(sa)
    bytes_b, off_b = pack(sb)
    counts_a = torch.tensor([p[0].size for p in pairs], dtype=torch.int32, device="cuda:0")
    counts_b = torch.tensor([p[2].size for p in pairs], dtype=torch.int32, device="cuda:0")
    prevs_a = to_dev(torch, np.array([p[1] for p in pairs], dtype=np.uint32))
    prevs_b = to_dev(torch, np.array([p[3] for p in pairs], dtype=np.uint32))
    out_off = np.zeros(len(pairs) + 1, dtype=np.int64)
    out_off[1:] = np.cumsum([p[0].size for p in pairs])
    out = torch.full((int(out_off[-1]) + 1,), 0x5A5A5A5A, dtype=torch.int32, device="cuda:0")
    _, n_out = codec.intersect_batch(bytes_a, off_a, counts_a, prevs_a, bytes_b, off_b, counts_b, prevs_b, out,
                                     torch.from_numpy(out_off[:-1].copy()).cuda(), variant=variant)
    got_n = n_out.cpu().numpy()
    got = to_np(out, np.uint32)
    for i, (a, pa, b, pb) in enumerate(pairs):
        expect = a[np.isin(a, b)]
        assert got_n[i] == expect.size, (i, a.size, b.size)
        assert np.array_equal(got[out_off[i]: out_off[i] + expect.size], expect), (i, a.size, b.size)
        # nothing written past the pair's hits
        assert np.all(got[out_off[i] + expect.size: out_off[i + 1]] == 0x5A5A5A5A), (i,)


@pytest.mark.parametrize("variant", [V1234, V0124])
def test_validate_batch(torch_mod, codec, oracle, variant):
    """svbgpu_validate_batch = the oracle's validate_stream stream by stream, for right and wrong lengths,
    empty streams, lengths below the key bytes and tampered keys."""
    torch = torch_mod
    rng = np.random.default_rng(60 + variant)
    counts = np.concatenate([rng.integers(0, 40, 60), rng.integers(200, 3000, 40), [0, 0, 1, 4, 5, 1024, 1025]]).astype(np.uint32)
    streams = [oracle.encode(variant, GENERATORS["width_mix"](rng, int(c))) if c else np.zeros(0, dtype=np.uint8) for c in counts]
    off = np.concatenate([[0], np.cumsum([s.size + 3 for s in streams])]).astype(np.int64)
    buf = np.full(int(off[-1]) + 16, 0xEE, dtype=np.uint8)
    for s, o in zip(streams, off[:-1]):
        buf[o:o + s.size] = s
    lengths = np.array([s.size for s in streams], dtype=np.int64)
    # wrong in different ways
    lengths[3] += 1
    lengths[7] = max(int(lengths[7]) - 1, 0)
    lengths[61] = 5                      # below the key bytes of a long list
    lengths[100] = 2                     # count 0 with a length
    k = 70                               # a tampered key byte of a long list
    buf[off[k] + 10] ^= 0x0C
    expect = np.array([oracle.validate(variant, buf[o:o + int(l)], int(c), int(l)) for o, l, c in zip(off[:-1], lengths, counts)],
                      dtype=np.int32)
    assert expect.sum() < counts.size and expect.sum() > counts.size // 2
    got = codec.validate_batch(torch.from_numpy(buf).cuda(), torch.from_numpy(off[:-1].copy()).cuda(),
                               torch.from_numpy(lengths).cuda(), to_dev(torch, counts), variant=variant)
    assert np.array_equal(got.cpu().numpy(), expect), np.flatnonzero(got.cpu().numpy() != expect)


def test_device_api_from_concurrent_threads(torch_mod, oracle):
    """The device-pointer API keeps no state outside what the caller passes in: eight threads, each with
    its own workspace, result words and HIP stream, encode and decode at once and get the oracle's bytes.
    (No lock inside the library covers these calls.)"""
    import threading
    import streamvbyte_amd as svb
    torch = torch_mod
    errors = []

    def worker(t):
        try:
            rng = np.random.default_rng(700 + t)
            codec = svb.DeviceCodec(max_count=1 << 21)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for i in range(10):
                    n = int(rng.integers(1, 1 << 21))
                    variant = (t + i) & 1
                    flags = svb.DELTA if (i % 3 == 0) else 0
                    v = GENERATORS[["width_mix", "postings", "perf_c", "sparse"][(t + i) % 4]](rng, n)
                    expect = oracle.encode(variant, v, bool(flags), t)
                    d_v = to_dev(torch, v)
                    out, size = codec.encode(d_v, variant=variant, flags=flags, prev=t)
                    back, used = codec.decode(out, n, variant=variant, flags=flags, prev=t)
                    stream.synchronize()
                    codec.check()
                    assert int(size.item()) == expect.size == int(used.item()), (t, i)
                    assert np.array_equal(out[: expect.size].cpu().numpy(), expect), (t, i)
                    assert np.array_equal(to_np(back, np.uint32), v), (t, i)
        except BaseException as exc:  # noqa: BLE001 - reported by the main thread
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]


def test_value_arrays_at_dword_alignment_with_nontemporal_accesses(torch_mod, codec, oracle):
    """The 16-byte stream accessors (ld_stream16 / st_stream16) get value pointers that are only 4-byte
    aligned: input and output one value past a 16-byte boundary, every cache policy forced in turn."""
    torch = torch_mod
    from streamvbyte_amd import DELTA
    n = (1 << 20) + 3
    v = gen("width_mix", n, 41)
    base_in = torch.zeros(n + 4, dtype=torch.int32, device="cuda:0")
    base_out = torch.zeros(n + 4, dtype=torch.int32, device="cuda:0")
    for shift in (1, 2, 3):
        d_v, d_out = base_in[shift:shift + n], base_out[shift:shift + n]
        assert d_v.data_ptr() % 16 == 4 * shift
        d_v.copy_(to_dev(torch, v))
        for nt in (3, 0):
            codec.lib.svbgpu_set_option(b"nt_force", nt)
            try:
                for variant, flags, prev in ((V1234, 0, 0), (V0124, 0, 0), (V1234, DELTA, 9)):
                    expect = oracle.encode(variant, v, bool(flags), prev)
                    out, size = codec.encode(d_v, variant=variant, flags=flags, prev=prev)
                    assert int(size.item()) == expect.size
                    assert np.array_equal(out[: expect.size].cpu().numpy(), expect), (shift, nt, variant, flags)
                    base_out.fill_(-1)
                    codec.decode(dev_stream(torch, expect), n, out=d_out, variant=variant, flags=flags, prev=prev)
                    assert np.array_equal(to_np(d_out, np.uint32), v), (shift, nt, variant, flags)
                    assert bool((base_out[:shift] == -1).all()) and bool((base_out[shift + n:] == -1).all())
            finally:
                codec.lib.svbgpu_set_option(b"nt_force", -1)
    codec.check()
