"""64-bit values, the 1248 variant (svbgpu_encode64 / svbgpu_decode64): every byte and every
value against the oracle's restatement of the format rule.  The reference has no 64-bit codec
(include/streamvbyte.h:23 there), so this parity is UNPINNED: there is no reference output or
fixture to anchor the oracle to, only the rule both sides implement and a hand-computed vector."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4097, 65537, 262144 + 7, (1 << 22) + 3]


def gen64(kind, rng, n):
    r = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    if kind == "mix":      # widths 1, 2, 4, 8 bytes equally likely
        w = rng.integers(0, 4, size=n)
        return r >> (np.uint64(64) - (np.uint64(8) << w.astype(np.uint64)))
    if kind == "shift":    # every bit length
        return r >> rng.integers(0, 64, size=n).astype(np.uint64)
    if kind == "small":
        return r & np.uint64(0xFF)
    if kind == "max":
        return np.full(n, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    return np.zeros(n, dtype=np.uint64)


@pytest.fixture(scope="module")
def env():
    import torch
    import streamvbyte_amd as svb
    assert torch.cuda.is_available()
    return torch, svb, svb.DeviceCodec(max_count=1 << 23)


def test_hand_computed_vector(env, oracle):
    torch, svb, codec = env
    v = np.array([1, 256, 65536, 1 << 32, 5], dtype=np.uint64)
    expect = bytes.fromhex("e400" "01" "0001" "00000100" "0000000001000000" "05")
    assert oracle.encode64(v).tobytes() == expect
    out, size = codec.encode64(torch.from_numpy(v.view(np.int64)).cuda())
    assert int(size.item()) == len(expect) and out[: len(expect)].cpu().numpy().tobytes() == expect


@pytest.mark.parametrize("kind", ["mix", "shift", "small", "max", "zeros"])
def test_encode64_decode64_match_oracle(env, oracle, kind):
    torch, svb, codec = env
    rng = np.random.default_rng(64)
    for i, n in enumerate(SIZES):
        v = gen64(kind, rng, n)
        expect = oracle.encode64(v)
        d_v = torch.from_numpy(v.view(np.int64)).cuda()
        mis = 3 if i % 2 else 0
        buf = torch.full((mis + (n + 3) // 4 + 8 * n + 32,), 0xAB, dtype=torch.uint8, device="cuda:0")
        out, size = codec.encode64(d_v, out=buf[mis:])
        torch.cuda.synchronize()
        assert int(size.item()) == expect.size, (kind, n)
        got = buf.cpu().numpy()
        assert np.array_equal(got[mis: mis + expect.size], expect), (kind, n)
        assert np.all(got[:mis] == 0xAB) and np.all(got[mis + expect.size:] == 0xAB), "wrote outside the stream"
        # the oracle's stream, with exactly 32 readable bytes behind it
        pad = np.zeros(mis + expect.size + 32, dtype=np.uint8)
        pad[mis: mis + expect.size] = expect
        d_in = torch.from_numpy(pad).cuda()[mis:]
        back = torch.full((n + 4,), -1, dtype=torch.int64, device="cuda:0")
        dec, used = codec.decode64(d_in, n, out=back[:n])
        torch.cuda.synchronize()
        assert int(used.item()) == expect.size, (kind, n)
        assert np.array_equal(back[:n].cpu().numpy().view(np.uint64), v), (kind, n)
        assert bool((back[n:] == -1).all()), "wrote past the output"
        assert int(codec.validate(d_in, expect.size, n, variant=2).item()) == 1
        assert int(codec.validate(d_in, expect.size + 1, n, variant=2).item()) == 0


def _transform64(v, flags, prev):
    """What SVBGPU_DELTA / SVBGPU_ZIGZAG code for 64-bit values (include/streamvbyte_gpu.h), in numpy:
    wrap-around difference to the previous value, then zigzag.  Like the rest of the 1248 variant this has
    no counterpart in the reference (parity unpinned); the rule mirrors its 32-bit forms
    (src/streamvbytedelta_encode.c:54-55, src/streamvbyte_zigzag.c:3-6 there)."""
    x = v.copy()
    if flags & 1:
        before = np.concatenate([np.array([prev], dtype=np.uint64), v[:-1]]) if v.size else v
        x = v - before   # uint64 arithmetic wraps
    if flags & 2:
        x = (x << np.uint64(1)) ^ (np.uint64(0) - (x >> np.uint64(63)))
    return x


@pytest.mark.parametrize("flags", [1, 2, 3])
def test_encode64_decode64_flags(env, oracle, flags):
    torch, svb, codec = env
    rng = np.random.default_rng(640 + flags)
    prev = 0xFEDCBA9876543210
    for n in [1, 5, 64, 255, 1025, 4096 * 3 + 1, 65537, (1 << 21) + 11]:
        for kind in ("sorted", "mix", "small"):
            if kind == "sorted":   # ascending 64-bit postings with gaps of every width
                v = np.cumsum(gen64("shift", rng, n) >> np.uint64(20), dtype=np.uint64)
            else:
                v = gen64(kind, rng, n)
            with np.errstate(over="ignore"):
                expect = oracle.encode64(_transform64(v, flags, np.uint64(prev)))
            d_v = torch.from_numpy(v.view(np.int64)).cuda()
            buf = torch.full(((n + 3) // 4 + 8 * n + 32,), 0xCD, dtype=torch.uint8, device="cuda:0")
            out, size = codec.encode64(d_v, out=buf, flags=flags, prev=prev)
            torch.cuda.synchronize()
            codec.check()
            assert int(size.item()) == expect.size, (flags, kind, n)
            got = buf.cpu().numpy()
            assert np.array_equal(got[: expect.size], expect), (flags, kind, n)
            assert np.all(got[expect.size:] == 0xCD), "wrote outside the stream"
            pad = np.zeros(expect.size + 32, dtype=np.uint8)
            pad[: expect.size] = expect
            dec, used = codec.decode64(torch.from_numpy(pad).cuda(), n, flags=flags, prev=prev)
            torch.cuda.synchronize()
            assert int(used.item()) == expect.size
            assert np.array_equal(dec.cpu().numpy().view(np.uint64), v), (flags, kind, n)


@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_encode64_decode64_batch(env, oracle, flags):
    """svbgpu_encode64_batch / svbgpu_decode64_batch: every stream's bytes equal the oracle's for that stream
    alone, nothing is written between the streams, the decoder gives the values back; stream borders at every
    alignment, counts from 0 over tile edges to several wave tiles."""
    torch, svb, codec = env
    rng = np.random.default_rng(6400 + flags)
    counts = np.array([0, 1, 2, 3, 4, 5, 7, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2049, 4099, 0, 9, 5000, 1, 3071],
                      dtype=np.uint32)
    ns = counts.size
    kinds = ["mix", "shift", "small", "max", "zeros", "sorted"]
    vals, prevs = [], rng.integers(0, 1 << 63, size=ns, dtype=np.uint64) * np.uint64(3)
    for s, n in enumerate(counts):
        kind = kinds[s % len(kinds)]
        if kind == "sorted":
            vals.append(np.cumsum(gen64("shift", rng, int(n)) >> np.uint64(20), dtype=np.uint64))
        else:
            vals.append(gen64(kind, rng, int(n)))
    value_off = np.zeros(ns, dtype=np.uint64)
    value_off[1:] = np.cumsum(counts[:-1])
    # places: worst case of each stream plus a gap of 0..5 bytes, so that streams start at every alignment
    worst = (counts.astype(np.uint64) + 3) // 4 + 8 * counts.astype(np.uint64)
    gap = rng.integers(0, 6, size=ns).astype(np.uint64)
    byte_off = np.zeros(ns, dtype=np.uint64)
    byte_off[1:] = np.cumsum(worst[:-1] + gap[:-1])
    total = int(byte_off[-1] + worst[-1]) + 32
    allv = np.concatenate(vals) if ns else np.zeros(0, dtype=np.uint64)
    d_vals = torch.from_numpy(allv.view(np.int64)).cuda()
    d_voff = torch.from_numpy(value_off.view(np.int64)).cuda()
    d_boff = torch.from_numpy(byte_off.view(np.int64)).cuda()
    d_cnt = torch.from_numpy(counts.view(np.int32)).cuda()
    d_prev = torch.from_numpy(prevs.view(np.int64)).cuda()
    buf = torch.full((total,), 0xEE, dtype=torch.uint8, device="cuda:0")
    sizes = codec.encode64_batch(d_vals, d_voff, buf, d_boff, d_cnt, prevs=d_prev, flags=flags)
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    sz = sizes.cpu().numpy()
    covered = np.zeros(total, dtype=bool)
    packed = []
    for s in range(ns):
        with np.errstate(over="ignore"):
            expect = oracle.encode64(_transform64(vals[s], flags, prevs[s]))
        assert int(sz[s]) == expect.size, (flags, s, counts[s])
        o = int(byte_off[s])
        assert np.array_equal(got[o: o + expect.size], expect), (flags, s, counts[s])
        covered[o: o + expect.size] = True
        packed.append(expect)
    assert np.all(got[~covered] == 0xEE), "wrote outside the streams"
    # decode from a packed buffer: streams back to back (each followed by its neighbour), 32 bytes behind the last
    poff = np.zeros(ns, dtype=np.uint64)
    poff[1:] = np.cumsum([p.size for p in packed[:-1]])
    pk = np.concatenate(packed + [np.zeros(32, dtype=np.uint8)])
    d_pk = torch.from_numpy(pk).cuda()
    d_poff = torch.from_numpy(poff.view(np.int64)).cuda()
    back = torch.full((allv.size + 4,), -1, dtype=torch.int64, device="cuda:0")
    used = codec.decode64_batch(d_pk, d_poff, back, d_voff, d_cnt, prevs=d_prev, flags=flags)
    torch.cuda.synchronize()
    assert np.array_equal(used.cpu().numpy(), np.array([p.size for p in packed]))
    assert np.array_equal(back[: allv.size].cpu().numpy().view(np.uint64), allv), flags
    assert bool((back[allv.size:] == -1).all()), "wrote past the output"
