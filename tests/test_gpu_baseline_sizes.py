"""Whole-stream parity at the sizes the benchmark configurations name: every
encoder byte and every decoded value of the full stream against the CPU
oracle (not a prefix, not a round trip).

  C2  one stream of 2^28 values, byte widths uniform over 1..4 (and 2^28 - 3: ragged tail)
  C3  2^28 dense postings, delta coded, prev != 0
  C4  2^26 gradients, zigzag + 0124
  C5  2^20 posting lists, delta coded, per-list bytes

Reference behaviour matched: src/streamvbyte_encode.c:39-62,
src/streamvbyte_decode.c:41-87, src/streamvbytedelta_decode.c:34-57,
src/streamvbyte_0124_encode.c:34-57, src/streamvbyte_zigzag.c:3-29 there.
The oracle runs once per stream on the host (about a second per 2^28 values);
comparison happens on the device so that only the streams cross PCIe.
"""
import numpy as np
import pytest

from oracle import V1234, V0124

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import streamvbyte_amd as svb
    assert torch.cuda.is_available()
    return torch, svb, svb.DeviceCodec(max_count=1 << 28)


def _width_mix(n, seed):
    rng = np.random.default_rng(seed)
    w = rng.integers(1, 5, size=n, dtype=np.uint8)
    v = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    # value uniform inside its width class: keep the low 8w bits and force the top byte of the class non-zero
    shift = (8 * (4 - w)).astype(np.uint32)
    v >>= shift
    low = np.where(w == 1, 0, np.uint32(1) << (8 * (w.astype(np.uint32) - 1))).astype(np.uint32)
    np.maximum(v, low, out=v)
    return v


def _check_stream(env, oracle, variant, flags, v_u32, prev, oracle_delta, oracle_in=None):
    """encode v on the device and compare every byte with the oracle's stream; decode the ORACLE's
    stream on the device and compare every value with v."""
    torch, svb, codec = env
    n = v_u32.size
    expect = oracle.encode(variant, v_u32 if oracle_in is None else oracle_in, oracle_delta, prev)
    d_v = torch.from_numpy(v_u32.view(np.int32)).to("cuda:0")
    out, size = codec.encode(d_v, variant=variant, flags=flags, prev=prev)
    codec.check()
    assert int(size.item()) == expect.size
    d_expect = torch.empty(expect.size + 16, dtype=torch.uint8, device="cuda:0")
    d_expect[: expect.size] = torch.from_numpy(expect).to("cuda:0")
    d_expect[expect.size:] = 0
    same = torch.equal(out[: expect.size], d_expect[: expect.size])
    if not same:
        bad = torch.nonzero(out[: expect.size] != d_expect[: expect.size])[:5].flatten().tolist()
        raise AssertionError(f"encoder bytes differ from the oracle's at {bad} of {expect.size}")
    del out
    assert int(codec.compressedbytes(d_v, variant=variant, flags=flags, prev=prev).item()) == expect.size
    assert int(codec.validate(d_expect, expect.size, n, variant=variant).item()) == 1
    assert int(codec.validate(d_expect, expect.size - 1, n, variant=variant).item()) == 0
    dec, used = codec.decode(d_expect, n, variant=variant, flags=flags, prev=prev)
    codec.check()
    assert int(used.item()) == expect.size
    if not torch.equal(dec, d_v):
        bad = torch.nonzero(dec != d_v)[:5].flatten().tolist()
        raise AssertionError(f"decoded values differ at {bad} of {n}")


@pytest.mark.parametrize("n", [1 << 28, (1 << 28) - 3])
def test_c2_uniform_widths_whole_stream(env, oracle, n):
    v = _width_mix(n, 0xC2)
    _check_stream(env, oracle, V1234, 0, v, 0, False)
    # and the oracle's own decode of its stream gives v back (the checker checks itself at this size once)
    if n == 1 << 28:
        s = oracle.encode(V1234, v)
        back, used = oracle.decode(V1234, s, n)
        assert used == s.size and np.array_equal(back, v)


def test_c3_dense_postings_delta_whole_stream(env, oracle):
    torch, svb, codec = env
    n = 1 << 28
    rng = np.random.default_rng(0xC3)
    prev = 0xDEADBEEF
    v = (np.cumsum(rng.integers(0, 32, size=n, dtype=np.uint64), dtype=np.uint64) + prev).astype(np.uint32)
    _check_stream(env, oracle, V1234, svb.DELTA, v, prev, True)


def test_c3_mixed_gaps_delta_whole_stream(env, oracle):
    torch, svb, codec = env
    n = 1 << 26
    rng = np.random.default_rng(0xC33)
    gaps = (rng.integers(0, 1 << 31, size=n, dtype=np.uint64) >> rng.integers(0, 32, size=n, dtype=np.uint64)) >> 8
    v = (np.cumsum(gaps, dtype=np.uint64) + 0xDEADBEEF).astype(np.uint32)  # wraps modulo 2^32
    _check_stream(env, oracle, V1234, svb.DELTA, v, 0xDEADBEEF, True)


def test_c4_gradients_zigzag_0124_whole_stream(env, oracle):
    torch, svb, codec = env
    n = 1 << 26
    rng = np.random.default_rng(0xC4)
    g = np.rint(rng.normal(0.0, 0.8, size=n)).astype(np.int64)
    outl = rng.random(n) < 0.001
    g[outl] = rng.integers(-(1 << 20), 1 << 20, size=int(outl.sum()))
    g = g.astype(np.int32)
    zz = oracle.zigzag_encode(g)  # the oracle's two separate steps: zigzag_encode, then encode_0124
    _check_stream(env, oracle, V0124, svb.ZIGZAG, g.view(np.uint32), 0, False, oracle_in=zz)


def test_c5_million_posting_lists_per_list_bytes(env, oracle):
    """2^20 lists, lengths log-uniform on 1..4096 (about 5e8 values), delta coded with prev = 0:
    every list's bytes and size against the oracle's streamvbyte_delta_encode of that list, and every
    value of the batched decode of the oracle's streams."""
    torch, svb, codec = env
    n_lists = 1 << 20
    rng = np.random.default_rng(0xC5)
    counts = np.clip(np.floor(np.exp(rng.uniform(0.0, np.log(4096.0), size=n_lists))), 1, 4096).astype(np.int64)
    voff = np.zeros(n_lists + 1, dtype=np.int64)
    np.cumsum(counts, out=voff[1:])
    total = int(voff[-1])
    gaps = rng.integers(0, 1 << 12, size=total, dtype=np.uint32)
    small = rng.random(total) < 0.7
    gaps[small] &= 0xFF
    cs = np.cumsum(gaps, dtype=np.uint64)
    first = cs[voff[:-1]] - gaps[voff[:-1]]
    vals = (cs - np.repeat(first, counts)).astype(np.uint32)
    del cs, gaps, small
    # the oracle, list by list, into worst-case slots (the layout the batched encoder is given)
    cap = (counts + 3) // 4 + 4 * counts + 16
    boff = np.zeros(n_lists + 1, dtype=np.int64)
    np.cumsum(cap, out=boff[1:])
    expect = np.zeros(int(boff[-1]) + 16, dtype=np.uint8)
    exp_sizes = np.zeros(n_lists, dtype=np.int64)
    import ctypes as C
    u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    enc = oracle.lib.svbo_encode
    vbase, ebase = vals.ctypes.data, expect.ctypes.data
    for i in range(n_lists):
        exp_sizes[i] = enc(V1234, C.cast(vbase + 4 * int(voff[i]), u32p), int(counts[i]),
                           C.cast(ebase + int(boff[i]), u8p), 1, 0)
    d_vals = torch.from_numpy(vals.view(np.int32)).to("cuda:0")
    d_counts = torch.from_numpy(counts.astype(np.int32)).to("cuda:0")
    d_voff = torch.from_numpy(voff[:-1].copy()).to("cuda:0")
    d_boff = torch.from_numpy(boff[:-1].copy()).to("cuda:0")
    d_bytes = torch.zeros(expect.size, dtype=torch.uint8, device="cuda:0")
    sizes = codec.encode_batch(d_vals, d_voff, d_bytes, d_boff, d_counts, None, flags=svb.DELTA)
    torch.cuda.synchronize()
    assert torch.equal(sizes.cpu(), torch.from_numpy(exp_sizes)), "per-list sizes differ from the oracle's"
    # bytes inside each list's size must equal the oracle's; bytes beyond it are unspecified -> mask them
    d_expect = torch.from_numpy(expect).to("cuda:0")
    for lo in range(0, expect.size, 1 << 28):
        hi = min(expect.size, lo + (1 << 28))
        pos = torch.arange(lo, hi, device="cuda:0", dtype=torch.int64)
        owner = (torch.searchsorted(d_boff, pos, right=True) - 1).clamp_(max=n_lists - 1)
        inside = (pos - d_boff[owner]) < sizes[owner]
        diff = (d_bytes[lo:hi] != d_expect[lo:hi]) & inside
        assert not bool(diff.any()), f"list bytes differ at {(torch.nonzero(diff)[:5].flatten() + lo).tolist()}"
        del pos, owner, inside, diff
    del d_bytes
    back = torch.full_like(d_vals, -1)
    used = codec.decode_batch(d_expect, d_boff, back, d_voff, d_counts, None, flags=svb.DELTA)
    torch.cuda.synchronize()
    assert torch.equal(used.cpu(), torch.from_numpy(exp_sizes))
    assert torch.equal(back, d_vals)
    ok = codec.validate_batch(d_expect, d_boff, torch.from_numpy(exp_sizes).to("cuda:0"), d_counts)
    assert bool((ok == 1).all())
