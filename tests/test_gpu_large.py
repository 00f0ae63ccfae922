"""GPU parity at the largest sizes the API admits (count is a uint32, reference
include/streamvbyte.h:23,64): byte offsets beyond 2^32, value indices beyond 2^31,
count = 2^32 - 1.  The oracle checks the head and the tail of each stream byte for
byte (a tail that starts on a group boundary codes the same bytes on its own);
the bulk is checked by size, validate and the decode round trip.
"""
import numpy as np
import pytest

from oracle import V1234, V0124

pytestmark = pytest.mark.gpu

HEAD = 1 << 16
TAIL = 50001


def _values(torch, n, seed):
    """width-mix uint32 (1..4 significant bytes, uniform), as int32 bits"""
    g = torch.Generator(device="cuda:0")
    g.manual_seed(seed)
    out = torch.empty(n, dtype=torch.int32, device="cuda:0")
    chunk = 1 << 27
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        bits = torch.randint(-(1 << 31), (1 << 31), (m,), generator=g, device="cuda:0", dtype=torch.int64).to(torch.int32)
        drop = 8 * torch.randint(0, 4, (m,), generator=g, device="cuda:0", dtype=torch.int32)
        # logical right shift by 0, 8, 16 or 24 bits
        out[s:s + m] = ((bits.to(torch.int64) & 0xFFFFFFFF) >> drop.to(torch.int64)).to(torch.int32)
        del bits, drop
    return out


@pytest.mark.parametrize("count,variant,delta", [
    ((1 << 30) + 5, V1234, False),
    ((1 << 31) + 1027, V0124, False),
    ((1 << 32) - 1, V1234, False),
    ((1 << 32) - 1, V1234, True),
])
def test_large_stream(oracle, count, variant, delta):
    import torch
    import streamvbyte_amd as svb
    free, _ = torch.cuda.mem_get_info()
    need = 4 * count * 2 + svb.streamvbyte_max_compressedbytes(count) + (4 << 30)
    if free < need:
        pytest.skip("not enough device memory")
    codec = svb.DeviceCodec(count)
    flags = svb.DELTA if delta else 0
    prev = 0x9E3779B9 if delta else 0
    v = _values(torch, count, 77 + (count & 0xFF))
    if delta:
        v = torch.cumsum(v, 0, dtype=torch.int64).to(torch.int32)  # wraps modulo 2^32 like the codec
    buf = torch.empty(svb.streamvbyte_max_compressedbytes(count) + 16, dtype=torch.uint8, device="cuda:0")
    _, size_t = codec.encode(v, out=buf, variant=variant, flags=flags, prev=prev)
    codec.check()
    size = int(size_t.item())
    keylen = (count + 3) // 4
    assert int(codec.compressedbytes(v, variant=variant, flags=flags, prev=prev).item()) == size
    if not delta:
        assert int(codec.validate(buf, size, count, variant=variant).item()) == 1
        assert int(codec.validate(buf, size - 1, count, variant=variant).item()) == 0

    # head against the oracle
    head = v[:HEAD].cpu().numpy().view(np.uint32)
    exp = oracle.encode(variant, head, delta=delta, prev=prev)
    hk = HEAD // 4
    assert np.array_equal(buf[:hk].cpu().numpy(), exp[:hk])
    assert np.array_equal(buf[keylen:keylen + exp.size - hk].cpu().numpy(), exp[hk:])
    # tail against the oracle (starts on a group boundary)
    s0 = (count - TAIL) & ~3
    tail = v[s0:].cpu().numpy().view(np.uint32)
    tprev = int(v[s0 - 1].item()) & 0xFFFFFFFF if delta else 0
    exp = oracle.encode(variant, tail, delta=delta, prev=tprev)
    tk = (tail.size + 3) // 4
    assert np.array_equal(buf[s0 // 4:keylen].cpu().numpy(), exp[:tk])
    assert np.array_equal(buf[size - (exp.size - tk):size].cpu().numpy(), exp[tk:])

    # round trip of everything
    back = torch.empty(count, dtype=torch.int32, device="cuda:0")
    _, used = codec.decode(buf, count, out=back, variant=variant, flags=flags, prev=prev)
    codec.check()
    assert int(used.item()) == size
    assert torch.equal(back, v)
    del back, buf, v, codec
    torch.cuda.empty_cache()


def test_zigzag_maps_beyond_2_to_32():
    """zigzag_encode / zigzag_decode take a size_t length (reference include/streamvbyte_zigzag.h:14,24)."""
    import torch
    import streamvbyte_amd as svb
    n = (1 << 32) + 7
    free, _ = torch.cuda.mem_get_info()
    if free < 4 * n * 4 + (4 << 30):
        pytest.skip("not enough device memory")
    codec = svb.DeviceCodec(1024)
    v = _values(torch, n, 5)
    v[-7:] = torch.tensor([-1, 1, -2, 2, 0, -(1 << 31), (1 << 31) - 1], dtype=torch.int32, device="cuda:0")
    zz = codec.zigzag(v, "encode")
    expect_tail = [1, 2, 3, 4, 0, 0xFFFFFFFF - (1 << 32), 0xFFFFFFFE - (1 << 32)]
    assert zz[-7:].tolist() == expect_tail
    ref = (v << 1) ^ (v >> 31)
    assert torch.equal(zz, ref)
    del ref
    back = codec.zigzag(zz, "decode")
    assert torch.equal(back, v)
    del back, zz, v, codec
    torch.cuda.empty_cache()
