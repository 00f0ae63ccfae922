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
