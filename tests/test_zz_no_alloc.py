"""svbgpu_decode_static (the entry point that allocates nothing: its workspace is a
static device array inside the library) against svbgpu_decode on the same input,
bit for bit, through the C ABI.  Two calls in a row share the array, the second
with a smaller input: whatever the first one left in the scratch must not leak
into the second."""
import numpy as np
import pytest

from oracle import V1234, V0124
from util import GENERATORS, dev_stream, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import streamvbyte_amd as svb
    assert torch.cuda.is_available()
    return torch, svb, svb.load(), svb.DeviceCodec(max_count=1 << 23)


def _both(env, oracle, variant, flags, v, prev):
    """decode the oracle's stream with both entry points; returns (static, regular, consumed x2, expected size)"""
    torch, svb, lib, codec = env
    n = v.size
    s = oracle.encode(variant, v, bool(flags & svb.DELTA), prev)
    d_in = dev_stream(torch, s)
    ref, used_ref = codec.decode(d_in, n, variant=variant, flags=flags, prev=prev)
    codec.check()
    out = torch.full((n + 8,), -1, dtype=torch.int32, device="cuda:0")  # 8 guard values behind the output
    used = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.svbgpu_decode_static(variant, flags, d_in.data_ptr(), out.data_ptr(), n, prev, used.data_ptr(), st)
    assert rc == 0, lib.svbgpu_strerror(rc)
    err = torch.full((1,), 7, dtype=torch.int32, device="cuda:0")
    assert lib.svbgpu_static_status(err.data_ptr(), st) == 0   # the scratch's error word, without a workspace pointer
    torch.cuda.synchronize()
    assert int(err.item()) == 0, "a bounded spin of the static-scratch decode gave up"
    assert bool((out[n:] == -1).all()), "wrote past the output"
    return to_np(out[:n], np.uint32), to_np(ref, np.uint32), int(used.item()), int(used_ref.item()), s.size


@pytest.mark.parametrize("variant", [V1234, V0124])
@pytest.mark.parametrize("flags", [0, 1])  # plain (directory only) and DELTA (directory + look-back words)
def test_static_scratch_decode_equals_workspace_decode(env, oracle, variant, flags):
    torch, svb, lib, codec = env
    rng = np.random.default_rng(77 + 2 * variant + flags)
    # first a large input (many look-back tiles, several super tiles), then a smaller one right after it
    for n, kind in ((3 * (1 << 20) + 12345, "width_mix"), (70001, "perf_c"), (5, "full"), (1 << 20, "postings")):
        v = GENERATORS[kind](rng, n)
        got, ref, used, used_ref, size = _both(env, oracle, variant, flags, v, prev=0xDEADBEEF if flags else 0)
        assert used == used_ref == size, (n, kind)
        assert np.array_equal(got, ref), (n, kind, np.flatnonzero(got != ref)[:5])
        assert np.array_equal(got, v), (n, kind)


def test_static_scratch_limits(env):
    torch, svb, lib, codec = env
    st = torch.cuda.current_stream().cuda_stream
    dummy = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    too_big = (1 << 28) + 1
    rc = lib.svbgpu_decode_static(0, 0, dummy.data_ptr(), dummy.data_ptr(), too_big, 0, None, st)
    assert rc == -6 and b"static scratch" in lib.svbgpu_strerror(rc)
    assert lib.svbgpu_decode_static(0, 0, None, None, 0, 0, None, st) == 0       # empty stream: nothing to do
    assert lib.svbgpu_decode_static(7, 0, dummy.data_ptr(), dummy.data_ptr(), 4, 0, None, st) == -1
    assert lib.svbgpu_static_status(None, st) == -1
