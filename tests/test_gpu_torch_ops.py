"""torch.ops.svb (streamvbyte_amd/torch_ops.py): the operators give the oracle's bytes and values."""
import numpy as np
import pytest

from oracle import V1234, V0124
from util import GENERATORS, dev_stream, to_dev, to_np

pytestmark = pytest.mark.gpu


def test_ops_match_oracle(oracle):
    import torch
    import streamvbyte_amd as svb
    import streamvbyte_amd.torch_ops  # noqa: F401 - registers torch.ops.svb
    rng = np.random.default_rng(12)
    for n, kind in ((1, "full"), (1000, "width_mix"), (100003, "perf_c"), ((1 << 21) + 5, "postings")):
        v = GENERATORS[kind](rng, n)
        d_v = to_dev(torch, v)
        for variant, flags, prev in ((V1234, 0, 0), (V0124, 0, 0), (V1234, svb.DELTA, 77), (V0124, svb.ZIGZAG, 0)):
            src = oracle.zigzag_encode(v.view(np.int32)) if flags & svb.ZIGZAG else v
            expect = oracle.encode(variant, src, bool(flags & svb.DELTA), prev)
            stream, size = torch.ops.svb.encode(d_v, variant, flags, prev)
            assert int(size.item()) == expect.size
            assert np.array_equal(stream[: expect.size].cpu().numpy(), expect)
            assert int(torch.ops.svb.compressedbytes(d_v, variant, flags, prev).item()) == expect.size
            assert int(torch.ops.svb.validate(stream, expect.size, n, variant).item()) == 1
            assert int(torch.ops.svb.validate(stream, expect.size + 1, n, variant).item()) == 0
            back = torch.ops.svb.decode(dev_stream(torch, expect), n, variant, flags, prev)
            assert np.array_equal(to_np(back, np.uint32), v)
    with pytest.raises(Exception):
        torch.ops.svb.encode(d_v.to(torch.int64), 0, 0, 0)


def test_ops_on_two_torch_streams_without_synchronising(oracle):
    """One thread, two torch streams, calls interleaved with no synchronisation between them: each call
    owns its workspace, so neither stream's directory or size words are touched by the other's calls."""
    import torch
    import streamvbyte_amd as svb
    import streamvbyte_amd.torch_ops  # noqa: F401
    rng = np.random.default_rng(77)
    n = (1 << 22) + 3
    va, vb = GENERATORS["width_mix"](rng, n), GENERATORS["postings"](rng, n)
    ea, eb = oracle.encode(V1234, va, False, 0), oracle.encode(V1234, vb, True, 5)
    d_a, d_b = to_dev(torch, va), to_dev(torch, vb)
    s_a, s_b = dev_stream(torch, ea), dev_stream(torch, eb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    got = []
    for _ in range(6):
        with torch.cuda.stream(s1):
            enc_a = torch.ops.svb.encode(d_a, V1234, 0, 0)
            dec_a = torch.ops.svb.decode(s_a, n, V1234, 0, 0)
        with torch.cuda.stream(s2):
            enc_b = torch.ops.svb.encode(d_b, V1234, svb.DELTA, 5)
            dec_b = torch.ops.svb.decode(s_b, n, V1234, svb.DELTA, 5)
        got.append((enc_a, dec_a, enc_b, dec_b))
    torch.cuda.synchronize()
    for (st_a, sz_a), dec_a, (st_b, sz_b), dec_b in got:
        assert int(sz_a.item()) == ea.size and int(sz_b.item()) == eb.size
        assert np.array_equal(st_a[: ea.size].cpu().numpy(), ea)
        assert np.array_equal(st_b[: eb.size].cpu().numpy(), eb)
        assert np.array_equal(to_np(dec_a, np.uint32), va)
        assert np.array_equal(to_np(dec_b, np.uint32), vb)
