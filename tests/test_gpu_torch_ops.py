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
