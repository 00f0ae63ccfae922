"""The decoupled look-back under hostile timing.  `debug_ablate` bit 64 makes a pseudo-random quarter of
the tiles sleep for tens of microseconds right before they publish their word, before a group leader
publishes the group's aggregate and before it publishes the group's prefix (lb_stress_delay in
csrc/svb_lookback.h), so words arrive late and out of order.  Every kernel that chains tiles through the
look-back -- both encoders, the differential decoder, the running sum, the intersection -- must still
give the oracle's bytes and values, and no bounded spin may give up.

The delays are compiled only into libstreamvbyte_stress.so.2 (in the product build even a never-taken
branch there cost the encoder 2.5-3 %), so the cases run in a child process that loads that build through
SVB_LIB_PATH; this process, which has the product library loaded, only starts the child."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import V1234, V0124
from util import GENERATORS, dev_stream, to_dev, to_np

pytestmark = pytest.mark.gpu


CHILD = os.environ.get("SVB_STRESS_CHILD") == "1"


@pytest.mark.skipif(CHILD, reason="this is the child")
def test_lookback_cases_pass_in_the_stress_build():
    from streamvbyte_amd.build import STRESS_LIB
    assert os.path.exists(STRESS_LIB), "stress build missing: python -m streamvbyte_amd.build"
    env = dict(os.environ, SVB_LIB_PATH=STRESS_LIB, SVB_STRESS_CHILD="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu",
                          "--timeout", "300", "--timeout-method", "thread", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=900,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and " passed" in out.stdout, (out.stdout[-3000:], out.stderr[-2000:])


@pytest.fixture
def stressed():
    import streamvbyte_amd as svb
    lib = svb.load()
    assert lib.svbgpu_build_has(b"lookback_stress") == 1, "not the stress build"
    assert lib.svbgpu_set_option(b"debug_ablate", 64) == 0
    yield lib
    lib.svbgpu_set_option(b"debug_ablate", 0)
    lib.svbgpu_set_option(b"encode_pipeline", 1)


@pytest.mark.skipif(not CHILD, reason="runs in the child process, against the stress build")
@pytest.mark.parametrize("n", [(1 << 22) + 77, (1 << 24) + 5])
def test_lookback_kernels_are_exact_under_delays(oracle, stressed, n):
    import torch
    import streamvbyte_amd as svb
    rng = np.random.default_rng(n)
    codec = svb.DeviceCodec(n, "cuda:0")
    v = GENERATORS["width_mix"](rng, n)
    d_v = to_dev(torch, v)
    for variant, flags, prev, pipeline in ((V1234, 0, 0, 1), (V1234, 0, 0, 0), (V0124, svb.ZIGZAG, 0, 1),
                                           (V1234, svb.DELTA, 99, 1), (V1234, svb.DELTA, 99, 2)):
        stressed.svbgpu_set_option(b"encode_pipeline", pipeline)
        src = oracle.zigzag_encode(v.view(np.int32)) if flags & svb.ZIGZAG else v
        expect = oracle.encode(variant, src, bool(flags & svb.DELTA), prev)
        out, size = codec.encode(d_v, variant=variant, flags=flags, prev=prev)
        codec.check()
        assert int(size.item()) == expect.size
        assert np.array_equal(out[: expect.size].cpu().numpy(), expect), (variant, flags, pipeline)
        if flags & svb.DELTA:
            dec, used = codec.decode(dev_stream(torch, expect), n, variant=variant, flags=flags, prev=prev)
            codec.check()
            assert int(used.item()) == expect.size
            assert np.array_equal(to_np(dec, np.uint32), v)
    # the stress build has only 15 tags: twenty more calls on this (initialised) workspace wrap them
    small = GENERATORS["perf_c"](rng, 70001)
    want = oracle.encode(V1234, small)
    stressed.svbgpu_set_option(b"encode_pipeline", 1)
    for _ in range(20):
        out, size = codec.encode(to_dev(torch, small))
        assert int(size.item()) == want.size and np.array_equal(out[: want.size].cpu().numpy(), want)
    codec.check()
    # running sum (zigzag_delta_decode) and its inverse
    z = codec.zigzag(d_v, "delta_decode", prev=-5)
    codec.check()
    assert np.array_equal(to_np(z, np.int32), oracle.zigzag_delta_decode(v, -5))
    # decode fused with an intersection (hit counts travel through the look-back)
    post = GENERATORS["postings"](rng, n)
    enc = oracle.encode(V1234, post, True, 0)
    probe = np.unique(post[rng.integers(0, n, size=n // 16)])
    hits, nhits, _ = codec.decode_intersect(dev_stream(torch, enc), n, to_dev(torch, probe))
    codec.check()
    want = post[np.isin(post, probe)]
    assert int(nhits.item()) == want.size
    assert np.array_equal(to_np(hits[: want.size], np.uint32), want)
    # ... and with the probe compressed as well (svbgpu_intersect_streams: the same chain of hit counts)
    enc_b = oracle.encode(V1234, probe, True, 0)
    hits, nhits = codec.intersect_streams(dev_stream(torch, enc), n, dev_stream(torch, enc_b), probe.size)
    codec.check()
    assert int(nhits.item()) == want.size
    assert np.array_equal(to_np(hits[: want.size], np.uint32), want)
    # 64-bit values: the single-pass 1248 encoder chains its tiles the same way
    w = (rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
         >> (np.uint64(64) - (np.uint64(8) << rng.integers(0, 4, size=n).astype(np.uint64))))
    want64 = oracle.encode64(w)
    out, size = codec.encode64(torch.from_numpy(w.view(np.int64)).cuda())
    codec.check()
    assert int(size.item()) == want64.size
    assert np.array_equal(out[: want64.size].cpu().numpy(), want64)
