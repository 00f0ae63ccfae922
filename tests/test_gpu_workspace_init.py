"""Calls on an initialised workspace (svbgpu_workspace_init) skip their clearing launch: look-back words
carry a tag per call, two ticket counters alternate, the validator's accumulator is left at zero.  What an
earlier call left in the workspace -- a longer stream's words, another variant's, a decoder's -- must read as
empty to every later call.  Every result against the oracle (the reference codes each stream from scratch:
src/streamvbyte_encode.c:113-137)."""
import numpy as np
import pytest

from oracle import V1234, V0124
from util import GENERATORS, dev_stream, to_dev, to_np

pytestmark = pytest.mark.gpu


def test_many_calls_of_changing_size_on_one_initialised_workspace(oracle):
    import torch
    import streamvbyte_amd as svb
    codec = svb.DeviceCodec(1 << 22, "cuda:0")     # initialises its workspace
    assert codec._registered
    rng = np.random.default_rng(31)
    sizes = [(1 << 22), 70001, (1 << 21) + 5, 9, 1 << 20, 40000, (1 << 22) - 3, 16385, 300, (1 << 21)]
    k = 0
    for rep in range(4):
        for n in sizes:
            kind = ("width_mix", "postings", "perf_c", "sparse")[k % 4]
            variant, flags, prev = ((V1234, 0, 0), (V0124, svb.ZIGZAG, 0), (V1234, svb.DELTA, 1234567),
                                    (V0124, 0, 0), (V1234, svb.DELTA | svb.ZIGZAG, 5))[k % 5]
            k += 1
            v = GENERATORS[kind](rng, n)
            src = v
            if flags & svb.DELTA:
                src = v - np.concatenate([[np.uint32(prev)], v[:-1]]).astype(np.uint32)
            if flags & svb.ZIGZAG:
                src = oracle.zigzag_encode(src.view(np.int32))
            expect = oracle.encode(variant, src)
            out, size = codec.encode(to_dev(torch, v), variant=variant, flags=flags, prev=prev)
            assert int(size.item()) == expect.size, (rep, n, variant, flags)
            assert np.array_equal(out[: expect.size].cpu().numpy(), expect), (rep, n, variant, flags)
            s = dev_stream(torch, expect)
            assert int(codec.validate(s, expect.size, n, variant=variant).item()) & 1 == 1
            assert int(codec.validate(s, expect.size - 1, n, variant=variant).item()) & 1 == 0
            if k % 3 == 0:   # a look-back decoder and a running sum in between: they clear or tag for themselves
                dec, used = codec.decode(s, n, variant=variant, flags=flags, prev=prev)
                assert np.array_equal(to_np(dec, np.uint32), v)
                z = codec.zigzag(to_dev(torch, v), "delta_decode", prev=7)
                assert np.array_equal(to_np(z, np.int32), oracle.zigzag_delta_decode(v, 7))
    codec.check()


def test_initialised_and_plain_workspace_give_the_same_bytes(oracle):
    import torch
    import streamvbyte_amd as svb
    lib = svb.load()
    n = (1 << 21) + 77
    v = GENERATORS["width_mix"](np.random.default_rng(3), n)
    expect = oracle.encode(V1234, v)
    d_v = to_dev(torch, v)
    plain = svb.DeviceCodec(n, "cuda:0", zero=False)      # not initialised: every call clears
    assert not plain._registered
    plain.ws.fill_(0xA5)                                   # garbage is fine there
    init = svb.DeviceCodec(n, "cuda:0")
    for codec in (plain, init, plain, init):
        out, size = codec.encode(d_v)
        assert int(size.item()) == expect.size and np.array_equal(out[: expect.size].cpu().numpy(), expect)
    # releasing and initialising again starts over
    lib.svbgpu_workspace_release(init.ws.data_ptr())
    init.ws.fill_(0x5A)
    assert lib.svbgpu_workspace_init(init.ws.data_ptr(), init.ws_bytes, torch.cuda.current_stream().cuda_stream) == 0
    out, size = init.encode(d_v)
    assert int(size.item()) == expect.size and np.array_equal(out[: expect.size].cpu().numpy(), expect)
    assert lib.svbgpu_workspace_init(None, 1024, None) == -3


def test_graph_replays_between_tagged_calls_on_one_workspace(oracle):
    """A call recorded into a HIP graph keeps its clearing launch (a replay would repeat a tag) and ends by
    putting the ticket counters back; replays and ordinary calls on the same initialised workspace may
    alternate in any order."""
    import torch
    import streamvbyte_amd as svb
    n = (1 << 20) + 17
    codec = svb.DeviceCodec(n, "cuda:0")
    rng = np.random.default_rng(8)
    v = GENERATORS["width_mix"](rng, n)
    d_v = to_dev(torch, v)
    expect = oracle.encode(V1234, v)
    expect0124 = oracle.encode(V0124, v)
    buf = torch.zeros(5 * n + 64, dtype=torch.uint8, device="cuda:0")
    for _ in range(3):   # tagged calls first, so that the workspace holds tagged words and a used ticket
        out, size = codec.encode(d_v)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            _, gsize = codec.encode(d_v, out=buf)
    for i in range(6):
        graph.replay()
        torch.cuda.synchronize()
        assert int(gsize.item()) == expect.size and np.array_equal(buf[: expect.size].cpu().numpy(), expect), i
        out, size = codec.encode(d_v, variant=V0124)          # an ordinary (tagged) call right behind a replay
        assert int(size.item()) == expect0124.size
        assert np.array_equal(out[: expect0124.size].cpu().numpy(), expect0124), i
        if i % 2:
            out, size = codec.encode(d_v, variant=V0124)      # ... and sometimes two, so both ticket counters take turns
            assert np.array_equal(out[: expect0124.size].cpu().numpy(), expect0124), i
    codec.check()
