"""Everything of the multi-GPU path that one GPU can execute: the RCCL all-gather of per-stream sizes
with a process group of one rank (torch.distributed "nccl" = RCCL), and the C driver of
include/streamvbyte_multi.h (librccl loaded by the library itself, ncclCommInitAll with one device),
both on the device sizes of the batched encoder against the oracle's sizes list by list (the reference
codes a list at a time: src/streamvbyte_encode.c:113-137, src/streamvbytedelta_encode.c:65-76)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from oracle import V1234, V0124

pytestmark = pytest.mark.gpu


def _lists(rng, nlists, longest):
    counts = np.floor(np.exp(rng.uniform(0, np.log(longest), size=nlists))).astype(np.int64)
    counts[rng.integers(0, nlists, size=3)] = 0
    lists = [np.cumsum(rng.integers(0, 3000, size=int(c)), dtype=np.uint64).astype(np.uint32) for c in counts]
    return counts, lists


def _device_batch(torch, counts, lists, dev):
    voff = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    cap = (counts + 3) // 4 + 4 * counts + 16
    boff = np.concatenate([[0], np.cumsum(cap)]).astype(np.int64)
    vals = np.concatenate(lists + [np.zeros(0, dtype=np.uint32)]).astype(np.uint32)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    return dict(vals=t(vals.view(np.int32), torch.int32), voff=t(voff, torch.int64), boff=t(boff, torch.int64),
                counts=t(counts.astype(np.int32), torch.int32), bbuf=torch.zeros(int(boff[-1]) + 16, dtype=torch.uint8, device=dev),
                boff_np=boff)


def test_rccl_all_gather_of_sizes_with_one_rank(oracle):
    import torch
    import torch.distributed as dist
    import streamvbyte_amd as svb
    from streamvbyte_amd.sharding import partition_streams, all_gather_sizes
    rng = np.random.default_rng(2025)
    counts, lists = _lists(rng, 3000, 2000)
    b = _device_batch(torch, counts, lists, "cuda:0")
    codec = svb.DeviceCodec(1 << 20, "cuda:0")
    sizes = codec.encode_batch(b["vals"], b["voff"][:-1], b["bbuf"], b["boff"][:-1], b["counts"], None, flags=svb.DELTA)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        cut = partition_streams(counts, 1)
        sizes_all, offsets = all_gather_sizes(dist, sizes.to(torch.int64), cut)
        torch.cuda.synchronize()
        assert sizes_all.is_cuda
    finally:
        dist.destroy_process_group()
    expect = np.array([oracle.encode(V1234, l, True, 0).size for l in lists], dtype=np.int64)
    assert np.array_equal(sizes_all.cpu().numpy(), expect)
    assert np.array_equal(offsets.cpu().numpy(), np.cumsum(expect) - expect)


@pytest.mark.parametrize("variant,flags", [(V1234, 1), (V0124, 0)])
def test_c_multi_device_driver_with_one_device(oracle, variant, flags):
    import torch
    import streamvbyte_amd as svb
    from streamvbyte_amd._lib import MultiShard
    lib = svb.load()
    rng = np.random.default_rng(7 + variant)
    counts, lists = _lists(rng, 9001, 1500)   # not a multiple of 4096: the last block of the scan is ragged
    b = _device_batch(torch, counts, lists, "cuda:0")
    n = counts.size
    sizes_all = torch.full((n,), -1, dtype=torch.int64, device="cuda:0")
    offsets_all = torch.full((n + 1,), -1, dtype=torch.int64, device="cuda:0")
    back = torch.zeros_like(b["vals"])
    torch.cuda.synchronize()
    handle = C.c_void_p()
    dev = (C.c_int * 1)(0)
    rc = lib.svbmulti_create(dev, 1, C.byref(handle))
    assert rc == 0, (rc, lib.svbmulti_last_error())
    try:
        assert lib.svbmulti_ndev(handle) == 1 and lib.svbmulti_stream(handle, 0)
        sh = MultiShard(b["vals"].data_ptr(), b["voff"].data_ptr(), b["bbuf"].data_ptr(), b["boff"].data_ptr(),
                        b["counts"].data_ptr(), None, n, sizes_all.data_ptr(), offsets_all.data_ptr())
        for _ in range(2):   # the second call reuses the context's buffers
            rc = lib.svbmulti_encode_batch(handle, variant, flags, C.byref(sh))
            assert rc == 0, (rc, lib.svbmulti_last_error())
        sh.d_values = back.data_ptr()
        assert lib.svbmulti_decode_batch(handle, variant, flags, C.byref(sh)) == 0
        assert lib.svbmulti_synchronize(handle) == 0
    finally:
        lib.svbmulti_destroy(handle)
    expect = [oracle.encode(variant, l, bool(flags & 1), 0) for l in lists]
    esz = np.array([e.size for e in expect], dtype=np.int64)
    assert np.array_equal(sizes_all.cpu().numpy(), esz)
    assert np.array_equal(offsets_all.cpu().numpy(), np.concatenate([[0], np.cumsum(esz)]))
    got = b["bbuf"].cpu().numpy()
    for i in rng.integers(0, n, size=300):
        o = int(b["boff_np"][i])
        assert np.array_equal(got[o:o + expect[i].size], expect[i]), i
    assert np.array_equal(back.cpu().numpy(), b["vals"].cpu().numpy())
