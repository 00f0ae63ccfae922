"""Multi-rank layer on CPU: stream partitioning and the all-gather of sizes
(gloo, world_size 2 and 3).  The per-stream sizes come from the CPU oracle here;
on GPUs they come from svbgpu_encode_batch."""
import os
import socket

import numpy as np
import pytest

from streamvbyte_amd.sharding import partition_streams


def test_partition_is_contiguous_and_balanced():
    rng = np.random.default_rng(0)
    counts = np.floor(np.exp(rng.uniform(0, np.log(4096), size=5000))).astype(np.int64)
    for world in (1, 2, 3, 4, 8):
        cut = partition_streams(counts, world)
        assert cut[0] == 0 and cut[-1] == counts.size and np.all(np.diff(cut) >= 0)
        loads = [counts[cut[r]:cut[r + 1]].sum() for r in range(world)]
        assert max(loads) - min(loads) <= 2 * 4096
    # degenerate cases: fewer streams than ranks, empty batch
    assert partition_streams([5], 4).tolist()[-1] == 1
    assert partition_streams([], 2).tolist() == [0, 0, 0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    from oracle import Oracle
    from streamvbyte_amd.sharding import partition_streams, all_gather_sizes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(42)  # same batch on every rank
    counts = np.floor(np.exp(rng.uniform(0, np.log(300), size=101))).astype(np.int64)
    lists = [np.cumsum(rng.integers(0, 500, size=int(c)), dtype=np.uint64).astype(np.uint32) for c in counts]
    cut = partition_streams(counts, world)
    orc = Oracle()
    mine = [orc.encode(0, lists[i], True, 0).size for i in range(int(cut[rank]), int(cut[rank + 1]))]
    sizes_all, offsets = all_gather_sizes(dist, torch.tensor(mine, dtype=torch.int64), cut)
    expect = np.array([orc.encode(0, l, True, 0).size for l in lists], dtype=np.int64)
    ok = np.array_equal(sizes_all.numpy(), expect) and np.array_equal(
        offsets.numpy(), np.concatenate([[0], np.cumsum(expect)[:-1]]))
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_all_gather_sizes_gloo(world):
    import torch.multiprocessing as mp
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world)) and len(out) == world


def test_c_partition_matches_the_python_one():
    """svbgpu_partition_streams (include/streamvbyte_multi.h) is the same cut as partition_streams."""
    import ctypes as C
    import streamvbyte_amd as svb
    lib = svb.load()
    rng = np.random.default_rng(5)
    cases = [np.floor(np.exp(rng.uniform(0, np.log(4096), size=s))).astype(np.uint32) for s in (0, 1, 7, 5000)]
    cases.append(np.zeros(9, dtype=np.uint32))
    cases.append(np.full(11, 2**32 - 1, dtype=np.uint32))  # totals beyond 32 bits
    for counts in cases:
        for world in (1, 2, 3, 8, 64):
            cut = np.zeros(world + 1, dtype=np.uint64)
            rc = lib.svbgpu_partition_streams(counts.ctypes.data_as(C.c_void_p), counts.size, world,
                                              cut.ctypes.data_as(C.c_void_p))
            assert rc == 0
            assert np.array_equal(cut.astype(np.int64), partition_streams(counts, world)), (counts.size, world)
    assert lib.svbgpu_partition_streams(None, 0, 0, None) == -1


def test_multi_device_context_needs_devices():
    """Without a GPU svbmulti_create fails with an error code (no RCCL is loaded, nothing aborts)."""
    import ctypes as C
    import torch
    import streamvbyte_amd as svb
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu tests")
    lib = svb.load()
    handle = C.c_void_p()
    dev = (C.c_int * 1)(0)
    assert lib.svbmulti_create(dev, 1, C.byref(handle)) in (-2, -4)
    assert not handle.value
    assert lib.svbmulti_create(dev, 0, C.byref(handle)) == -1
    assert lib.svbmulti_ndev(None) == 0
    lib.svbmulti_destroy(None)
