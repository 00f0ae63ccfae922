"""The reference's unit sweeps at full density (tests/unit.c:90-185 basictests,
:187-252 aqrittests there): every length 0..128 and 256..4096, every gap 3^0..3^18
(3^0..3^12 for the differential codec), and all 65 536 zero/non-zero byte patterns
of four values -- each sweep as ONE batched launch per direction and variant, every
stream's bytes, size, validity and decoded values against the oracle.  (The
host-pointer versions in test_gpu_dropin.py thin these sweeps out, a call at a time.)"""
import ctypes as C

import numpy as np
import pytest

from oracle import V1234, V0124

pytestmark = pytest.mark.gpu

LENGTHS = list(range(0, 129)) + [256, 512, 1024, 2048, 4096]


@pytest.fixture(scope="module")
def env():
    import torch
    import streamvbyte_amd as svb
    assert torch.cuda.is_available()
    return torch, svb, svb.DeviceCodec(max_count=1 << 20)


def _oracle_streams(oracle, variant, lists, delta, prev=0):
    """per-list oracle streams laid out in worst-case slots shifted by 3 bytes (misaligned on purpose,
    as the reference's test does): returns (buffer, byte_off, sizes)"""
    counts = np.array([l.size for l in lists], dtype=np.int64)
    cap = (counts + 3) // 4 + 4 * counts + 16
    boff = np.zeros(len(lists), dtype=np.int64)
    boff[1:] = np.cumsum(cap)[:-1]
    boff += 3
    buf = np.zeros(int(cap.sum()) + 3 + 16, dtype=np.uint8)
    sizes = np.zeros(len(lists), dtype=np.int64)
    u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    for i, l in enumerate(lists):
        if l.size:
            sizes[i] = oracle.lib.svbo_encode(variant, l.ctypes.data_as(u32p), l.size,
                                              C.cast(buf.ctypes.data + int(boff[i]), u8p), int(delta), prev)
    return buf, boff, sizes


def _check_batch(env, oracle, variant, flags, lists):
    torch, svb, codec = env
    delta = bool(flags & svb.DELTA)
    counts = np.array([l.size for l in lists], dtype=np.int32)
    voff = np.zeros(len(lists), dtype=np.int64)
    voff[1:] = np.cumsum(counts.astype(np.int64))[:-1]
    vals = np.concatenate([l for l in lists if l.size] or [np.zeros(0, dtype=np.uint32)])
    expect, boff, exp_sizes = _oracle_streams(oracle, variant, lists, delta)
    d_vals = torch.from_numpy(vals.view(np.int32)).cuda()
    d_counts, d_voff, d_boff = (torch.from_numpy(a).cuda() for a in (counts, voff, boff))
    d_bytes = torch.full((expect.size,), 0xAB, dtype=torch.uint8, device="cuda:0")
    sizes = codec.encode_batch(d_vals, d_voff, d_bytes, d_boff, d_counts, None, variant=variant, flags=flags)
    torch.cuda.synchronize()
    assert np.array_equal(sizes.cpu().numpy(), exp_sizes), "sizes differ from the oracle's"
    got = d_bytes.cpu().numpy()
    for i in range(len(lists)):  # every byte of every stream, and nothing written before a stream's slot
        lo, n = int(boff[i]), int(exp_sizes[i])
        assert np.array_equal(got[lo: lo + n], expect[lo: lo + n]), (i, lists[i].size)
    assert np.all(got[:3] == 0xAB)
    # validity of the oracle's streams, list by list, and of each stream cut short by one byte
    d_expect = torch.from_numpy(expect).cuda()
    d_sizes = torch.from_numpy(exp_sizes).cuda()
    ok = codec.validate_batch(d_expect, d_boff, d_sizes, d_counts, variant=variant)
    assert bool((ok == 1).all())
    short = torch.clamp(d_sizes - 1, min=0)
    ok_short = codec.validate_batch(d_expect, d_boff, short, d_counts, variant=variant).cpu().numpy()
    assert np.array_equal(ok_short == 1, exp_sizes == 0)  # only the empty streams stay valid
    # decode the ORACLE's streams
    back = torch.full_like(d_vals, -1)
    used = codec.decode_batch(d_expect, d_boff, back, d_voff, d_counts, None, variant=variant, flags=flags)
    torch.cuda.synchronize()
    assert np.array_equal(used.cpu().numpy(), exp_sizes)
    assert torch.equal(back, d_vals)


@pytest.mark.parametrize("variant", [V1234, V0124])
def test_basictests_every_length_every_gap(env, oracle, variant):
    rng = np.random.default_rng(99)
    lists = []
    for length in LENGTHS:
        gap = 1
        while gap <= 387420489:  # 3^18
            jitter = rng.integers(0, 8, size=max(length, 1), dtype=np.uint64)
            steps = np.full(max(length, 1), gap - 1, dtype=np.uint64) + jitter
            steps[0] = jitter[0]
            lists.append(np.cumsum(steps, dtype=np.uint64).astype(np.uint32)[:length])  # wraps like the C code
            gap *= 3
    _check_batch(env, oracle, variant, 0, lists)


def test_basictests_differential(env, oracle):
    torch, svb, codec = env
    lists = []
    for length in LENGTHS:
        gap = 1
        while gap <= 531441:  # 3^12
            lists.append((np.arange(length, dtype=np.uint64) * gap).astype(np.uint32))
            gap *= 3
    _check_batch(env, oracle, V1234, svb.DELTA, lists)


@pytest.mark.parametrize("variant", [V1234, V0124])
def test_aqrit_all_65536_byte_patterns(env, oracle, variant):
    idx = np.arange(65536, dtype=np.uint32)
    quads = np.zeros((65536, 4), dtype=np.uint32)
    for b in range(16):
        quads[:, b // 4] |= ((idx >> b) & 1) << (8 * (b % 4))
    _check_batch(env, oracle, variant, 0, [quads[i] for i in range(65536)])
