"""Shared helpers for the parity tests: synthetic inputs and device plumbing."""
import numpy as np


def width_mix(rng, n):
    """widths uniform over {1,2,3,4} bytes, value uniform inside its width class"""
    w = rng.integers(1, 5, size=n)
    lo = np.where(w == 1, 0, 1 << (8 * (w - 1))).astype(np.uint64)
    hi = (1 << (8 * w)).astype(np.uint64)
    return (lo + (rng.random(n) * (hi - lo)).astype(np.uint64)).astype(np.uint32)


def perf_c_like(rng, n):
    """tests/perf.c:52-53 of the reference: r31 >> (s & 31)"""
    r = rng.integers(0, 2**31, size=n, dtype=np.uint64)
    s = rng.integers(0, 32, size=n, dtype=np.uint64)
    return (r >> s).astype(np.uint32)


def sparse_small(rng, n):
    """many zeros, small magnitudes: the 0124 / gradient shape"""
    g = np.rint(rng.normal(0.0, 0.8, size=n)).astype(np.int64)
    out = rng.random(n) < 0.001
    g[out] = rng.integers(-(1 << 20), 1 << 20, size=int(out.sum()))
    return g.astype(np.int32)


def postings(rng, n, max_gap=32):
    return np.cumsum(rng.integers(0, max_gap, size=n, dtype=np.uint64), dtype=np.uint64).astype(np.uint32)


def full_range(rng, n):
    return rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)


GENERATORS = {"width_mix": width_mix, "perf_c": perf_c_like, "postings": postings, "full": full_range,
              "zeros": lambda rng, n: np.zeros(n, dtype=np.uint32),
              "ones": lambda rng, n: np.ones(n, dtype=np.uint32),
              "max": lambda rng, n: np.full(n, 0xFFFFFFFF, dtype=np.uint32),
              "sparse": lambda rng, n: sparse_small(rng, n).view(np.uint32)}


def to_dev(torch, a, device="cuda:0"):
    """numpy (uint8 / uint32 / int32) -> device tensor with the same bits"""
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).to(device)


def to_np(t, dtype):
    a = t.detach().cpu().numpy()
    return a.view(dtype)


def dev_stream(torch, stream_bytes, device="cuda:0", misalign=0):
    """device copy of a stream with STREAMVBYTE_PADDING readable bytes behind it,
    optionally shifted off alignment by `misalign` bytes"""
    buf = np.zeros(misalign + stream_bytes.size + 16, dtype=np.uint8)
    buf[misalign: misalign + stream_bytes.size] = stream_bytes
    t = torch.from_numpy(buf).to(device)
    return t[misalign:]
