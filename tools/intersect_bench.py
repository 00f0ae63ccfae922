#!/usr/bin/env python3
"""svbgpu_decode_intersect against "decode, then look the values up": dense postings
(gaps uniform 0..31) intersected with sorted arrays of several densities.

    python tools/intersect_bench.py [--log2n 28]
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=28)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
n = 1 << a.log2n
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(11)
gaps = torch.randint(0, 32, (n,), generator=g, device=dev, dtype=torch.int64)
v = torch.cumsum(gaps, 0).to(torch.int32)          # ascending, below 2^32 for n <= 2^28
top = int(gaps.sum().item())
del gaps
codec = svb.DeviceCodec(n)
stream, size = codec.encode(v, variant=svb.V1234, flags=svb.DELTA)
size = int(size.item())
out = torch.empty(n, dtype=torch.int32, device=dev)
dec = torch.empty(n, dtype=torch.int32, device=dev)


def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


t_dec = timed(lambda: codec.decode(stream, n, out=dec, variant=svb.V1234, flags=svb.DELTA))
print(f"n = 2^{a.log2n}, stream {size / n:.2f} B/value; plain delta decode {t_dec:.3f} ms")
for label, m in (("1/64 as dense", n // 64), ("1/8 as dense", n // 8), ("as dense", n), ("4x as dense", min(4 * n, 1 << 30))):
    b = torch.randint(0, top, (m,), generator=g, device=dev, dtype=torch.int64)
    b = torch.unique(b).to(torch.int32)            # values below 2^31 sort the same as int32
    if top >= 1 << 31:
        b = torch.sort(b.to(torch.int64) & 0xFFFFFFFF)[0].to(torch.int32)
    t = timed(lambda: codec.decode_intersect(stream, n, b, out=out, variant=svb.V1234))
    _, n_out, used = codec.decode_intersect(stream, n, b, out=out, variant=svb.V1234)
    codec.check()
    k = int(n_out.item())
    # check against torch: decode, then membership by searchsorted on unsigned keys
    codec.decode(stream, n, out=dec, variant=svb.V1234, flags=svb.DELTA)
    du, bu = dec.to(torch.int64) & 0xFFFFFFFF, b.to(torch.int64) & 0xFFFFFFFF
    pos = torch.searchsorted(bu, du).clamp_(max=bu.numel() - 1)
    ref = dec[bu[pos] == du]
    assert ref.numel() == k and torch.equal(ref, out[:k]), (label, ref.numel(), k)
    algo = size + 4 * b.numel() + 4 * k            # stream once, the array once, the hits
    print(f"  sorted array {label:13s} ({b.numel():>10d} entries): {t:.3f} ms, {k:>10d} hits, "
          f"{n / t / 1e6:.1f} Gvalues/s, {algo / t / 1e6:.0f} GB/s of stream+array+hits")
    del b, du, bu, pos, ref

# ---- both sides compressed: svbgpu_intersect_streams against "decode B, then svbgpu_decode_intersect" ----
print("stream x stream (svbgpu_intersect_streams); B = a delta-coded ascending stream")
for label, m in (("1/64 as dense", n // 64), ("1/8 as dense", n // 8), ("as dense", n), ("4x as dense", min(4 * n, 1 << 30))):
    b = torch.randint(0, min(top, (1 << 31) - 1), (m,), generator=g, device=dev, dtype=torch.int64)
    b = torch.unique(b).to(torch.int32)
    nb = b.numel()
    cb = svb.DeviceCodec(nb)
    sb, size_b = cb.encode(b, variant=svb.V1234, flags=svb.DELTA)
    size_b = int(size_b.item())
    t = timed(lambda: codec.intersect_streams(stream, n, sb, nb, out=out))
    _, n_out = codec.intersect_streams(stream, n, sb, nb, out=out)
    k = int(n_out.item())
    decb = torch.empty(nb, dtype=torch.int32, device=dev)

    def two_step():
        cb.decode(sb, nb, out=decb, variant=svb.V1234, flags=svb.DELTA)
        codec.decode_intersect(stream, n, decb, out=dec, variant=svb.V1234)
    t2 = timed(two_step)
    _, n_ref, _ = codec.decode_intersect(stream, n, decb, out=dec, variant=svb.V1234)
    assert int(n_ref.item()) == k and torch.equal(dec[:k], out[:k]), (label, int(n_ref.item()), k)
    algo = size + size_b + 4 * k
    print(f"  B {label:13s} ({nb:>10d} values, {size_b / nb:.2f} B/value): {t:.3f} ms, {k:>10d} hits, "
          f"{(n + nb) / t / 1e6:.1f} Gvalues/s of both streams, {algo / t / 1e6:.0f} GB/s of streams+hits; "
          f"decode B then decode_intersect: {t2:.3f} ms")
    del b, sb, decb, cb
