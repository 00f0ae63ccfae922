#!/usr/bin/env python3
"""Plain decoder and zigzag map at several sizes, a workgroup per unit (option decode_grid_per_cu = 0,
the default) against persistent grids of 7 and 256 workgroups per CU.  python tools/grid_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import streamvbyte_amd as svb
dev = "cuda:0"
def t(f):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for log2n in (20, 24, 30):
    n = 1 << log2n
    g = torch.Generator(device=dev); g.manual_seed(3)
    v = torch.empty(n, dtype=torch.int32, device=dev)
    for s0 in range(0, n, 1 << 27):
        m = min(1 << 27, n - s0)
        w = torch.randint(0, 4, (m,), generator=g, device=dev, dtype=torch.int64)
        v[s0:s0 + m] = (torch.randint(0, 1 << 32, (m,), generator=g, device=dev, dtype=torch.int64) >> (8 * (3 - w))).to(torch.int32)
        del w
    codec = svb.DeviceCodec(n)
    s, sz = codec.encode(v)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    for pc in (0, 7, 256):
        codec.lib.svbgpu_set_option(b"decode_grid_per_cu", pc)
        codec.decode_plan(s, n)
        ms = t(lambda: codec.decode_run(s, n, out))
        assert torch.equal(out, v)
        print(f"n=2^{log2n} decode_grid_per_cu {pc:5d}: decode_run {ms:.4f} ms  ({6.75 * n / ms / 1e6 / 8000:.3f} of peak)")
    codec.lib.svbgpu_set_option(b"decode_grid_per_cu", 0)
    mz = t(lambda: codec.zigzag(v, "encode", out=out))
    print(f"n=2^{log2n} zigzag_encode {mz:.4f} ms ({8.0 * n / mz / 1e6 / 8000:.3f} of peak)")
    del codec, s, out, v
