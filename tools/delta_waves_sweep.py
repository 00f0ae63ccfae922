#!/usr/bin/env python3
"""Differential decode with 16 and with 8 waves per workgroup (option scan_waves) over stream sizes, dense
postings and mixed gaps, each timed as 10 back-to-back calls, median of 5."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import gen_width_mix
dev = "cuda:0"
for lg in (22, 23, 24, 25, 26, 27, 28):
    n = 1 << lg
    codec = svb.DeviceCodec(n, dev)
    g = torch.Generator(device=dev); g.manual_seed(7)
    post = torch.cumsum(torch.randint(0, 32, (n,), generator=g, device=dev, dtype=torch.int64), 0)
    post = torch.where(post >= (1 << 31), post - (1 << 32), post).to(torch.int32)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    row = [f"2^{lg}"]
    for name, vals in (("dense", post), ("mixed", gen_width_mix(torch, n, 1, dev))):
        s, z = codec.encode(vals, flags=svb.DELTA)
        s = s.clone()
        nbytes = int(z.item()) + 4.0 * n
        for waves in (16, 8):
            codec.lib.svbgpu_set_option(b"scan_waves", waves)
            f = lambda: codec.decode(s, n, out=out, flags=svb.DELTA)
            f(); f(); torch.cuda.synchronize()
            t = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f()
                e1.record(); torch.cuda.synchronize()
                t.append(e0.elapsed_time(e1) / 10)
            ms = float(np.median(t))
            row.append(f"{name} w{waves}: {ms:.4f} ms {nbytes / ms / 1e6 / 8000:.3f}")
        codec.lib.svbgpu_set_option(b"scan_waves", 16)
    print("  ".join(row), flush=True)
    del codec
