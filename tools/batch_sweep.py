#!/usr/bin/env python3
"""Batched encode / decode throughput against list length (all lists the same length)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
dev = "cuda:0"
tot = 1 << 27
codec = svb.DeviceCodec(1 << 20, dev)
g = torch.Generator(device=dev); g.manual_seed(1)
gaps = torch.randint(0, 1 << 12, (tot,), generator=g, device=dev, dtype=torch.int64)
lengths = (4, 16, 64, 256, 300, 512, 1000, 1024, 1100, 1536, 4096, 16384)
if len(sys.argv) > 1:  # tools/batch_sweep.py "4,16,64" [batch_chunk]
    lengths = tuple(int(x) for x in sys.argv[1].split(","))
if len(sys.argv) > 2:
    codec.lib.svbgpu_set_option(b"batch_chunk", int(sys.argv[2]))
    print("batch_chunk", sys.argv[2])
for L in lengths:
    nl = tot // L
    n = nl * L
    vals = gaps[:n].to(torch.int32).contiguous()
    counts = torch.full((nl,), L, dtype=torch.int32, device=dev)
    voff = torch.arange(nl, dtype=torch.int64, device=dev) * L
    cap = (L + 3) // 4 + 4 * L + 16
    boff = torch.arange(nl, dtype=torch.int64, device=dev) * cap
    bbuf = torch.empty(nl * cap + 16, dtype=torch.uint8, device=dev)
    back = torch.empty_like(vals)
    sizes = codec.encode_batch(vals, voff, bbuf, boff, counts, None, flags=0)
    dB = float(sizes.sum().item()) / n - 0.25
    def t(f):
        f(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return float(np.median(ts))
    te = t(lambda: codec.encode_batch(vals, voff, bbuf, boff, counts, None, flags=0, sizes=sizes))
    td = t(lambda: codec.decode_batch(bbuf, boff, back, voff, counts, None, flags=0))
    assert torch.equal(back, vals)
    ab = (4.25 + dB) * n
    print(f"L={L:8d} lists={nl:8d}  encode {te:7.3f} ms {ab/te/1e6/8000:5.3f} of peak   decode {td:7.3f} ms {ab/td/1e6/8000:5.3f} of peak")
