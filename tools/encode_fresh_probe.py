#!/usr/bin/env python3
"""Tile-at-a-time encoder as persistent workgroups (ticket loop) against a fresh workgroup per tile
(debug_ablate 8), by tile shape, on the uniform-width workload (1234) and the gradients (0124 + zigzag)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import gen_width_mix, event_ms
dev = "cuda:0"
opt = svb.load().svbgpu_set_option
for name, lg, variant, flags in (("1234 uniform widths", 28, svb.V1234, 0), ("0124 + zigzag gradients", 26, svb.V0124, svb.ZIGZAG),
                                 ("1234 delta dense postings", 28, svb.V1234, svb.DELTA), ("1234 delta mixed gaps", 26, svb.V1234, svb.DELTA),
                                 ("1234 uniform widths", 24, svb.V1234, 0)):
    n = 1 << lg
    codec = svb.DeviceCodec(n, dev)
    if "dense" in name:
        g = torch.Generator(device=dev); g.manual_seed(3)
        v = torch.cumsum(torch.randint(0, 16, (n,), generator=g, device=dev, dtype=torch.int64), 0).to(torch.int32)
    elif "mixed" in name:
        g = torch.Generator(device=dev); g.manual_seed(3)
        r31 = torch.randint(0, 1 << 31, (n,), generator=g, device=dev, dtype=torch.int64)
        sh = torch.randint(0, 32, (n,), generator=g, device=dev, dtype=torch.int64)
        v = torch.cumsum((r31 >> sh) >> 8, 0) & 0xFFFFFFFF
        v = torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32)
    elif variant == svb.V1234:
        v = gen_width_mix(torch, n, 1, dev)
    else:
        g = torch.Generator(device=dev); g.manual_seed(3)
        v = torch.round(torch.randn(n, generator=g, device=dev) * 0.8).to(torch.int32)
    out = torch.empty(svb.streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device=dev)
    ref, z = codec.encode(v, variant=variant, flags=flags)
    size = int(z.item())
    ref = ref[:size].clone()
    nbytes = size + 4.0 * n
    f = lambda: codec.encode(v, out=out, variant=variant, flags=flags)
    _, ms = event_ms(torch, f, 10)
    print(f"{name}: default {ms:.4f} ms {nbytes / ms / 1e6 / 8000:.3f}")
    opt(b"encode_pipeline", 0)
    for waves in (4, 8, 16):
        opt(b"encode_waves", waves)
        for dbg in (0, 8):
            opt(b"encode_fresh", 1 if dbg else 0)
            _, ms = event_ms(torch, f, 10)
            _, z2 = f()
            ok = int(z2.item()) == size and torch.equal(out[:size], ref)
            codec.check()
            print(f"  k_encode waves={waves} {'fresh' if dbg else 'persistent'}: {ms:.4f} ms {nbytes / ms / 1e6 / 8000:.3f} exact={ok}", flush=True)
    opt(b"encode_fresh", 1); opt(b"encode_waves", 8); opt(b"encode_pipeline", 3)
    del codec, v, out
