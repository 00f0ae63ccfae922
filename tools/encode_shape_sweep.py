#!/usr/bin/env python3
"""Which encoder by stream size, in the steady state (60 untimed calls first: the encoders take some 20 ms of
back-to-back calls to settle): pipelined k_encode_pipe, tile-at-a-time k_encode as a workgroup per tile with 8 and
with 16 waves, and as persistent workgroups; uniform widths (1234), gradients (0124 + zigzag), mixed gaps (delta)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import gen_width_mix, event_ms
dev = "cuda:0"
opt = svb.load().svbgpu_set_option
sizes = [int(x) for x in sys.argv[1:]] or [20, 22, 23, 24, 25, 26, 27, 28]
for name, variant, flags in (("1234", svb.V1234, 0), ("0124+zz", svb.V0124, svb.ZIGZAG), ("delta mixed", svb.V1234, svb.DELTA), ("delta dense", svb.V1234, svb.DELTA)):
    for lg in sizes:
        n = 1 << lg
        codec = svb.DeviceCodec(n, dev)
        g = torch.Generator(device=dev); g.manual_seed(3)
        if name == "delta dense":
            v = torch.cumsum(torch.randint(0, 16, (n,), generator=g, device=dev, dtype=torch.int64), 0).to(torch.int32)
        elif name == "delta mixed":
            r31 = torch.randint(0, 1 << 31, (n,), generator=g, device=dev, dtype=torch.int64)
            sh = torch.randint(0, 32, (n,), generator=g, device=dev, dtype=torch.int64)
            v = torch.cumsum((r31 >> sh) >> 8, 0) & 0xFFFFFFFF
            v = torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32)
            del r31, sh
        elif variant == svb.V1234:
            v = gen_width_mix(torch, n, 1, dev)
        else:
            v = torch.round(torch.randn(n, generator=g, device=dev) * 0.8).to(torch.int32)
        out = torch.empty(svb.streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device=dev)
        f = lambda: codec.encode(v, out=out, variant=variant, flags=flags)
        _, z = f()
        nbytes = int(z.item()) + 4.0 * n
        row = [f"{name:11s} 2^{lg}"]
        shapes = (("pipe", 2, 1, 8, 4), ("8w fresh", 0, 1, 8, 4), ("16w", 0, 1, 16, 4), ("8w persistent", 0, 0, 8, 4), ("4w fresh", 0, 1, 4, 4))
        if os.environ.get("SMALL_TILES"):
            shapes = (("8w fresh", 0, 1, 8, 4), ("8w x 2 wave tiles fresh", 0, 1, 8, 2), ("4w x 2 fresh", 0, 1, 4, 2), ("4w x 1 fresh", 0, 1, 4, 1))
        for label, pipe, fresh, waves, qw in shapes:
            opt(b"encode_pipeline", pipe); opt(b"encode_fresh", fresh); opt(b"encode_waves", waves); opt(b"encode_tile", qw)
            _, ms = event_ms(torch, f, 10, warmup=60)
            row.append(f"{label} {ms * 1e3:8.1f} us {nbytes / ms / 1e6 / 8000:.3f}")
        opt(b"encode_pipeline", 3); opt(b"encode_fresh", 1); opt(b"encode_waves", 8); opt(b"encode_tile", 4)
        codec.check()
        print("  ".join(row), flush=True)
        del codec, v, out
