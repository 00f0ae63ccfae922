#!/usr/bin/env python3
"""Differential encode, tile-at-a-time kernel (encode_pipeline 1 -> k_encode) against the pipelined one
(encode_pipeline 2 -> k_encode_pipe), dense postings (1-byte gaps) and mixed gaps, over stream sizes.
Fractions of 8 TB/s, 10 calls back to back, middle of 3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import event_ms
dev = "cuda:0"
opt = svb.load().svbgpu_set_option
for lg in (22, 24, 25, 26, 27, 28):
    n = 1 << lg
    g = torch.Generator(device=dev); g.manual_seed(5)
    codec = svb.DeviceCodec(n, dev)
    enc_out = torch.empty(svb.streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device=dev)
    row = [f"2^{lg}"]
    for name in ("dense", "mixed"):
        if name == "dense":
            gaps = torch.randint(0, 32, (n,), generator=g, device=dev, dtype=torch.int64)
        else:
            gaps = (torch.randint(0, 1 << 31, (n,), generator=g, device=dev, dtype=torch.int64) >>
                    torch.randint(0, 32, (n,), generator=g, device=dev, dtype=torch.int64)) >> 8
        post = torch.cumsum(gaps, 0) & 0xFFFFFFFF
        post = torch.where(post >= (1 << 31), post - (1 << 32), post).to(torch.int32)
        del gaps
        _, z = codec.encode(post, out=enc_out, flags=svb.DELTA)
        nbytes = int(z.item()) + 4.0 * n
        for pipe in (1, 2):
            opt(b"encode_pipeline", pipe)
            _, ms = event_ms(torch, lambda: codec.encode(post, out=enc_out, flags=svb.DELTA), 10)
            row.append(f"{name} p{pipe}: {ms:.4f} ms {nbytes / ms / 1e6 / 8000:.3f}")
        opt(b"encode_pipeline", 3)
        del post
    print("  ".join(row), flush=True)
    del codec, enc_out
