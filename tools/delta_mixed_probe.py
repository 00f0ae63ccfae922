#!/usr/bin/env python3
"""The benchmark's mixed-gap differential workload (2^26 values, 1-3 byte gaps, wrapping sums) under the
options that matter for it: encoder kernel (tile-at-a-time / pipelined), decoder tile (waves per
workgroup), two-pass decoder, cache policy.  Fractions of 8 TB/s, 10 calls back to back, middle of 3."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import event_ms
dev = "cuda:0"
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << log2n
g = torch.Generator(device=dev); g.manual_seed(123)
r31 = torch.randint(0, 1 << 31, (n,), generator=g, device=dev, dtype=torch.int64)
sh = torch.randint(0, 32, (n,), generator=g, device=dev, dtype=torch.int64)
post = (torch.cumsum((r31 >> sh) >> 8, 0) + 0xDEADBEEF) & 0xFFFFFFFF
post = torch.where(post >= (1 << 31), post - (1 << 32), post).to(torch.int32)
del r31, sh
codec = svb.DeviceCodec(n, dev)
s, z = codec.encode(post, flags=svb.DELTA, prev=0xDEADBEEF)
s = s.clone()
nbytes = int(z.item()) + 4.0 * n
out = torch.empty(n, dtype=torch.int32, device=dev)
enc_out = torch.empty(svb.streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device=dev)
opt = codec.lib.svbgpu_set_option
def show(name, f):
    _, ms = event_ms(torch, f, 10)
    print(f"{name:34s} {ms:8.4f} ms  {nbytes / ms / 1e6 / 8000:6.3f} of 8 TB/s", flush=True)
enc = lambda: codec.encode(post, out=enc_out, flags=svb.DELTA, prev=0xDEADBEEF)
dec = lambda: codec.decode(s, n, out=out, flags=svb.DELTA, prev=0xDEADBEEF)
for nt in (-1, 0, 3):
    opt(b"nt_force", nt)
    for pipe in (1, 2):
        opt(b"encode_pipeline", pipe)
        show(f"encode pipeline={pipe} nt_force={nt}", enc)
    opt(b"encode_pipeline", 3)
    for w in (16, 8, 4):
        opt(b"scan_waves", w)
        show(f"decode scan_waves={w} nt_force={nt}", dec)
    opt(b"scan_waves", 16)
opt(b"nt_force", -1)
opt(b"delta_decode_lookback", 0)
show("decode two passes", dec)
opt(b"delta_decode_lookback", 1)
codec.check()
assert torch.equal(out, post)
