#!/usr/bin/env python3
"""Streams per wavefront in the long-stream blocks of the chunked batch kernels (option batch_chunk),
on the benchmark's posting lists (dense gaps, delta coded):  python tools/batch_chunk_sweep.py [chunk ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import list_lengths, gen_posting_lists, event_ms
dev = "cuda:0"
lengths = list_lengths(1 << 20, 20241218)
counts, voff, vals = gen_posting_lists(torch, lengths, 1234, dev)
tot = int(voff[-1].item())
cap = (counts.to(torch.int64) + 3) // 4 + 4 * counts.to(torch.int64) + 16
boff = torch.zeros(len(lengths) + 1, dtype=torch.int64, device=dev)
boff[1:] = torch.cumsum(cap, 0)
bbuf = torch.empty(int(boff[-1].item()) + 16, dtype=torch.uint8, device=dev)
back = torch.empty_like(vals)
codec = svb.DeviceCodec(1 << 20, dev)
sizes = codec.encode_batch(vals, voff[:-1], bbuf, boff[:-1], counts, None, flags=svb.DELTA)
ab = 4.0 * tot + float(sizes.sum().item())
for k in [int(x) for x in sys.argv[1:]] or [-1, 4, 8, 16, 32, 64]:
    codec.lib.svbgpu_set_option(b"batch_chunk", k)
    te = event_ms(torch, lambda: codec.encode_batch(vals, voff[:-1], bbuf, boff[:-1], counts, None, flags=svb.DELTA, sizes=sizes), 5)[1]
    td = event_ms(torch, lambda: codec.decode_batch(bbuf, boff[:-1], back, voff[:-1], counts, None, flags=svb.DELTA), 5)[1]
    assert torch.equal(back, vals)
    print(f"batch_chunk {k:3d}: encode {te:6.3f} ms {ab / te / 1e6 / 8000:5.3f} of peak   decode {td:6.3f} ms {ab / td / 1e6 / 8000:5.3f} of peak")
codec.lib.svbgpu_set_option(b"batch_chunk", -1)
