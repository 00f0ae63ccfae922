#!/usr/bin/env python3
"""How long a call has to be repeated before its time settles: decode, the pipelined encoder and the tile-at-a-time
encoder (a workgroup per tile) on the 2^28-value workload, 24 groups of 10 back-to-back calls each, after an idle second."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import gen_width_mix
dev = "cuda:0"
n = 1 << 28
codec = svb.DeviceCodec(n, dev)
v = gen_width_mix(torch, n, 1, dev)
s, z = codec.encode(v)
s = s.clone()
out = torch.empty(n, dtype=torch.int32, device=dev)
enc = torch.empty(svb.streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device=dev)
opt = codec.lib.svbgpu_set_option


def curve(name, f):
    torch.cuda.synchronize()
    time.sleep(1.0)
    row = []
    for g in range(24):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            f()
        b.record()
        torch.cuda.synchronize()
        row.append(a.elapsed_time(b) / 10)
    print(f"{name:28s}" + " ".join(f"{x:.4f}" for x in row), flush=True)


curve("decode", lambda: codec.decode(s, n, out=out))
opt(b"encode_pipeline", 1)
curve("encode, pipelined", lambda: codec.encode(v, out=enc))
opt(b"encode_pipeline", 0)
curve("encode, workgroup per tile", lambda: codec.encode(v, out=enc))
opt(b"encode_fresh", 0)
curve("encode, persistent k_encode", lambda: codec.encode(v, out=enc))
opt(b"encode_fresh", 1)
curve("encode, workgroup per tile", lambda: codec.encode(v, out=enc))
opt(b"encode_pipeline", 1)
curve("encode, pipelined", lambda: codec.encode(v, out=enc))
curve("decode", lambda: codec.decode(s, n, out=out))
