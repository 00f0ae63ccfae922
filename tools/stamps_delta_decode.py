#!/usr/bin/env python3
"""Diagnostic: where a tile of k_delta_decode (one workgroup of 16 waves, 64 wave tiles) spends its life,
from in-kernel wall-clock stamps:  python tools/stamps_delta_decode.py [log2n] [dense|mix] [waves per workgroup]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import gen_width_mix
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
mode = sys.argv[2] if len(sys.argv) > 2 else "dense"
n = 1 << log2n
dev = "cuda:0"
codec = svb.DeviceCodec(n, dev)
need = codec.ws_bytes
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 16   # waves per workgroup: 16 or 8 (option scan_waves)
tiles = (n // 1024 + 4 * waves - 1) // (4 * waves)
codec.lib.svbgpu_workspace_release(codec.ws.data_ptr()); codec._registered = False  # the stamps need room behind a workspace of their own
codec.ws = torch.zeros(need + 64 * (n // 1024 // 16 + 8), dtype=torch.uint8, device=dev)
codec.ws_bytes = codec.ws.numel()
if mode == "dense":
    g = torch.Generator(device=dev); g.manual_seed(7)
    post = torch.cumsum(torch.randint(0, 32, (n,), generator=g, device=dev, dtype=torch.int64), 0)
    values = torch.where(post >= (1 << 31), post - (1 << 32), post).to(torch.int32)
else:
    values = gen_width_mix(torch, n, 1, dev)
stream, _ = codec.encode(values, flags=svb.DELTA)
stream = stream.clone()
out = torch.empty(n, dtype=torch.int32, device=dev)
codec.lib.svbgpu_set_option(b"scan_waves", waves)
for _ in range(3):
    codec.decode(stream, n, out=out, flags=svb.DELTA)
codec.lib.svbgpu_set_option(b"debug_ablate", 16)
codec.decode(stream, n, out=out, flags=svb.DELTA)
torch.cuda.synchronize()
codec.lib.svbgpu_set_option(b"debug_ablate", 0)
st = codec.ws[need: need + 64 * tiles].cpu().numpy().view(np.uint64).reshape(tiles, 8).astype(np.float64)
t0 = st[:, 0].min()
T = (st[:, :6] - t0) / 100.0
print(f"tiles {tiles}  kernel span {T[:, 5].max():.1f} us")
names = ["ticket", "keys + offsets in", "decoded + summed", "prefix (barriers + look-back)", "stored", "whole tile"]
spans = [T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2], T[:, 4] - T[:, 3], T[:, 5] - T[:, 4], T[:, 5] - T[:, 0]]
for nm, v in zip(names, spans):
    print(f"{nm:30s} mean {v.mean():6.2f}  p10 {np.percentile(v, 10):6.2f}  p50 {np.percentile(v, 50):6.2f}  p90 {np.percentile(v, 90):6.2f} us")
starts = np.sort(T[:, 0])
print("tiles started per us (steady state): %.2f" % ((tiles // 2) / (starts[3 * tiles // 4] - starts[tiles // 4])))
