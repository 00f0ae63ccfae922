#!/usr/bin/env python3
"""Diagnostic: per-tile phase timings of k_encode_pipe from in-kernel wall-clock stamps."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import gen_width_mix
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
tile_wt = int(sys.argv[2]) if len(sys.argv) > 2 else 16  # wave tiles per look-back tile
n = 1 << log2n
codec = svb.DeviceCodec(n, "cuda:0")
need = codec.ws_bytes
codec.lib.svbgpu_workspace_release(codec.ws.data_ptr()); codec._registered = False  # the stamps need room behind a workspace of their own
codec.ws = torch.zeros(need + 64 * (n // 4096 + 8), dtype=torch.uint8, device="cuda:0")
codec.ws_bytes = codec.ws.numel()
mode = sys.argv[3] if len(sys.argv) > 3 else "mix"
kw = {}
if mode == "zz0124":   # gradient shape: round(N(0, 0.8)), zigzag + 0124
    g = torch.Generator(device="cuda:0"); g.manual_seed(9)
    values = torch.round(torch.randn(n, generator=g, device="cuda:0") * 0.8).to(torch.int32)
    kw = dict(variant=svb.V0124, flags=svb.ZIGZAG)
elif mode == "dense":  # dense postings, delta coded
    g = torch.Generator(device="cuda:0"); g.manual_seed(7)
    post = torch.cumsum(torch.randint(0, 32, (n,), generator=g, device="cuda:0", dtype=torch.int64), 0)
    values = torch.where(post >= (1 << 31), post - (1 << 32), post).to(torch.int32)
    kw = dict(flags=svb.DELTA)
else:
    values = gen_width_mix(torch, n, 1, "cuda:0")
out = torch.empty(svb.streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device="cuda:0")
if len(sys.argv) > 4:   # encode_ticket_early value
    codec.lib.svbgpu_set_option(b"encode_ticket_early", int(sys.argv[4]))
for _ in range(3):
    codec.encode(values, out=out, **kw)
codec.lib.svbgpu_set_option(b"debug_ablate", 16)
codec.encode(values, out=out, **kw)
torch.cuda.synchronize()
codec.lib.svbgpu_set_option(b"debug_ablate", 0)
tiles = n // 1024 // tile_wt
st = codec.ws[need: need + 40 * tiles].cpu().numpy().view(np.uint64).reshape(tiles, 5)
rounds = (st[:, 4] & np.uint64(0xFF)).astype(np.int64)
wg_id = (st[:, 4] >> np.uint64(32)).astype(np.int64)
res_us = np.zeros(len(wg_id))
st = st.astype(np.float64)
t0 = st[:, 0].min()
start, published, resolved, done = [(st[:, i] - t0) / 100.0 for i in range(4)]
print(f"tiles {tiles}  kernel span {done.max():.1f} us")
for name, v in (("load+publish", published - start), ("publish->resolved", resolved - published), ("resolve itself", res_us),
                ("store", done - resolved), ("total", done - start)):
    print(f"{name:18s} mean {v.mean():7.2f}  p10 {np.percentile(v,10):7.2f}  p50 {np.percentile(v,50):7.2f}  p90 {np.percentile(v,90):7.2f}  max {v.max():7.2f} us")
# the ticket phase: per workgroup, from the end of a tile's store to the start of the load of the tile after next
wg = wg_id
gaps, cycles = [], []
for w in np.unique(wg)[:64]:
    idx = np.where(wg == w)[0]
    idx = idx[np.argsort(start[idx])]
    for a, c in zip(idx[:-2], idx[2:]):
        gaps.append(start[c] - done[a])
    cycles.extend(np.diff(start[idx]))
gaps, cycles = np.array(gaps), np.array(cycles)
print(f"ticket phase (store end -> next load start): mean {gaps.mean():.2f} p50 {np.percentile(gaps,50):.2f} p90 {np.percentile(gaps,90):.2f} us;  tile cycle per workgroup: mean {cycles.mean():.2f} us")
print("poll rounds: mean %.2f p50 %.0f p90 %.0f max %.0f" % (rounds.mean(), np.percentile(rounds, 50), np.percentile(rounds, 90), rounds.max()))
print("tiles started per us (steady state): %.1f" % ((tiles // 2) / (start[3 * tiles // 4] - start[tiles // 4])))
for lo in (0, tiles // 2, tiles - 6):
    print("tile start published resolved done rounds")
    for i in range(lo, min(lo + 6, tiles)):
        print(f"{i:6d} {start[i]:8.2f} {published[i]:8.2f} {resolved[i]:8.2f} {done[i]:8.2f} {rounds[i]:3d}")
