#!/usr/bin/env python3
"""Diagnostic: per-tile phase timings of k_encode_stream from in-kernel wall-clock stamps.

    python tools/stamps_encode_stream.py [log2n] [encode_pipeline mode] [waves per tile]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import gen_width_mix
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tile_wt = int(sys.argv[3]) if len(sys.argv) > 3 else 4
n = 1 << log2n
codec = svb.DeviceCodec(n, "cuda:0")
need = codec.ws_bytes
tiles = n // 1024 // tile_wt
codec.ws = torch.zeros(need + 40 * (tiles + 8), dtype=torch.uint8, device="cuda:0")
codec.ws_bytes = codec.ws.numel()
values = gen_width_mix(torch, n, 1, "cuda:0")
out = torch.empty(svb.streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device="cuda:0")
codec.lib.svbgpu_set_option(b"encode_pipeline", mode)
for _ in range(3):
    codec.encode(values, out=out)
codec.lib.svbgpu_set_option(b"debug_ablate", 16 | (int(sys.argv[4]) if len(sys.argv) > 4 else 0))
codec.encode(values, out=out)
torch.cuda.synchronize()
codec.lib.svbgpu_set_option(b"debug_ablate", 0)
st = codec.ws[need: need + 40 * tiles].cpu().numpy().view(np.uint64).reshape(tiles, 5)
rounds = (st[:, 4] & np.uint64(0xFF)).astype(np.int64)
tround = ((st[:, 4] >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int64)
ground = ((st[:, 4] >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
xcc = ((st[:, 4] >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
hwid = ((st[:, 4] >> np.uint64(40)) & np.uint64(0xFFFF)).astype(np.int64)
st = st.astype(np.float64)
t0 = st[:, 0].min()
start, classified, resolved, done = [(st[:, i] - t0) / 100.0 for i in range(4)]
print(f"mode {mode}: tiles {tiles}  kernel span {done.max():.1f} us")
for name, v in (("load+classify", classified - start), ("look-back", resolved - classified), ("store", done - resolved), ("total", done - start)):
    print(f"{name:14s} mean {v.mean():7.2f}  p10 {np.percentile(v,10):7.2f}  p50 {np.percentile(v,50):7.2f}  p90 {np.percentile(v,90):7.2f}  max {v.max():7.2f} us")
print("full rounds: mean %.2f p50 %.0f p90 %.0f max %.0f; watches mean %.2f, single-word polls mean %.2f" % (
    rounds.mean(), np.percentile(rounds, 50), np.percentile(rounds, 90), rounds.max(), tround.mean(), ground.mean()))
lead = (np.arange(tiles) & 15) == 15
print("leaders: rounds mean %.2f look-back mean %.2f us; others: rounds %.2f look-back %.2f us" % (
    rounds[lead].mean(), (resolved - classified)[lead].mean(), rounds[~lead].mean(), (resolved - classified)[~lead].mean()))
print("tiles started per us (middle half): %.1f" % ((tiles // 2) / (start[3 * tiles // 4] - start[tiles // 4])))
order = np.argsort(start, kind="stable")
disp = np.abs(order - np.arange(tiles))
print("start order vs tile index: mean |rank - index| %.1f, max %d; tiles that started before a lower-indexed one by > 64: %d" % (
    disp.mean(), disp.max(), int((disp > 64).sum())))
print("xcc of tiles 0..15:", xcc[:16].tolist(), " tile%8 -> xcc consistent:", all(len(set(xcc[r::8].tolist())) == 1 for r in range(8)))
late = classified - start
print("tiles in flight (started, not done) at mid-kernel:", int(((start < done.max() / 2) & (done > done.max() / 2)).sum()))
for lo in (0, tiles // 2, tiles - 4):
    print("tile start classified resolved done rounds(t,g) xcc")
    for i in range(lo, min(lo + 20 if lo else 4, tiles)):
        print(f"{i:6d} {start[i]:8.2f} {classified[i]:8.2f} {resolved[i]:8.2f} {done[i]:8.2f} {rounds[i]:3d} ({tround[i]},{ground[i]}) {xcc[i]}")

cu = hwid >> 8   # cu_id, sh_id, se_id
for x in range(8):
    m = xcc == x
    cus = np.unique(cu[m])
    per = np.bincount(cu[m])[cus]
    print(f"xcc {x}: tiles {int(m.sum())}  distinct CU ids {len(cus)}  tiles per CU min {per.min()} max {per.max()}  mean start {start[m].mean():.1f} last done {done[m].max():.1f}  mean life {(done-start)[m].mean():.2f}")
