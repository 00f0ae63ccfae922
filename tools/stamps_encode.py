#!/usr/bin/env python3
"""Diagnostic: per-tile phase timings of k_encode from in-kernel wall-clock stamps."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import gen_width_mix
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
qw = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1 << log2n
codec = svb.DeviceCodec(n, "cuda:0")
# over-allocate the workspace so the stamps fit behind it
need = codec.ws_bytes
codec.lib.svbgpu_workspace_release(codec.ws.data_ptr()); codec._registered = False  # the stamps need room behind a workspace of their own
codec.ws = torch.zeros(need + 64 * (n // 4096 + 8), dtype=torch.uint8, device="cuda:0")
codec.ws_bytes = codec.ws.numel()
values = gen_width_mix(torch, n, 1, "cuda:0")
out = torch.empty(svb.streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device="cuda:0")
codec.lib.svbgpu_set_option(b"encode_tile", qw)
for _ in range(3):
    codec.encode(values, out=out)
codec.lib.svbgpu_set_option(b"debug_ablate", 16)
codec.encode(values, out=out)
torch.cuda.synchronize()
codec.lib.svbgpu_set_option(b"debug_ablate", 0)
tiles = (n // 1024 + 4 * qw - 1) // (4 * qw)
st = codec.ws[need: need + 40 * tiles].cpu().numpy().view(np.uint64).reshape(tiles, 5).astype(np.float64)
t0 = st[:, 0].min()
start, loaded, resolved, done = [(st[:, i] - t0) / 100.0 for i in range(4)]  # microseconds
print(f"tiles {tiles}  kernel span {done.max():.1f} us")
for name, v in (("load+classify", loaded - start), ("lookback", resolved - loaded), ("pass2", done - resolved), ("total", done - start)):
    print(f"{name:14s} mean {v.mean():7.2f}  p10 {np.percentile(v,10):7.2f}  p50 {np.percentile(v,50):7.2f}  p90 {np.percentile(v,90):7.2f}  max {v.max():7.2f} us")
# how long after the predecessor resolved did this tile resolve, and after own publish?
raw = st[:, 4].astype(np.int64)
r_tile, r_group = (raw >> 8) & 0xFF, (raw >> 16) & 0xFF
st[:, 4] = raw & 0xFF
pos = np.arange(tiles) & 15
lb = resolved - loaded
print("position in group: look-back us / rounds until own group's tile words / until group words")
for p_ in (0, 1, 4, 8, 12, 14, 15):
    m = pos == p_
    print(f"   r={p_:2d}  {lb[m].mean():6.2f} us   tile part {r_tile[m].mean():5.2f}   group part {r_group[m].mean():5.2f}   all {st[m, 4].mean():5.2f}")
r = st[1:, 4]
print("poll rounds: mean %.2f p50 %.0f p90 %.0f max %.0f; lookback us per round p50 %.2f" % (r.mean(), np.percentile(r,50), np.percentile(r,90), r.max(), np.percentile((resolved-loaded)[1:]/np.maximum(r,1),50)))
print("resolve(i) - resolve(i-1): p50 %.3f p90 %.3f us" % (np.percentile(np.diff(resolved), 50), np.percentile(np.diff(resolved), 90)))
print("resolved - max(loaded(i), loaded(i-1)): p50 %.2f p90 %.2f us" % tuple(np.percentile(resolved[1:] - np.maximum(loaded[1:], loaded[:-1]), [50, 90])))
# the earliest a tile could resolve whatever the protocol: when the last of its predecessors has published
runmax = np.maximum.accumulate(loaded)
inherent = np.maximum(runmax[:-1] - loaded[1:], 0.0)
actual = (resolved - loaded)[1:]
print("inherent wait (max predecessor publish - own publish): mean %.2f p50 %.2f p90 %.2f us" % (inherent.mean(), *np.percentile(inherent, [50, 90])))
print("protocol overhead (actual - inherent): mean %.2f p50 %.2f p90 %.2f us" % ((actual - inherent).mean(), *np.percentile(actual - inherent, [50, 90])))
print("tiles started per us (steady state): %.1f" % ((tiles // 2) / (start[3 * tiles // 4] - start[tiles // 4])))
# occupancy of the phases over time (2 us bins over the middle of the run): a convoy shows as
# the three counts swinging in turn instead of staying level
t_lo, t_hi = done.max() * 0.4, done.max() * 0.4 + 80
print("   t_us  loading waiting writing  starts")
for b in np.arange(t_lo, t_hi, 2.0):
    mid = b + 1.0
    print(f"{b:7.0f} {int(((start <= mid) & (mid < loaded)).sum()):8d} {int(((loaded <= mid) & (mid < resolved)).sum()):7d} "
          f"{int(((resolved <= mid) & (mid < done)).sum()):7d} {int(((start >= b) & (start < b + 2)).sum()):7d}")
for lo in (0, tiles // 2, tiles - 12):
    print("tile start loaded resolved done")
    for i in range(lo, min(lo + 10, tiles)):
        print(f"{i:6d} {start[i]:8.2f} {loaded[i]:8.2f} {resolved[i]:8.2f} {done[i]:8.2f}")
