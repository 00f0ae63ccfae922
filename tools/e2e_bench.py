#!/usr/bin/env python3
"""End-to-end rate with the data starting and ending in (pinned) host memory.

A batch of independent streams is pushed through a three-stage pipeline
(hipMemcpyAsync H2D -> kernels -> hipMemcpyAsync D2H) on three HIP streams with
a ring of device buffers, so that copies in both directions overlap the
kernels.  Prints one JSON line; the numbers are PCIe-bound: a companion to the
device-resident rates of bench.py, never its `value`.

    python tools/e2e_bench.py [--log2n 24] [--streams 16] [--ring 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import streamvbyte_amd as svb  # noqa: E402
from bench import gen_width_mix  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=24)
ap.add_argument("--streams", type=int, default=16)
ap.add_argument("--ring", type=int, default=3)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = "cuda:0"
n = 1 << a.log2n
cap = svb.streamvbyte_max_compressedbytes(n)

# host side: values and their encoded streams in pinned memory
codec0 = svb.DeviceCodec(n, dev)
h_vals = [torch.empty(n, dtype=torch.int32).pin_memory() for _ in range(a.streams)]
h_strm = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(a.streams)]
sizes = []
for i in range(a.streams):
    v = gen_width_mix(torch, n, 100 + i, dev)
    s, z = codec0.encode(v)
    sizes.append(int(z.item()))
    h_vals[i].copy_(v)
    h_strm[i][: sizes[i] + 16].copy_(s[: sizes[i] + 16])
torch.cuda.synchronize()
h_out_vals = [torch.empty(n, dtype=torch.int32).pin_memory() for _ in range(a.streams)]
h_out_strm = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(a.streams)]

ring = a.ring
codecs = [svb.DeviceCodec(n, dev) for _ in range(ring)]
d_vals = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(ring)]
d_strm = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(ring)]
s_in, s_k, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


def run(direction):
    ev_in = [torch.cuda.Event() for _ in range(a.streams)]
    ev_k = [torch.cuda.Event() for _ in range(a.streams)]
    ev_out = [torch.cuda.Event() for _ in range(a.streams)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.streams):
        b = i % ring
        with torch.cuda.stream(s_in):
            if i >= ring:
                s_in.wait_event(ev_out[i - ring])  # slot free again
            if direction == "decode":
                d_strm[b][: sizes[i] + 16].copy_(h_strm[i][: sizes[i] + 16], non_blocking=True)
            else:
                d_vals[b].copy_(h_vals[i], non_blocking=True)
            ev_in[i].record(s_in)
        with torch.cuda.stream(s_k):
            s_k.wait_event(ev_in[i])
            if i >= ring:
                s_k.wait_event(ev_out[i - ring])
            if direction == "decode":
                codecs[b].decode(d_strm[b], n, out=d_vals[b])
            else:
                codecs[b].encode(d_vals[b], out=d_strm[b])
            ev_k[i].record(s_k)
        with torch.cuda.stream(s_out):
            s_out.wait_event(ev_k[i])
            if direction == "decode":
                h_out_vals[i].copy_(d_vals[b], non_blocking=True)
            else:  # size known from the first pass; a real caller reads the 8-byte size word first
                h_out_strm[i][: sizes[i]].copy_(d_strm[b][: sizes[i]], non_blocking=True)
            ev_out[i].record(s_out)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


res = {"n_per_stream": n, "streams": a.streams, "ring": ring}
for direction in ("decode", "encode"):
    run(direction)
    t = min(run(direction) for _ in range(a.reps))
    ok = all(torch.equal(h_out_vals[i], h_vals[i]) for i in range(a.streams)) if direction == "decode" else \
        all(torch.equal(h_out_strm[i][: sizes[i]], h_strm[i][: sizes[i]]) for i in range(a.streams))
    moved = sum(sizes) + 4.0 * n * a.streams
    res[direction] = {"GBps_uint32_side": round(4.0 * n * a.streams / t / 1e9, 2),
                      "Gints_per_s": round(n * a.streams / t / 1e9, 3),
                      "pcie_GBps_both_directions": round(moved / t / 1e9, 2), "ms": round(t * 1e3, 3), "parity": ok}
print(json.dumps(res))
