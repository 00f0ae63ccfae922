#!/usr/bin/env python3
"""Batched encode / decode of the posting-list mix (lists of log-uniform length 1..4096) under
different grids: python tools/batch_mix.py [lists_per_wave ...]   (0 = persistent grid)
SVB_NT_FORCE=<0..3> in the environment forces a cache policy (default: the library's own choice);
MIX_MIN / MIX_MAX / MIX_LISTS restrict the lengths to a part of the mix (default 1 / 4096 / 2^20 lists);
MIX_ROUND rounds them up to a multiple (aligned list starts), MIX_BALIGN does the same for the lists' bytes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
dev = "cuda:0"
nl = int(os.environ.get("MIX_LISTS", 1 << 20))
rng = np.random.default_rng(5)
lo, hi = int(os.environ.get("MIX_MIN", 1)), int(os.environ.get("MIX_MAX", 4096))
lens = np.floor(np.exp(rng.uniform(np.log(lo), np.log(hi), nl))).astype(np.int64)
rnd = int(os.environ.get("MIX_ROUND", 1))  # lengths rounded up to a multiple: every list then starts on that alignment
lens = (lens + rnd - 1) // rnd * rnd
start = np.concatenate([[0], np.cumsum(lens)])
n = int(start[-1])
g = torch.Generator(device=dev); g.manual_seed(1)
gaps = torch.randint(0, 1 << 16, (n,), generator=g, device=dev, dtype=torch.int64) >> torch.randint(0, 12, (n,), generator=g, device=dev, dtype=torch.int64)
vals = torch.cumsum(gaps, 0)
first = torch.from_numpy(start[:-1]).to(dev)
vals = (vals - torch.repeat_interleave(vals[first] - gaps[first], torch.from_numpy(lens).to(dev))).to(torch.int32)
del gaps
counts = torch.from_numpy(lens.astype(np.int32)).to(dev)
voff = first
cap = (lens + 3) // 4 + 4 * lens + 16
bal = int(os.environ.get("MIX_BALIGN", 1))  # room per list rounded up to a multiple: every list's bytes then start on that alignment
cap = (cap + bal - 1) // bal * bal
boff = torch.from_numpy(np.concatenate([[0], np.cumsum(cap)])[:-1]).to(dev)
bbuf = torch.empty(int(cap.sum()) + 16, dtype=torch.uint8, device=dev)
back = torch.empty_like(vals)
codec = svb.DeviceCodec(1 << 20, dev)
codec.lib.svbgpu_set_option(b"nt_force", int(os.environ.get("SVB_NT_FORCE", "-1")))
sizes = codec.encode_batch(vals, voff, bbuf, boff, counts, None, flags=svb.DELTA)
ab = 4.0 * n + float(sizes.sum().item())


def t(f):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))


for k in [int(x) for x in sys.argv[1:]] or [0, 1, 2, 4, 8, 16]:
    codec.lib.svbgpu_set_option(b"batch_lists_per_wave", k)
    te = t(lambda: codec.encode_batch(vals, voff, bbuf, boff, counts, None, flags=svb.DELTA, sizes=sizes))
    td = t(lambda: codec.decode_batch(bbuf, boff, back, voff, counts, None, flags=svb.DELTA))
    assert torch.equal(back, vals)
    print(f"{nl} lists of {lo}..{hi} values, {n} in all; lists per wave {k:3d}: encode {te:6.3f} ms {ab / te / 1e6 / 8000:5.3f} of peak   decode {td:6.3f} ms {ab / td / 1e6 / 8000:5.3f} of peak")
codec.lib.svbgpu_set_option(b"batch_lists_per_wave", 0)
