#!/usr/bin/env python3
"""Non-temporal policy sweep: every streaming call under nt_force = 0 (none), 1 (loads), 2 (stores),
3 (both) and -1 (the library's own choice by size), each timed as `reps` back-to-back calls inside one
event pair (the regime bench.py measures), median of `rounds`.  A policy a kernel is not built for runs
as 0.

    python tools/nt_sweep.py [--log2n 28] [--log2n-zz 26]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import streamvbyte_amd as svb  # noqa: E402
from bench import gen_width_mix  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=28)
ap.add_argument("--log2n-zz", type=int, default=26)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
dev = "cuda:0"
n = 1 << a.log2n
codec = svb.DeviceCodec(n, dev)
ops = {}

values = gen_width_mix(torch, n, 1, dev)
stream, sz = codec.encode(values)
stream = stream.clone()
keylen = (n + 3) // 4
dbar = (int(sz.item()) - keylen) / n
out = torch.empty(n, dtype=torch.int32, device=dev)
enc_out = torch.empty(svb.streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device=dev)
ops["decode"] = (lambda: codec.decode(stream, n, out=out), (4.25 + dbar) * n)
ops["encode"] = (lambda: codec.encode(values, out=enc_out), (4.25 + dbar) * n)
ops["size"] = (lambda: codec.compressedbytes(values), 4.0 * n)
ops["zigzag_enc"] = (lambda: codec.zigzag(values, "encode", out=out), 8.0 * n)
ops["zigzag_delta_dec"] = (lambda: codec.zigzag(values, "delta_decode", out=out), 8.0 * n)

g = torch.Generator(device=dev); g.manual_seed(7)
post = torch.cumsum(torch.randint(0, 32, (n,), generator=g, device=dev, dtype=torch.int64), 0)
post = torch.where(post >= (1 << 31), post - (1 << 32), post).to(torch.int32)
s3, z3 = codec.encode(post, flags=svb.DELTA)
s3 = s3.clone()
d3 = (int(z3.item()) - keylen) / n
ops["delta_encode"] = (lambda: codec.encode(post, out=enc_out, flags=svb.DELTA), (4.25 + d3) * n)
ops["delta_decode"] = (lambda: codec.decode(s3, n, out=out, flags=svb.DELTA), (4.25 + d3) * n)

n4 = 1 << a.log2n_zz
g = torch.Generator(device=dev); g.manual_seed(9)
gr = torch.round(torch.randn(n4, generator=g, device=dev) * 0.8).to(torch.int32)
s4, z4 = codec.encode(gr, variant=svb.V0124, flags=svb.ZIGZAG)
s4 = s4.clone()
d4 = (int(z4.item()) - (n4 + 3) // 4) / n4
o4 = out[:n4]
ops["zz0124_encode"] = (lambda: codec.encode(gr, out=enc_out, variant=svb.V0124, flags=svb.ZIGZAG), (4.25 + d4) * n4)
ops["zz0124_decode"] = (lambda: codec.decode(s4, n4, out=o4, variant=svb.V0124, flags=svb.ZIGZAG), (4.25 + d4) * n4)

policies = (0, 1, 2, 3, -1)
res = {k: {} for k in ops}
for k, (f, nbytes) in ops.items():
    for p in policies:
        codec.lib.svbgpu_set_option(b"nt_force", p)
        f(); f()
        torch.cuda.synchronize()
        t = []
        for _ in range(a.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) / a.reps)
        res[k][p] = float(np.median(t))
codec.lib.svbgpu_set_option(b"nt_force", -1)
codec.check()
print(f"n=2^{a.log2n} (zz: 2^{a.log2n_zz}); ms per call and fraction of 8 TB/s, policies none / loads / stores / both / auto")
for k, (f, nbytes) in ops.items():
    print(f"{k:18s} " + "  ".join(f"{res[k][p]:.4f} ({nbytes / res[k][p] / 1e6 / 8000:.3f})" for p in policies))
