#!/usr/bin/env python3
"""Kernel-level timings with HIP events (one process, interleaved rounds).

    python tools/microbench.py [--log2n 28] [--rounds 5] [--only decode,encode,...]
Prints one line per operation: median ms, algorithmic GB/s and fraction of 8 TB/s.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import streamvbyte_amd as svb  # noqa: E402
from bench import gen_width_mix, gen_posting_lists, list_lengths  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=28)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--only", default="")
ap.add_argument("--skip", default="", help="groups to leave out (the profile run skips 'ablate': its timing-only launches share kernel names with the real ones)")
a = ap.parse_args()
only = set(a.only.split(",")) if a.only else None
skip = set(a.skip.split(",")) if a.skip else set()
n = 1 << a.log2n
dev = "cuda:0"
codec = svb.DeviceCodec(n, dev)
ops = {}


def want(name):
    return name not in skip and (only is None or name in only)


values = gen_width_mix(torch, n, 1, dev)
stream, sz = codec.encode(values)
size = int(sz.item())
keylen = (n + 3) // 4
dbar = (size - keylen) / n
out = torch.empty(n, dtype=torch.int32, device=dev)
enc_out = torch.empty(svb.streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device=dev)
if want("decode"):
    ops["decode_plan"] = (lambda: codec.decode_plan(stream, n), 0.25 * n)
    ops["decode_run"] = (lambda: codec.decode_run(stream, n, out), (4.25 + dbar) * n)

    def no_shortcut():
        codec.lib.svbgpu_set_option(b"decode_shortcut", 0)
        codec.decode_run(stream, n, out)
        codec.lib.svbgpu_set_option(b"decode_shortcut", 1)
    ops["decode_run_noshortcut"] = (no_shortcut, (4.25 + dbar) * n)
def enc_with(qw, f):
    def g():
        codec.lib.svbgpu_set_option(b"encode_tile", qw)
        f()
        codec.lib.svbgpu_set_option(b"encode_tile", 4)
    return g


def dbg_with(flag, f):
    def g():
        codec.lib.svbgpu_set_option(b"debug_ablate", flag)
        f()
        codec.lib.svbgpu_set_option(b"debug_ablate", 0)
    return g


if want("ablate"):
    ops["encode_nolookback"] = (dbg_with(1, lambda: codec.encode(values, out=enc_out)), (4.25 + dbar) * n)
    ops["encode_nopass2"] = (dbg_with(2, lambda: codec.encode(values, out=enc_out)), (4.25 + dbar) * n)
    ops["encode_idle11us"] = (dbg_with(5, lambda: codec.encode(values, out=enc_out)), (4.25 + dbar) * n)
    ops["encode_neither"] = (dbg_with(3, lambda: codec.encode(values, out=enc_out)), (4.25 + dbar) * n)
def opt_with(name, value, default, f):
    def g():
        codec.lib.svbgpu_set_option(name, value)
        f()
        codec.lib.svbgpu_set_option(name, default)
    return g


if want("encode"):
    ops["encode"] = (lambda: codec.encode(values, out=enc_out), (4.25 + dbar) * n)
    ops["encode_nopipe"] = (opt_with(b"encode_pipeline", 0, 3, lambda: codec.encode(values, out=enc_out)), (4.25 + dbar) * n)
    ops["encode_early"] = (opt_with(b"encode_ticket_early", 1, -1, lambda: codec.encode(values, out=enc_out)), (4.25 + dbar) * n)
    ops["encode_q2w8"] = (opt_with(b"encode_pipeline", 0, 3, enc_with(2, lambda: codec.encode(values, out=enc_out))), (4.25 + dbar) * n)
if want("encode_qw"):
    for qw in (1, 2, 4):
        ops[f"encode_qw{qw}"] = (opt_with(b"encode_pipeline", 0, 3, enc_with(qw, lambda: codec.encode(values, out=enc_out))), (4.25 + dbar) * n)
if want("delta"):
    g = torch.Generator(device=dev); g.manual_seed(7)
    post = torch.cumsum(torch.randint(0, 32, (n,), generator=g, device=dev, dtype=torch.int64), 0)
    post = torch.where(post >= (1 << 31), post - (1 << 32), post).to(torch.int32)
    s3, z3 = codec.encode(post, flags=svb.DELTA)
    s3 = s3.clone()
    d3 = (int(z3.item()) - keylen) / n
    ops["delta_encode"] = (lambda: codec.encode(post, out=enc_out, flags=svb.DELTA), (4.25 + d3) * n)
    ops["delta_encode_pipe"] = (opt_with(b"encode_pipeline", 2, 3, lambda: codec.encode(post, out=enc_out, flags=svb.DELTA)), (4.25 + d3) * n)
    ops["delta_decode"] = (lambda: codec.decode(s3, n, out=out, flags=svb.DELTA), (4.25 + d3) * n)

    ops["delta_decode_w8"] = (opt_with(b"scan_waves", 8, 16, lambda: codec.decode(s3, n, out=out, flags=svb.DELTA)), (4.25 + d3) * n)

    def dd_twopass():
        codec.lib.svbgpu_set_option(b"delta_decode_lookback", 0)
        codec.decode(s3, n, out=out, flags=svb.DELTA)
        codec.lib.svbgpu_set_option(b"delta_decode_lookback", 1)
    ops["delta_decode_twopass"] = (dd_twopass, (4.25 + d3) * n)
    ops["delta_encode_mixed"] = (lambda: codec.encode(values, out=enc_out, flags=svb.DELTA), (4.25 + dbar) * n)
    ops["delta_encode_mixed_pipe"] = (opt_with(b"encode_pipeline", 2, 3, lambda: codec.encode(values, out=enc_out, flags=svb.DELTA)), (4.25 + dbar) * n)
    sm, zm = codec.encode(values, flags=svb.DELTA)
    sm = sm.clone()
    ops["delta_decode_mixed"] = (lambda: codec.decode(sm, n, out=out, flags=svb.DELTA), (int(zm.item()) + 4.0 * n))

    def ddm_twopass():
        codec.lib.svbgpu_set_option(b"delta_decode_lookback", 0)
        codec.decode(sm, n, out=out, flags=svb.DELTA)
        codec.lib.svbgpu_set_option(b"delta_decode_lookback", 1)
    ops["delta_decode_mixed_twopass"] = (ddm_twopass, (int(zm.item()) + 4.0 * n))
if want("zz"):
    n4 = min(n, 1 << 26)
    g = torch.Generator(device=dev); g.manual_seed(9)
    gr = torch.round(torch.randn(n4, generator=g, device=dev) * 0.8).to(torch.int32)
    s4, z4 = codec.encode(gr, variant=svb.V0124, flags=svb.ZIGZAG)
    s4 = s4.clone()
    d4 = (int(z4.item()) - (n4 + 3) // 4) / n4
    o4 = out[:n4]
    ops["zz0124_encode"] = (lambda: codec.encode(gr, out=enc_out, variant=svb.V0124, flags=svb.ZIGZAG), (4.25 + d4) * n4)
    ops["zz0124_encode_nopipe"] = (opt_with(b"encode_pipeline", 0, 3, lambda: codec.encode(gr, out=enc_out, variant=svb.V0124, flags=svb.ZIGZAG)), (4.25 + d4) * n4)
    ops["zz0124_encode_early"] = (opt_with(b"encode_ticket_early", 1, -1, lambda: codec.encode(gr, out=enc_out, variant=svb.V0124, flags=svb.ZIGZAG)), (4.25 + d4) * n4)
    ops["zz0124_encode_q2w8"] = (opt_with(b"encode_pipeline", 0, 3, enc_with(2, lambda: codec.encode(gr, out=enc_out, variant=svb.V0124, flags=svb.ZIGZAG))), (4.25 + d4) * n4)
    ops["zz0124_decode"] = (lambda: codec.decode(s4, n4, out=o4, variant=svb.V0124, flags=svb.ZIGZAG), (4.25 + d4) * n4)
if want("misc"):
    ops["validate"] = (lambda: codec.validate(stream, size, n), 0.25 * n)
    ops["size"] = (lambda: codec.compressedbytes(values), 4.0 * n)
    ops["zigzag_enc"] = (lambda: codec.zigzag(values, "encode", out=out), 8.0 * n)
    tmpz = torch.empty(n, dtype=torch.int32, device=dev)
    ops["zigzag_delta_dec"] = (lambda: codec.zigzag(values, "delta_decode", out=tmpz), 8.0 * n)
    ops["zigzag_delta_dec_fresh"] = (opt_with(b"scan_fresh", 1, 0, lambda: codec.zigzag(values, "delta_decode", out=tmpz)), 8.0 * n)
    ops["zigzag_delta_dec_fresh8"] = (opt_with(b"scan_waves", 8, 16, opt_with(b"scan_fresh", 1, 0, lambda: codec.zigzag(values, "delta_decode", out=tmpz))), 8.0 * n)
if want("intersect"):
    # dense postings against every 8th of their own values: as a sorted array, and as a delta-coded stream
    gi = torch.Generator(device=dev); gi.manual_seed(5)
    pst = torch.cumsum(torch.randint(0, 32 if n <= (1 << 27) else 16, (n,), generator=gi, device=dev, dtype=torch.int64), 0)
    pst = torch.where(pst >= (1 << 31), pst - (1 << 32), pst).to(torch.int32)
    si, zi = codec.encode(pst, flags=svb.DELTA)
    si = si[: int(zi.item()) + 16].clone()
    bi = pst[::8].contiguous()
    sbi, zbi = codec.encode(bi, flags=svb.DELTA)
    sbi = sbi[: int(zbi.item()) + 16].clone()
    hits_i = torch.empty(n, dtype=torch.int32, device=dev)
    ops["intersect_array"] = (lambda: codec.decode_intersect(si, n, bi, out=hits_i), int(zi.item()) + 4.0 * bi.numel() + 4.0 * bi.numel())
    ops["intersect_streams"] = (lambda: codec.intersect_streams(si, n, sbi, bi.numel(), out=hits_i), int(zi.item()) + int(zbi.item()) + 4.0 * bi.numel())
if want("wide"):
    n8 = min(n, 1 << 27)
    g = torch.Generator(device=dev); g.manual_seed(11)
    w8 = torch.randint(0, 4, (n8,), generator=g, device=dev, dtype=torch.int64)
    r8 = torch.randint(0, 1 << 62, (n8,), generator=g, device=dev, dtype=torch.int64)
    v8 = r8 >> (62 - torch.minimum(8 << w8, torch.full_like(w8, 62)))   # widths 1, 2, 4, 8 bytes equally likely
    del w8, r8
    s8, z8 = codec.encode64(v8)
    s8 = s8.clone()
    d8 = (int(z8.item()) - (n8 + 3) // 4) / n8
    o8 = torch.empty(n8, dtype=torch.int64, device=dev)
    e8 = torch.empty((n8 + 3) // 4 + 8 * n8 + 32, dtype=torch.uint8, device=dev)
    ops["encode64"] = (lambda: codec.encode64(v8, out=e8), (8.25 + d8) * n8)
    ops["decode64"] = (lambda: codec.decode64(s8, n8, out=o8), (8.25 + d8) * n8)
    for nw8 in (4, 16):
        ops[f"encode64_waves{nw8}"] = (opt_with(b"encode64_waves", nw8, 8, lambda: codec.encode64(v8, out=e8)), (8.25 + d8) * n8)
if want("wide_batch"):
    # 2^19 lists of 64-bit postings (the 32-bit posting lists' lengths and gaps, widened), delta coded
    nl8 = 1 << 19
    c8, vo8, va8 = gen_posting_lists(torch, list_lengths(nl8, 20241218), 1234, dev)
    va8 = (va8.to(torch.int64) & 0xFFFFFFFF) + (1 << 40)
    tot8 = int(vo8[-1].item())
    cap8 = (c8.to(torch.int64) + 3) // 4 + 8 * c8.to(torch.int64)
    bo8 = torch.zeros(nl8 + 1, dtype=torch.int64, device=dev)
    bo8[1:] = torch.cumsum(cap8, 0)
    bb8 = torch.empty(int(bo8[-1].item()) + 32, dtype=torch.uint8, device=dev)
    bk8 = torch.empty_like(va8)
    pv8 = torch.full((nl8,), 1 << 40, dtype=torch.int64, device=dev)
    sz8 = codec.encode64_batch(va8, vo8[:-1], bb8, bo8[:-1], c8, pv8, flags=svb.DELTA)
    dB8 = float((sz8.sum().item() - ((c8.to(torch.int64) + 3) // 4).sum().item()) / tot8)
    codec.decode64_batch(bb8, bo8[:-1], bk8, vo8[:-1], c8, pv8, flags=svb.DELTA)
    assert torch.equal(bk8, va8)
    ops["batch_encode64"] = (lambda: codec.encode64_batch(va8, vo8[:-1], bb8, bo8[:-1], c8, pv8, flags=svb.DELTA, sizes=sz8), (8.25 + dB8) * tot8)
    ops["batch_decode64"] = (lambda: codec.decode64_batch(bb8, bo8[:-1], bk8, vo8[:-1], c8, pv8, flags=svb.DELTA, sizes=sz8), (8.25 + dB8) * tot8)
if want("batch"):
    nl = 1 << 20
    counts, voff, vals = gen_posting_lists(torch, list_lengths(nl, 20241218), 1234, dev)
    tot = int(voff[-1].item())
    cap = (counts.to(torch.int64) + 3) // 4 + 4 * counts.to(torch.int64) + 16
    boff = torch.zeros(nl + 1, dtype=torch.int64, device=dev)
    boff[1:] = torch.cumsum(cap, 0)
    bbuf = torch.empty(int(boff[-1].item()) + 16, dtype=torch.uint8, device=dev)
    back = torch.empty_like(vals)
    sizes = codec.encode_batch(vals, voff[:-1], bbuf, boff[:-1], counts, None, flags=svb.DELTA)
    dB = float((sizes.sum().item() - ((counts.to(torch.int64) + 3) // 4).sum().item()) / tot)
    ops["batch_encode"] = (lambda: codec.encode_batch(vals, voff[:-1], bbuf, boff[:-1], counts, None, flags=svb.DELTA, sizes=sizes), (4.25 + dB) * tot)
    ops["batch_decode"] = (lambda: codec.decode_batch(bbuf, boff[:-1], back, voff[:-1], counts, None, flags=svb.DELTA, sizes=sizes), (4.25 + dB) * tot)
    for ch in (0, 1, 2, 4, 8, 16, 32):
        ops[f"batch_encode_chunk{ch}"] = (opt_with(b"batch_chunk", ch, -1, lambda: codec.encode_batch(vals, voff[:-1], bbuf, boff[:-1], counts, None, flags=svb.DELTA, sizes=sizes)), (4.25 + dB) * tot)
        ops[f"batch_decode_chunk{ch}"] = (opt_with(b"batch_chunk", ch, -1, lambda: codec.decode_batch(bbuf, boff[:-1], back, voff[:-1], counts, None, flags=svb.DELTA, sizes=sizes)), (4.25 + dB) * tot)

for f, _ in ops.values():
    f()
torch.cuda.synchronize()
times = {k: [] for k in ops}
for r in range(a.rounds):
    for k, (f, _) in ops.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1))
codec.check()
print(f"n=2^{a.log2n} dbar={dbar:.3f}")
for k, (f, nbytes) in ops.items():
    ms = float(np.median(times[k]))
    print(f"{k:18s} {ms:8.4f} ms  min {min(times[k]):8.4f}  {nbytes / ms / 1e6:8.1f} GB/s alg  {nbytes / ms / 1e6 / 8000:6.3f} of 8TB/s")
