#!/usr/bin/env python3
"""svbgpu_intersect_batch: 2^18 pairs of delta-coded posting lists (lengths log-uniform 1..4096, the batched
benchmark's lists), list i of one set against list i of another, checked against decoding both and torch.isin."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import gen_posting_lists, list_lengths, event_ms
dev = "cuda:0"
npairs = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 18)
codec = svb.DeviceCodec(1 << 20, dev)


def make(seed_len, seed_val):
    counts, voff, vals = gen_posting_lists(torch, list_lengths(npairs, seed_len), seed_val, dev)
    cap = (counts.to(torch.int64) + 3) // 4 + 4 * counts.to(torch.int64)
    boff = torch.zeros(npairs + 1, dtype=torch.int64, device=dev)
    boff[1:] = torch.cumsum(cap, 0)
    bbuf = torch.zeros(int(boff[-1].item()) + 16, dtype=torch.uint8, device=dev)
    sizes = codec.encode_batch(vals, voff[:-1], bbuf, boff[:-1], counts, None, flags=svb.DELTA)
    return counts, voff, vals, bbuf, boff, int(sizes.sum().item())


ca, voa, va, ba, boa, bytes_a = make(1, 11)
cb, vob, vb, bb, bob, bytes_b = make(2, 22)
out = torch.empty(int(voa[-1].item()) + 1, dtype=torch.int32, device=dev)
f = lambda: codec.intersect_batch(ba, boa[:-1], ca, None, bb, bob[:-1], cb, None, out, voa[:-1])
_, ms = event_ms(torch, f, 10)
_, n_out = f()
torch.cuda.synchronize()
na, nb = int(voa[-1].item()), int(vob[-1].item())
# check: pair ids make the values of different pairs distinct, then one isin over everything
ida = torch.repeat_interleave(torch.arange(npairs, device=dev), ca.to(torch.int64))
idb = torch.repeat_interleave(torch.arange(npairs, device=dev), cb.to(torch.int64))
ka = (ida << 32) | (va.to(torch.int64) & 0xFFFFFFFF)
kb = (idb << 32) | (vb.to(torch.int64) & 0xFFFFFFFF)
hit = torch.isin(ka, kb)
exp_n = torch.zeros(npairs, dtype=torch.int64, device=dev).scatter_add_(0, ida, hit.to(torch.int64))
assert torch.equal(exp_n, n_out.to(torch.int64)), "hit counts differ"
pos = torch.arange(na, device=dev) - torch.repeat_interleave(voa[:-1], ca.to(torch.int64))   # index inside the list
rank = torch.cumsum(hit.to(torch.int64), 0) - 1
first_rank = torch.repeat_interleave((torch.cumsum(exp_n, 0) - exp_n), ca.to(torch.int64))
dst = torch.repeat_interleave(voa[:-1], ca.to(torch.int64)) + (rank - first_rank)
assert torch.equal(out[dst[hit]], va[hit]), "hits differ"
print(f"{npairs} pairs, {na} + {nb} values, {bytes_a + bytes_b} stream bytes, {int(exp_n.sum().item())} hits: "
      f"{ms:.3f} ms, {(na + nb) / ms / 1e6:.1f} G values/s, {(bytes_a + bytes_b) / ms / 1e6:.0f} GB/s of streams")
dec_a, dec_b = torch.empty_like(va), torch.empty_like(vb)
_, ms2 = event_ms(torch, lambda: (codec.decode_batch(ba, boa[:-1], dec_a, voa[:-1], ca, None, flags=svb.DELTA),
                                  codec.decode_batch(bb, bob[:-1], dec_b, vob[:-1], cb, None, flags=svb.DELTA)), 10)
print(f"decoding both sets to memory alone (svbgpu_decode_batch x 2): {ms2:.3f} ms")
