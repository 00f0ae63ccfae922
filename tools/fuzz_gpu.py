#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity: sizes, value distributions, variants, flags, misalignments,
single stream, batched, host-pointer calls, frames and list frames (also corrupted), fused intersection.  python tools/fuzz_gpu.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import streamvbyte_amd as svb
from oracle import Oracle
from util import GENERATORS, to_dev, to_np, dev_stream
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
oracle = Oracle()
codec = svb.DeviceCodec(1 << 21)
kinds = list(GENERATORS)


def expected(variant, flags, v, prev):
    if flags & 2:
        s = v.view(np.int32)
        sp = prev - (1 << 32) if prev >= (1 << 31) else prev
        pre = oracle.zigzag_delta_encode(s, sp) if flags & 1 else oracle.zigzag_encode(s)
        return oracle.encode(variant, pre)
    return oracle.encode(variant, v, bool(flags & 1), prev)


def blocky(n):
    """mix of 1024-blocks of different character (exercises the per-wave-tile shortcuts)"""
    v = np.empty(n, dtype=np.uint32)
    for t in range(0, n, 1024):
        m = min(1024, n - t)
        v[t:t + m] = GENERATORS[kinds[int(rng.integers(len(kinds)))]](rng, m) if rng.random() < 0.5 else rng.integers(0, 256, size=m, dtype=np.uint32)
    return v


t_end, it = time.time() + budget, 0
while time.time() < t_end:
    it += 1
    n = int(rng.choice([rng.integers(1, 70), rng.integers(1, 5000), rng.integers(1, 300000), rng.integers(900000, 1200000)]))
    v = blocky(n) if rng.random() < 0.4 else GENERATORS[kinds[int(rng.integers(len(kinds)))]](rng, n)
    variant, flags = int(rng.integers(2)), int(rng.integers(4))
    prev = int(rng.integers(0, 1 << 32)) if flags & 1 else 0
    qw = int(rng.choice([1, 2, 4]))
    lb = int(rng.integers(2))
    tag = (seed, it, n, variant, flags, prev, qw, lb)
    exp = expected(variant, flags, v, prev)
    codec.lib.svbgpu_set_option(b"encode_tile", qw)
    codec.lib.svbgpu_set_option(b"delta_decode_lookback", lb)
    mis = int(rng.integers(0, 4))
    buf = torch.full((mis + svb.streamvbyte_max_compressedbytes(n),), 0xCD, dtype=torch.uint8, device="cuda:0")
    _, size = codec.encode(to_dev(torch, v), out=buf[mis:], variant=variant, flags=flags, prev=prev)
    codec.check()
    assert int(size.item()) == exp.size, ("size",) + tag
    h = buf.cpu().numpy()
    assert np.array_equal(h[mis:mis + exp.size], exp), ("bytes",) + tag
    assert (h[:mis] == 0xCD).all() and (h[mis + exp.size:] == 0xCD).all(), ("stray write",) + tag
    assert int(codec.compressedbytes(to_dev(torch, v), variant=variant, flags=flags, prev=prev).item()) == exp.size, ("cb",) + tag
    d_s = dev_stream(torch, exp, misalign=int(rng.integers(0, 4)))
    assert int(codec.validate(d_s, exp.size, n, variant=variant).item()) == 1, ("validate",) + tag
    if exp.size > (n + 3) // 4:
        assert int(codec.validate(d_s, exp.size - 1, n, variant=variant).item()) == 0, ("validate-1",) + tag
    dec, used = codec.decode(d_s, n, variant=variant, flags=flags, prev=prev)
    codec.check()
    assert int(used.item()) == exp.size and np.array_equal(to_np(dec, np.uint32), v), ("decode",) + tag
    if it % 5 == 0 and not (flags & 2):  # host-pointer API (reference names)
        if variant == 0 and flags & 1:
            e = svb.streamvbyte_delta_encode(v, prev); o, u = svb.streamvbyte_delta_decode(e, n, prev)
        elif variant == 0:
            e = svb.streamvbyte_encode(v); o, u = svb.streamvbyte_decode(e, n)
        elif not flags:
            e = svb.streamvbyte_encode_0124(v); o, u = svb.streamvbyte_decode_0124(e, n)
        else:
            e = None
        if e is not None:
            assert np.array_equal(e, exp) and u == exp.size and np.array_equal(o, v), ("host api",) + tag
    if it % 7 == 0:  # frames
        fv = int(rng.integers(1, 60000))
        fr = svb.frame_encode_many(v, fv, variant=variant, flags=flags)
        assert np.array_equal(svb.frame_decode_many(fr), v), ("frames", fv) + tag
    if it % 11 == 0:  # batched entry points
        ns = int(rng.integers(1, 400))
        top = int(rng.choice([8, 300, 5000]))
        counts = rng.integers(0, top, size=ns).astype(np.uint32)
        prevs = rng.integers(0, 1 << 32, size=ns, dtype=np.uint64).astype(np.uint32)
        voff = np.concatenate([[0], np.cumsum(counts.astype(np.uint64))]).astype(np.int64)
        cap = np.array([svb.streamvbyte_max_compressedbytes(int(c)) for c in counts], dtype=np.int64)
        boff = np.concatenate([[0], np.cumsum(cap)]).astype(np.int64)
        vals = blocky(int(voff[-1])) if voff[-1] else np.zeros(0, dtype=np.uint32)
        d_vals = to_dev(torch, vals) if vals.size else torch.zeros(1, dtype=torch.int32, device="cuda:0")
        d_voff, d_boff = torch.from_numpy(voff[:-1].copy()).cuda(), torch.from_numpy(boff[:-1].copy()).cuda()
        d_counts, d_prevs = to_dev(torch, counts), to_dev(torch, prevs)
        bbuf = torch.zeros(int(boff[-1]) + 16, dtype=torch.uint8, device="cuda:0")
        sizes = codec.encode_batch(d_vals, d_voff, bbuf, d_boff, d_counts, d_prevs, variant=variant, flags=flags)
        hs, hb = sizes.cpu().numpy(), bbuf.cpu().numpy()
        for k in range(ns):
            lv = vals[voff[k]:voff[k + 1]]
            e = expected(variant, flags, lv, int(prevs[k]) if flags & 1 else 0)
            assert hs[k] == e.size and np.array_equal(hb[boff[k]:boff[k] + e.size], e), ("batch encode", k, int(counts[k])) + tag
        back = torch.full_like(d_vals, 0x7B7B7B7B)
        used = codec.decode_batch(bbuf, d_boff, back, d_voff, d_counts, d_prevs, variant=variant, flags=flags)
        assert np.array_equal(used.cpu().numpy(), hs), ("batch used",) + tag
        if vals.size:
            assert np.array_equal(to_np(back, np.uint32), vals), ("batch decode",) + tag
    if it % 9 == 0:  # list frames: round trip, then a corrupted copy must be refused or decode to something, never crash
        ns = int(rng.integers(1, 600))
        counts = rng.integers(0, int(rng.choice([8, 300, 5000])), size=ns)
        start = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        vals = blocky(int(start[-1])) if start[-1] else np.zeros(0, dtype=np.uint32)
        fv = int(rng.integers(1, 200000))
        fr = svb.frame_encode_lists(vals, start, fv, variant=variant, flags=flags)
        got, got_start = svb.frame_decode_lists(fr)
        assert np.array_equal(got, vals) and np.array_equal(got_start, start), ("list frames", ns, fv) + tag
        bad = fr.copy()
        for _ in range(int(rng.integers(1, 4))):
            where = int(rng.integers(0, min(bad.size, 4096))) if rng.random() < 0.7 else int(rng.integers(0, bad.size))
            bad[where] ^= np.uint8(1 << int(rng.integers(0, 8)))
        try:
            svb.frame_decode_lists(bad[: int(rng.integers(1, bad.size + 1))] if rng.random() < 0.3 else bad)
        except svb.SvbError:
            pass
    if it % 6 == 0:  # differential decode fused with an intersection
        a = np.cumsum(rng.integers(0, int(rng.choice([2, 40, 5000])), n, dtype=np.uint64)).astype(np.uint32) if rng.random() < 0.8 else v
        pv = int(rng.integers(0, 1 << 32)) if rng.random() < 0.3 else 0
        a = (a + np.uint32(pv)).astype(np.uint32)
        sa = oracle.encode(variant, a, True, pv)
        m = int(rng.choice([1, max(1, n // 30), n, 8 * n]))
        b = np.unique(np.concatenate([rng.integers(0, int(a.max()) + 2, m, dtype=np.uint64).astype(np.uint32),
                                      a[:: int(rng.integers(1, 50))]]))
        exp_hits = a[np.isin(a, b)]
        o, k, u = codec.decode_intersect(dev_stream(torch, sa), n, to_dev(torch, b), variant=variant, prev=pv)
        codec.check()
        assert int(u.item()) == sa.size and int(k.item()) == exp_hits.size, ("intersect count", m, pv) + tag
        assert np.array_equal(to_np(o[: exp_hits.size], np.uint32), exp_hits), ("intersect", m, pv) + tag
        # the same with both sides compressed (B ascending, its own variant and seed)
        vb = int(rng.integers(0, 2))
        pb = int(b[0]) if (b.size and rng.random() < 0.3) else 0
        sb = oracle.encode(vb, b, True, pb)
        o, k = codec.intersect_streams(dev_stream(torch, sa), n, dev_stream(torch, sb), b.size, variant_a=variant,
                                       prev_a=pv, variant_b=vb, prev_b=pb)
        codec.check()
        assert int(k.item()) == exp_hits.size, ("stream intersect count", m, pv, vb) + tag
        assert np.array_equal(to_np(o[: exp_hits.size], np.uint32), exp_hits), ("stream intersect", m, pv, vb) + tag
codec.lib.svbgpu_set_option(b"encode_tile", 4)
codec.lib.svbgpu_set_option(b"delta_decode_lookback", 1)
print(f"fuzz ok: {it} cases, seed {seed}")
