#!/usr/bin/env python3
"""Differential decoder A/B check: svbgpu_decode(.., DELTA) under an option value against the default's
output, every integer, both variants, with and without zigzag, dense and wide gaps.

    python tools/check_delta_option.py scan_waves 8 16
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import gen_width_mix

name, value = sys.argv[1].encode(), int(sys.argv[2])
default = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = "cuda:0"
bad = 0
for n in (1, 5, 1000, 1024, 16384, 16385, 65537, (1 << 20) + 7, (1 << 22), (1 << 24) + 12345, (1 << 26) - 3, 1 << 28):
    codec = svb.DeviceCodec(n, dev)
    g = torch.Generator(device=dev); g.manual_seed(n & 0xFFFF)
    cases = {"mix": gen_width_mix(torch, n, 1, dev),
             "dense": torch.randint(0, 32, (n,), generator=g, device=dev, dtype=torch.int32),
             "zeros": torch.zeros(n, dtype=torch.int32, device=dev)}
    for cname, v in cases.items():
        for variant in (svb.V1234, svb.V0124):
            for flags in (svb.DELTA, svb.DELTA | svb.ZIGZAG):
                # the values are used as the stream's deltas: encode them plainly, decode differentially
                stream, ss = codec.encode(v, variant=variant, flags=flags & svb.ZIGZAG)
                stream = stream.clone()
                ref, ru = codec.decode(stream, n, variant=variant, flags=flags, prev=12345)
                ref = ref.clone()
                codec.lib.svbgpu_set_option(name, value)
                got, gu = codec.decode(stream, n, variant=variant, flags=flags, prev=12345)
                codec.lib.svbgpu_set_option(name, default)
                codec.check()
                ok = int(gu.item()) == int(ru.item()) == int(ss.item()) and torch.equal(got, ref)
                if not ok:
                    bad += 1
                    print("MISMATCH", n, cname, variant, flags, int(gu.item()), int(ru.item()))
    del codec
    print("n", n, "done", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
