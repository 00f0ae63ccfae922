#!/usr/bin/env python3
"""Encoder A/B check: output of svbgpu_encode under an option value against the default's, byte for byte.

    python tools/check_encode_option.py encode_pipeline 2
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import streamvbyte_amd as svb
from bench import gen_width_mix

name, value = sys.argv[1].encode(), int(sys.argv[2])
dev = "cuda:0"
bad = 0
for n in (1, 1000, 16384, 16385, (1 << 20) + 7, (1 << 22), (1 << 24) + 12345, (1 << 26) - 3, 1 << 28):
    codec = svb.DeviceCodec(n, dev)
    g = torch.Generator(device=dev); g.manual_seed(n & 0xFFFF)
    cases = {"mix": gen_width_mix(torch, n, 1, dev),
             "small": torch.round(torch.randn(n, generator=g, device=dev) * 0.8).to(torch.int32),
             "zeros": torch.zeros(n, dtype=torch.int32, device=dev),
             "max": torch.full((n,), -1, dtype=torch.int32, device=dev)}
    for cname, v in cases.items():
        for variant in (svb.V1234, svb.V0124):
            for flags in (0, svb.ZIGZAG):
                ref, rs = codec.encode(v, variant=variant, flags=flags)
                ref = ref[: int(rs.item())].clone()
                codec.lib.svbgpu_set_option(name, value)
                got, gs = codec.encode(v, variant=variant, flags=flags)
                default = {b"encode_pipeline": 3, b"nt_force": -1, b"encode_ticket_early": -1}.get(name, 0)
                codec.lib.svbgpu_set_option(name, default)
                codec.check()
                ok = int(gs.item()) == ref.numel() and torch.equal(got[: ref.numel()], ref)
                if not ok:
                    bad += 1
                    print("MISMATCH", n, cname, variant, flags, int(gs.item()), ref.numel())
    del codec
    print("n", n, "done", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
