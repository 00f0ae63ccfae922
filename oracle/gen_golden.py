#!/usr/bin/env python3
"""Generate tests/golden/ from the reference (run in the build container only).

Inputs : /root/reference/tests/unit.c   (the issue42 byte array is DATA: it is
                                          parsed into a .bin, no source text kept)
         oracle/_ref/libstreamvbyte_ref.so (reference compiled by oracle/Makefile)
Outputs: tests/golden/issue42_stream.bin   36 494-byte stream  (tests/unit.c:312-2746)
         tests/golden/golden.npz           arrays, see keys below
         tests/golden/golden.json          scalar known answers

Everything written is input/expected-output data.  Re-run with
    make -C oracle && python oracle/gen_golden.py
"""
import ctypes
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from oracle import Reference, V1234, V0124  # noqa: E402

REF_UNIT = "/root/reference/tests/unit.c"
OUT = os.path.join(ROOT, "tests", "golden")


def parse_issue42():
    """Pull the integers between `uint8_t a[36494] = {` and `};`."""
    text = open(REF_UNIT).read()
    m = re.search(r"uint8_t a\[36494\] = \{(.*?)\};", text, re.S)
    vals = [int(x) for x in re.findall(r"\d+", m.group(1))]
    assert len(vals) == 36494
    return np.array(vals, dtype=np.uint8)


def width_mix(rng, n):
    """uniform widths 1..4 bytes, value uniform inside the width class"""
    w = rng.integers(1, 5, size=n)
    lo = np.where(w == 1, 0, 1 << (8 * (w - 1)))
    hi = 1 << (8 * w)
    return (lo + (rng.random(n) * (hi - lo)).astype(np.uint64)).astype(np.uint32)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = Reference()
    arrays, scal = {}, {}

    # --- issue42 (tests/unit.c:310-2783): decode returns 36494, re-encode returns 36494
    a = parse_issue42()
    a.tofile(os.path.join(OUT, "issue42_stream.bin"))
    dec, used = ref.decode(V1234, a, 29159)
    assert used == 36494
    arrays["issue42_decoded"] = dec
    re_enc = ref.encode(V1234, dec)
    scal["issue42"] = {"count": 29159, "bytes": 36494,
                       "reencode_equal": bool(np.array_equal(re_enc, a))}

    # --- issue69 (tests/unit.c:2786-2805)
    i69 = np.array([431, 292, 979, 994, 761, 879, 672, 690, 296, 931, 379, 98, 132, 105,
                    116, 841, 387, 831, 335, 333, 557, 915], dtype=np.uint32)
    arrays["issue69_values"] = i69
    arrays["issue69_stream"] = ref.encode(V1234, i69)

    # --- README.md:228-242 known answer
    rd = np.arange(0, 800, 100, dtype=np.uint32)
    arrays["readme_values"] = rd
    arrays["readme_stream"] = ref.encode(V1234, rd)

    scal["readme_hex"] = arrays["readme_stream"].tobytes().hex()

    # --- zigzag fixtures (tests/unit.c:60-87)
    zi = np.array([0, -1, 1, -2, 2**31 - 1, -2**31], dtype=np.int64).astype(np.int32)
    arrays["zigzag_fix_in"] = zi
    arrays["zigzag_fix_out"] = ref.zigzag_encode(zi)
    assert arrays["zigzag_fix_out"].tolist() == [0, 1, 2, 3, 2**32 - 2, 2**32 - 1]

    # --- compressedbytes known answers (tests/unit.c:254-308)
    cb = {}
    for name, vals in (("ones", [1] * 4), ("260", [260] * 4), ("max", [2**32 - 1] * 4),
                       ("zeros", [0] * 4), ("empty", [])):
        v = np.array(vals, dtype=np.uint32)
        cb[name] = [int(ref.compressedbytes(V1234, v)), int(ref.compressedbytes(V0124, v))]
    scal["compressedbytes"] = cb

    # --- perf.c-style data (tests/perf.c:52-53): glibc rand(), default seed
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    n = 500000
    d = np.empty(n, dtype=np.uint32)
    for k in range(n):
        r = libc.rand()
        s = libc.rand()
        d[k] = (r >> (31 & s)) & 0xFFFFFFFF
    scal["perf_c"] = {"n": n, "bytes": int(ref.encode(V1234, d).size),
                      "xor": int(np.bitwise_xor.reduce(d)), "sum": int(d.astype(np.uint64).sum())}

    # --- seeded random cases, every codec variant, reference outputs
    rng = np.random.default_rng(0x5356425F)  # "SVB_"
    sizes = [0, 1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 255, 256, 257,
             1000, 1023, 1024, 1025, 4095, 4096, 4097, 5000, 16385]
    cases = []
    for ci, n in enumerate(sizes):
        kind = ci % 4
        if kind == 0:
            v = width_mix(rng, n)
        elif kind == 1:  # small values with zeros (0124-friendly)
            v = (rng.integers(0, 4, size=n) * rng.integers(0, 300, size=n)).astype(np.uint32)
        elif kind == 2:  # sorted postings
            v = np.cumsum(rng.integers(0, 70000, size=n), dtype=np.uint64).astype(np.uint32)
        else:  # full-range, unsorted (delta wraps)
            v = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
        prev = int(rng.integers(0, 2**32)) if ci % 2 else 0
        arrays[f"case{ci}_values"] = v
        arrays[f"case{ci}_s1234"] = ref.encode(V1234, v)
        arrays[f"case{ci}_s0124"] = ref.encode(V0124, v)
        arrays[f"case{ci}_sdelta"] = ref.encode(V1234, v, delta=True, prev=prev)
        # decode the delta stream with the reference too (pins delta_decode on its own)
        dd, used = ref.decode(V1234, arrays[f"case{ci}_sdelta"], n, delta=True, prev=prev)
        assert np.array_equal(dd, v) and used == arrays[f"case{ci}_sdelta"].size
        sv = v.view(np.int32)
        zprev = int(np.int32(np.uint32(prev)))
        arrays[f"case{ci}_zz"] = ref.zigzag_encode(sv)
        arrays[f"case{ci}_zzd"] = ref.zigzag_delta_encode(sv, zprev)
        assert np.array_equal(ref.zigzag_decode(arrays[f"case{ci}_zz"]), sv)
        assert np.array_equal(ref.zigzag_delta_decode(arrays[f"case{ci}_zzd"], zprev), sv)
        cases.append({"id": ci, "n": n, "prev": prev,
                      "cb1234": int(ref.compressedbytes(V1234, v)),
                      "cb0124": int(ref.compressedbytes(V0124, v))})
    scal["cases"] = cases

    # --- small hand-checkable vectors (widths 1,2,3,4; 0124 zeros; delta wrap-around)
    sv = {}
    for name, variant, vals, delta, prev in (
            ("mixed1234", V1234, [1, 256, 65536, 1 << 24, 5], False, 0),
            ("mixed0124", V0124, [0, 1, 256, 65536, 0, 7], False, 0),
            ("deltawrap", V1234, [10, 5, 5, 4000000000, 3], True, 7)):
        sv[name] = {"values": vals, "prev": prev,
                    "hex": ref.encode(variant, np.array(vals, dtype=np.uint32), delta, prev).tobytes().hex()}
    scal["vectors"] = sv

    np.savez_compressed(os.path.join(OUT, "golden.npz"), **arrays)
    json.dump(scal, open(os.path.join(OUT, "golden.json"), "w"), indent=1, sort_keys=True)
    print("wrote", OUT, {k: os.path.getsize(os.path.join(OUT, k)) for k in os.listdir(OUT)})


if __name__ == "__main__":
    main()
