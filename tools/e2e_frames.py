#!/usr/bin/env python3
"""End-to-end rates of the frame pipeline (host memory in, host memory out):
svbframe_encode_many / svbframe_decode_many over pinned and pageable buffers.

    python tools/e2e_frames.py [--log2n 28] [--log2frame 24]
"""
import argparse, ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import streamvbyte_amd as svb

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=28)
ap.add_argument("--log2frame", type=int, default=24)
a = ap.parse_args()
lib = svb.load()
n, fv = 1 << a.log2n, 1 << a.log2frame
rng = np.random.default_rng(3)
w = rng.integers(0, 4, size=n, dtype=np.uint8)
v = (rng.integers(0, 1 << 32, size=n, dtype=np.uint64) >> (8 * (3 - w)).astype(np.uint64)).astype(np.uint32)
del w
cap = lib.svbframe_max_bytes_many(n, fv)


def pinned(nbytes):
    p = lib.svbframe_alloc_pinned(nbytes)
    assert p
    return p


def run(label, p_in, p_frames, p_out):
    written, nv = C.c_size_t(0), C.c_size_t(0)
    best_e = best_d = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        rc = lib.svbframe_encode_many(p_in, n, fv, 0, 0, p_frames, cap, C.addressof(written))
        t1 = time.perf_counter()
        assert rc == 0, rc
        rc = lib.svbframe_decode_many(p_frames, written.value, p_out, n, C.addressof(nv))
        t2 = time.perf_counter()
        assert rc == 0 and nv.value == n, (rc, nv.value)
        best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
    back = np.ctypeslib.as_array(C.cast(p_out, C.POINTER(C.c_uint32)), shape=(n,))
    assert np.array_equal(back, v)
    fb = written.value
    print(f"{label:9s} frames of 2^{a.log2frame}: encode {4 * n / best_e / 1e9:6.1f} GB/s (uint32 in), "
          f"decode {4 * n / best_d / 1e9:6.1f} GB/s (uint32 out); bytes moved over PCIe per value {(4 * n + fb) / n:.2f}; "
          f"{fb} frame bytes")


# pinned
pi, pf, po = pinned(4 * n), pinned(cap), pinned(4 * n)
C.memmove(pi, v.ctypes.data, 4 * n)
run("pinned", pi, pf, po)
for p in (pi, pf, po):
    lib.svbframe_free_pinned(p)
# pageable (registered for the duration of each call when possible)
frames = np.empty(cap, dtype=np.uint8)
out = np.empty(n, dtype=np.uint32)
run("pageable", v.ctypes.data, frames.ctypes.data, out.ctypes.data)

# list frames: posting lists of log-uniform length 1..4096, sorted, delta coded, whole lists per frame
n_lists = max(1, n // 492)
lens = np.floor(np.exp(rng.uniform(0, np.log(4096), n_lists))).astype(np.int64)
start = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
nl_values = int(start[-1])
gaps = (rng.integers(1, 1 << 16, size=nl_values, dtype=np.uint32) >> rng.integers(0, 12, size=nl_values, dtype=np.uint32)) + 1
lists = np.cumsum(gaps, dtype=np.uint64).astype(np.uint32)
lists -= np.repeat(lists[start[:-1].astype(np.int64)] - gaps[start[:-1].astype(np.int64)], lens)  # every list starts near zero
del gaps
lcap = lib.svbframe_lists_max_bytes(nl_values, n_lists, fv)
pi, pf, po = pinned(4 * nl_values), pinned(lcap), pinned(4 * nl_values)
C.memmove(pi, lists.ctypes.data, 4 * nl_values)
start_back = np.empty(n_lists + 1, dtype=np.uint64)
written, nl, nv = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
best_e = best_d = 1e9
for _ in range(4):
    t0 = time.perf_counter()
    rc = lib.svbframe_encode_lists(pi, start.ctypes.data, n_lists, fv, 0, 1, pf, lcap, C.addressof(written))
    t1 = time.perf_counter()
    assert rc == 0, rc
    rc = lib.svbframe_decode_lists(pf, written.value, po, nl_values, start_back.ctypes.data, n_lists, C.addressof(nl),
                                   C.addressof(nv))
    t2 = time.perf_counter()
    assert rc == 0 and nv.value == nl_values and nl.value == n_lists, (rc, nv.value, nl.value)
    best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
back = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint32)), shape=(nl_values,))
assert np.array_equal(back, lists) and np.array_equal(start_back, start)
print(f"pinned    list frames, {n_lists} delta-coded lists, {nl_values} values: encode {4 * nl_values / best_e / 1e9:6.1f} GB/s, "
      f"decode {4 * nl_values / best_d / 1e9:6.1f} GB/s (uint32 side); {written.value / nl_values:.2f} frame bytes per value")
for p in (pi, pf, po):
    lib.svbframe_free_pinned(p)
