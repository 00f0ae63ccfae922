#!/usr/bin/env python3
"""Host-pointer API from many threads at once (re-entrancy stress)."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ctypes, subprocess, tempfile
_so = os.path.join(tempfile.gettempdir(), "svb_abort_trace.so")
subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", _so, os.path.join(ROOT, "tools", "native", "abort_trace.c")])
import streamvbyte_amd as svb
svb.load()
ctypes.CDLL(_so).abort_trace_install()
from oracle import Oracle, V1234, V0124
from util import GENERATORS
oracle = Oracle()
T, ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 40
what = sys.argv[3] if len(sys.argv) > 3 else "all"
serialize = int(sys.argv[4]) if len(sys.argv) > 4 else 1   # one host-pointer call at a time per device
stage = int(sys.argv[5]) if len(sys.argv) > 5 else 1       # pageable memory through page-locked chunks
L = svb.load()
assert L.svbgpu_set_option(b"host_cpu_cutoff", 0) == 0     # every call to the GPU, whatever its size
assert L.svbgpu_set_option(b"host_serialize", serialize) == 0
assert L.svbgpu_set_option(b"host_stage_pageable", stage) == 0
stack_status = len(sys.argv) > 6 and sys.argv[6] == "stack"  # look-back status copied into a stack variable (as before the fix)
assert L.svbgpu_set_option(b"debug_ablate", 128 if stack_status else 0) == 0
print(f"threads {T} iters {ITERS} what {what} serialize {serialize} stage_pageable {stage} stack_status {stack_status}", flush=True)
errors = []

def worker(t):
    try:
        rng = np.random.default_rng(900 + t)
        for i in range(ITERS):
            n = int(rng.integers(1, 60000))
            v = GENERATORS[["width_mix", "sparse", "postings", "perf_c"][(t + i) % 4]](rng, n)
            if what in ("all", "codec"):
                e = svb.streamvbyte_encode(v)
                assert np.array_equal(e, oracle.encode(V1234, v)), ("encode", t, i)
                out, used = svb.streamvbyte_decode(e, n)
                assert used == e.size and np.array_equal(out, v), ("decode", t, i)
                assert svb.streamvbyte_validate_stream(e, n)
                d = svb.streamvbyte_delta_encode(v, prev=t)
                assert np.array_equal(svb.streamvbyte_delta_decode(d, n, prev=t)[0], v)
                s = v.view(np.int32)
                assert np.array_equal(svb.zigzag_delta_decode(svb.zigzag_delta_encode(s, -t), -t), s)
            if what in ("all", "frames"):
                f = svb.frame_encode_many(v, 7000, flags=svb.DELTA)
                assert np.array_equal(svb.frame_decode_many(f), v), ("frames", t, i)
    except BaseException as exc:
        errors.append(repr(exc))

ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
[th.start() for th in ths]; [th.join() for th in ths]
print("errors:", errors[:3] if errors else "none", flush=True)
