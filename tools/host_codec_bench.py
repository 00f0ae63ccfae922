#!/usr/bin/env python3
"""ns per call of the reference-named functions on small host arrays, this library against the reference
(oracle/_ref, when built): compiles tools/native/host_bench.c with gcc and runs it.

    python tools/host_codec_bench.py
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "streamvbyte_amd", "libstreamvbyte.so.2")
ref = os.path.join(ROOT, "oracle", "_ref", "libstreamvbyte_ref.so")
exe = os.path.join(tempfile.mkdtemp(), "host_bench")
subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "native", "host_bench.c"), "-ldl"])
sys.exit(subprocess.call([exe, lib] + ([ref] if os.path.exists(ref) else [])))
