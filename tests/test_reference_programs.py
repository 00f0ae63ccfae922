"""The reference's own test programs, compiled unchanged from where they lie under /root/reference
against libstreamvbyte.so.2 and this repository's headers (nothing of them is copied here).

tests/unit.c (reference tests/unit.c:2807-2836) is the reference's whole acceptance suite: the issue42
golden stream, issue69, the zigzag fixtures and round trips, the length x width sweeps, the 65 536 aqrit
patterns and the compressedbytes answers, all through the 14 public functions.  Every call in it is below
the library's host cutoff, so it runs here without a GPU, on the host codec -- which is what a drop-in user
of the reference API gets for such sizes.  tests/perf.c (reference tests/perf.c:38-91) is the reference's
benchmark; it must run and report equal data.  Skipped where /root/reference is absent (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "tests", "unit.c")),
                                reason="reference sources not mounted")


def _build(tmp_path, name, extra=()):
    import streamvbyte_amd
    exe = str(tmp_path / name)
    libdir = os.path.dirname(streamvbyte_amd.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-O2", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "c"),
           os.path.join(REF, "tests", name + ".c"), "-o", exe, "-L", libdir, "-l:libstreamvbyte.so.2",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib", *extra]
    subprocess.check_call(cmd)
    return exe


def test_reference_unit_suite_passes_against_this_library(tmp_path):
    exe = _build(tmp_path, "unit")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    assert "Code looks good." in out.stdout, out.stdout[-2000:]


def test_reference_perf_program_runs_against_this_library(tmp_path):
    exe = _build(tmp_path, "perf", extra=("-lm",))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    assert "bug" not in out.stdout.lower(), out.stdout[-2000:]
    print(out.stdout)
