"""Build libstreamvbyte.so.2 (HIP, gfx950) in-tree.

    python -m streamvbyte_amd.build [--force]

One hipcc invocation over csrc/svb_api.hip; the library lands next to this file
so that it travels with the source tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstreamvbyte.so.2")
LINK = os.path.join(HERE, "libstreamvbyte.so")


def sources():
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    out += [os.path.join(ROOT, "include", f) for f in sorted(os.listdir(os.path.join(ROOT, "include")))]
    return out


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function",
           "-Wl,-soname,libstreamvbyte.so.2",
           "-I", os.path.join(ROOT, "include"),
           "-o", LIB, os.path.join(CSRC, "svb_api.hip"), "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if os.path.lexists(LINK):
        os.remove(LINK)
    os.symlink(os.path.basename(LIB), LINK)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
