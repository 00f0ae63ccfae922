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
# the same library with the look-back's stress hooks compiled in (csrc/svb_lookback.h: lb_stress_delay);
# loaded only by tests/test_gpu_lookback_stress.py, through SVB_LIB_PATH
STRESS_LIB = os.path.join(HERE, "libstreamvbyte_stress.so.2")


def sources():
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    out += [os.path.join(ROOT, "include", f) for f in sorted(os.listdir(os.path.join(ROOT, "include")))]
    return out


def stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(s) > t for s in sources())


def _compile(out, extra, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", *extra,
           "-Wl,-soname,libstreamvbyte.so.2",
           "-I", os.path.join(ROOT, "include"),
           "-o", out, os.path.join(CSRC, "svb_api.hip"), "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    return subprocess.Popen(cmd)


def build(force=False, verbose=True):
    """Both libraries, compiled side by side (two hipcc processes)."""
    jobs = []
    if force or stale(LIB):
        jobs.append(_compile(LIB, [], verbose))
    if force or stale(STRESS_LIB):
        jobs.append(_compile(STRESS_LIB, ["-DSVB_LB_STRESS"], verbose))
    for j in jobs:
        if j.wait() != 0:
            raise subprocess.CalledProcessError(j.returncode, j.args)
    if jobs:
        if os.path.lexists(LINK):
            os.remove(LINK)
        os.symlink(os.path.basename(LIB), LINK)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
