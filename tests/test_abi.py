"""C-ABI checks that need no GPU: the library loads and exports exactly what
include/*.h declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")


def declared_functions():
    names = []
    for h in sorted(os.listdir(INCLUDE)):
        text = open(os.path.join(INCLUDE, h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"static inline.*?\n}\n", "", text, flags=re.S)
        for m in re.finditer(r"^[\w\s\*]+?\b(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M):
            names.append(m.group(1))
    return sorted(set(names))


def test_headers_declare_the_reference_symbols():
    names = declared_functions()
    ref14 = ["streamvbyte_compressedbytes", "streamvbyte_compressedbytes_0124", "streamvbyte_decode",
             "streamvbyte_decode_0124", "streamvbyte_delta_decode", "streamvbyte_delta_encode",
             "streamvbyte_encode", "streamvbyte_encode_0124", "streamvbyte_validate_stream",
             "streamvbyte_validate_stream_0124", "zigzag_decode", "zigzag_delta_decode",
             "zigzag_delta_encode", "zigzag_encode"]
    for n in ref14:
        assert n in names
    assert any(n.startswith("svbgpu_") for n in names)


def test_library_exports_every_declared_symbol():
    import streamvbyte_amd
    lib = streamvbyte_amd.load()  # raises if the .so is missing or lacks a bound symbol
    out = subprocess.run(["nm", "-D", "--defined-only", streamvbyte_amd.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for n in declared_functions():
        assert n in exported, n
        assert hasattr(lib, n)
    assert set(streamvbyte_amd.SIGNATURES) == set(declared_functions())


def test_host_side_helpers_without_gpu():
    import streamvbyte_amd as svb
    lib = svb.load()
    assert svb.streamvbyte_max_compressedbytes(0) == 16
    assert svb.streamvbyte_max_compressedbytes(5) == 2 + 20 + 16
    sizes = [lib.svbgpu_workspace_bytes(n) for n in (0, 1, 1024, 1 << 20, 1 << 28, 2**32 - 1)]
    assert sizes == sorted(sizes) and sizes[0] >= 256 and all(s % 256 == 0 for s in sizes)
    assert lib.svbgpu_strerror(0) == b"ok"
    assert b"workspace" in lib.svbgpu_strerror(-3)
    # argument checks come before any device work
    assert lib.svbgpu_decode(7, 0, None, None, 4, 0, None, 0, None, None) == -1
    assert lib.svbgpu_encode(0, 99, None, 4, None, 0, None, 0, None, None) == -1


def test_missing_library_is_an_error(monkeypatch):
    import streamvbyte_amd._lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libstreamvbyte.so.2")
    with pytest.raises(RuntimeError):
        L.load()


C_PROGRAMS = [("tests", "c", "dropin_roundtrip.c"), ("examples", "example.c"), ("examples", "writeseq.c")]


def _build_c_caller(tmp_path, source=C_PROGRAMS[0]):
    import streamvbyte_amd
    exe = str(tmp_path / os.path.splitext(source[-1])[0])
    libdir = os.path.dirname(streamvbyte_amd.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", INCLUDE,
           os.path.join(ROOT, *source), "-o", exe,
           "-L", libdir, "-l:libstreamvbyte.so.2", f"-Wl,-rpath,{libdir}",
           "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.parametrize("source", C_PROGRAMS, ids=lambda s: s[-1])
def test_plain_c_caller_compiles_and_links(tmp_path, source):
    """C99 programs using only the headers under include/ build against the library."""
    _build_c_caller(tmp_path, source)


@pytest.mark.gpu
@pytest.mark.parametrize("source", C_PROGRAMS, ids=lambda s: s[-1])
def test_plain_c_caller_runs(tmp_path, source):
    exe = _build_c_caller(tmp_path, source)
    args = [exe, str(tmp_path / "seq.bin")] if source[-1] == "writeseq.c" else [exe]
    out = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
