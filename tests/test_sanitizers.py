"""ASan + UBSan build of the host-side code (CPU only): the library's host codec and the
scalar oracle, run on buffers of exactly the promised sizes.  The reference ships the same
two sanitizer configurations for its own code (CMakeLists.txt:63,82-94,149-153 there)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_codec_and_oracle_under_asan_ubsan():
    exe = os.path.join(tempfile.mkdtemp(prefix="svb_san_"), "host_codec_sanitize")
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
    obj = exe + "_oracle.o"
    subprocess.check_call(["gcc", "-std=c99", "-c", *flags, "-o", obj, os.path.join(ROOT, "oracle", "svb_oracle.c")])
    subprocess.check_call(["g++", "-std=c++17", *flags, "-o", exe,
                           os.path.join(ROOT, "tests", "c", "host_codec_sanitize.cpp"), obj])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "clean under ASan + UBSan" in p.stdout
