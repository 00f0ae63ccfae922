"""The hot kernels sit at their register limits (128 VGPRs for two workgroups of eight waves per CU, 106
SGPRs): one more scalar in the source has already pushed the pipelined encoder into scratch spills and cost
10 % of its speed without failing anything.  This compiles the device code with the compiler's resource
remarks and holds the line: no scratch in the product instantiations of the streaming kernels, and the
occupancy their launch shapes count on."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hot_kernels_do_not_spill():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")],
                         capture_output=True, text=True, timeout=900).stdout
    rows = {}
    for line in out.splitlines():
        m = re.match(r"(\S.*?)\s+vgpr=\s*(\d+)\s+agpr=\s*(\d+)\s+sgpr=\s*(\d+)\s+scratch=\s*(\d+)\s+lds=\s*(\d+)\s+occ=(\d+)", line)
        if m:
            rows[m.group(1).strip()] = dict(vgpr=int(m.group(2)), scratch=int(m.group(5)), occ=int(m.group(7)))
    assert rows, out[-2000:]
    hot = [k for k in rows if re.match(r"(svb::)?k_(decode<|encode_pipe<|delta_decode<|running_sum<|zigzag<|size<|key_sums<|validate_sum<|decode_batch2<|encode_batch2<)", k)]
    assert len(hot) > 20, sorted(rows)[:40]
    for k in hot:
        if k.startswith("k_encode_pipe<") and k.rstrip(">").endswith("true"):
            continue  # the diagnostic instantiation with in-kernel stamps
        assert rows[k]["scratch"] == 0, (k, rows[k])
    for k in hot:
        if k.startswith("k_encode_pipe<"):
            assert rows[k]["occ"] >= 4, (k, rows[k])   # two workgroups of eight waves per CU
        if k.startswith("k_decode<") or k.startswith("k_decode_batch2<"):
            assert rows[k]["occ"] >= 7, (k, rows[k])   # seven workgroups of four waves per CU (the LDS limit)
        if k.startswith("k_encode_batch2<"):
            assert rows[k]["occ"] >= 6, (k, rows[k])   # two tiles of values in registers took it to four and cost 15 %
    # the tile-at-a-time encoder on full tiles of four wave tiles a wave (the default from 2^27 values up, and for
    # differential input from 2^25): two workgroups of eight waves per CU, nothing spilled
    full_tile = [k for k in rows if re.match(r"(svb::)?k_encode<[01], (true|false), (true|false), 4, true, ", k)]
    assert len(full_tile) >= 16, sorted(rows)[:40]
    for k in full_tile:
        assert rows[k]["scratch"] == 0 and rows[k]["occ"] >= 4, (k, rows[k])
    for k in rows:
        if re.match(r"(svb::)?k_(intersect_batch|delta_intersect|size_batch)<", k):
            assert rows[k]["scratch"] == 0, (k, rows[k])
        if re.match(r"(svb::)?k_stream_intersect<", k):  # five waves per SIMD at the price of two or three spilled dwords
            assert rows[k]["scratch"] <= 16 and rows[k]["occ"] >= 5, (k, rows[k])
