#!/usr/bin/env python3
"""Per-kernel achieved HBM bandwidth from rocprofv3 output of ONE workload (tools/microbench.py):
durations from a --kernel-trace pass, HBM bytes from the two --pmc passes.

    python tools/roofline_table.py KERNEL_TRACE.csv FETCH.csv WRITE.csv > profiles/rNN_roofline.csv
"""
import csv, re, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from traffic_from_pmc import per_kernel

PEAK = 8000.0  # GB/s, MI355X HBM3E


def short(name):
    return re.sub(r"\(.*", "", name).replace("void svb::", "").replace("svb::", "")


import collections
import numpy as np
raw = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):  # *_kernel_trace.csv: one row per dispatch
    if "svb::" in r["Kernel_Name"]:
        raw[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
dur = {}
for k, v in raw.items():
    # the full-size launches of a kernel are its longest (the workload also holds ablations and one-tile tails)
    full = [x for x in v if x >= 0.8 * max(v)]
    dur[k] = (len(v), float(np.median(full)), float(max(v)))
f, w = per_kernel(sys.argv[2], "FETCH_SIZE"), per_kernel(sys.argv[3], "WRITE_SIZE")
out = csv.writer(sys.stdout)
out.writerow(["kernel", "calls", "full_size_median_us", "max_us", "hbm_read_MB", "hbm_write_MB", "hbm_GBps", "frac_of_8TBps"])
for k in sorted(dur):
    calls, avg, mx = dur[k]
    rd, wr = 2 * f.get(k, 0.0) * 1024, w.get(k, 0.0) * 1024
    gbps = (rd + wr) / avg if avg else 0.0  # bytes per ns = GB/s
    out.writerow([k, calls, f"{avg / 1e3:.1f}", f"{mx / 1e3:.1f}", f"{rd / 1e6:.1f}", f"{wr / 1e6:.1f}", f"{gbps:.0f}", f"{gbps / PEAK:.3f}"])
