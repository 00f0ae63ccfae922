#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

gfx950 corrections: FETCH_SIZE counts 64 B per 128-B read request of a wide
coalesced stream, i.e. exactly half the bytes (checked here on k_zigzag<0> and
k_size, whose reads are a known 4 B per value) -> doubled; WRITE_SIZE is exact
for 16-B-per-lane stores.  Both are in KiB.

    python tools/traffic_from_pmc.py FETCH.csv WRITE.csv > profiles/rNN_pmc_traffic.json
"""
import collections, csv, json, re, sys
import numpy as np


def per_kernel(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "svb::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void svb::", "").replace("svb::", "")
            d[k].append(float(r["Counter_Value"]))
    # timing-only ablation launches move fewer bytes: keep the full launches
    return {k: float(np.median([x for x in v if x >= 0.95 * max(v)])) for k, v in d.items()}


if __name__ == "__main__":
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        out[k] = {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "hbm_bytes": int((2 * fk + wk) * 1024)}
    json.dump({"note": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, n = 2^28 (2^26 for the 0124 zigzag kernels), "
                       "tools/microbench.py inputs; a kernel launched on several workloads is reported for the "
                       "largest one (the delta decoders: deltas of full-width values, ~3.75 data bytes per value)",
               "kernels": out}, sys.stdout, indent=1)
