#!/usr/bin/env python3
"""Print one line per kernel: VGPRs, SGPRs, scratch, LDS, occupancy (hipcc remarks)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17"] + sys.argv[1:] + [ "-I", os.path.join(ROOT, "include"),
       "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null",
       os.path.join(ROOT, "streamvbyte_amd/csrc/svb_api.hip")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":",1)[1].strip()}; rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void svb::", "")
    print(f"{name:45s} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>3} sgpr={r.get('TotalSGPRs','?'):>4} "
          f"scratch={r.get('ScratchSize [bytes/lane]','?'):>4} lds={r.get('LDS Size [bytes/block]','?'):>6} occ={r.get('Occupancy [waves/SIMD]','?')}")
