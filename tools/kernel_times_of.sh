#!/bin/bash
# Per-kernel durations (rocprofv3 kernel trace) of any tool:  tools/kernel_times_of.sh tools/intersect_bench.py --log2n 27
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/kt
rm -rf $OUT && mkdir -p $OUT
PROG=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $PROG "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "svb::" in r["Name"]:
        print(f'{r["Name"].split("(")[0].replace("void svb::", "")[:58]:58s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f} us  max {float(r["MaxNs"]) / 1e3:8.1f} us')
PY
