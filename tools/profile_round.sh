#!/bin/bash
# Collect the profiles that get committed under profiles/ (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench.py run
#   2. a kernel trace and two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/microbench.py
#      (tools/roofline_table.py joins them into rNN_roofline.csv)
# Results land in gpurun_out/prof_*; tools/traffic_from_pmc.py turns 2. into rNN_pmc_traffic.json.
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $ROOT/bench.py > $OUT/prof_bench.log 2>&1
echo "bench trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_micro -- python3 $ROOT/tools/microbench.py --rounds 2 --skip ablate > $OUT/prof_micro.log 2>&1
echo "microbench trace done"
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python3 $ROOT/tools/microbench.py --rounds 2 --skip ablate > $OUT/prof_fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python3 $ROOT/tools/microbench.py --rounds 2 --skip ablate > $OUT/prof_write.log 2>&1
echo "write pass done"
