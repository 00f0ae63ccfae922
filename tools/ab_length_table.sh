#!/bin/bash
# A/B: length table in LDS (the reference's 256-byte table, looked up per key) against the arithmetic
# form (the product) in k_decode.  Build the variant library here, on the CPU box:
#     tools/ab_length_table.sh build
# then on the GPU box:
#     tools/ab_length_table.sh run
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ALT=$ROOT/streamvbyte_amd/libstreamvbyte_lds_table.so
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSVB_LENGTH_TABLE_LDS -Wl,-soname,libstreamvbyte.so.2 \
      -I "$ROOT/include" -o "$ALT" "$ROOT/streamvbyte_amd/csrc/svb_api.hip"
  echo "built $ALT"
else
  for i in 1 2 3; do
    echo "arithmetic: $(python3 "$ROOT/bench.py" --quick 2>/dev/null | python3 -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["parity"], r["roofline"]["kernel_ms"], r["roofline"]["frac"])')"
    echo "lds table:  $(SVB_LIB_PATH=$ALT python3 "$ROOT/bench.py" --quick 2>/dev/null | python3 -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["parity"], r["roofline"]["kernel_ms"], r["roofline"]["frac"])')"
  done
fi
