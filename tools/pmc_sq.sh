#!/bin/bash
# Where a kernel's wave cycles go: two rocprofv3 --pmc passes (8 SQ counters each) over one
# microbench selection:  tools/pmc_sq.sh encode [log2n]
# or over another script of tools/:  tools/pmc_sq.sh batch_mix.py 8
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_sq
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [[ "$1" == *.py ]]; then PROG=("$ROOT/tools/$1" "${@:2}"); else PROG=("$ROOT/tools/microbench.py" --rounds 2 --only "$1" --log2n "${2:-28}"); fi
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -- python3 "${PROG[@]}" > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p2 -- python3 "${PROG[@]}" > $OUT/p2.log 2>&1
python3 $ROOT/tools/pmc_table.py $(find $OUT -name '*counter_collection.csv')
