#!/bin/bash
# A/B of non-temporal streaming loads / stores: variant libraries built with -DSVB_NT_LOAD / -DSVB_NT_STORE
# (streamvbyte_amd/libstreamvbyte_nt_{load,store,both}.so) against the product, one microbench process each.
#   ORDER="default nt_both" ONLY=decode,encode tools/nt_ab.sh
for v in ${ORDER:-default nt_load nt_store nt_both}; do
  if [ "$v" = default ]; then unset SVB_LIB_PATH; else export SVB_LIB_PATH=$PWD/streamvbyte_amd/libstreamvbyte_$v.so; fi
  echo "== $v"
  timeout -k 5 200 python tools/microbench.py --only ${ONLY:-decode,encode,misc,zz,delta} --rounds 7 2>&1 | grep -E "of 8TB" | grep -vE "chunk|early|q2w8|pipe2|twopass|noshortcut|mixed" || exit 1
done
