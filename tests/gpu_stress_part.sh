#!/bin/bash
# usage: stress_part.sh <log> variants...
cd "$(dirname "$0")/.."
LOG=$1; shift
rm -f $LOG
for v in "$@"; do
  export BZX_LIB=bzip2-rust_amd/libbzx_$v.so
  echo "== $v" >> $LOG
  timeout -k 10 120 python tests/gpu_probe_fuzz.py 131 700 2>&1 | tail -n 1 >> $LOG
  timeout -k 10 120 python tests/gpu_probe_fuzz_big.py 132 100 2>&1 | tail -n 1 >> $LOG
  for c in hdr py; do timeout -k 10 150 python tests/gpu_probe_pieces.py $c 64 2>&1 | tail -n 1 >> $LOG; done
  timeout -k 10 120 python tests/gpu_probe_bsort.py 128 so,zeros,text 2>&1 | grep -o "^[a-z-]* \|blocks-left [0-9]* from-scratch [0-9]*\|parity=[A-Z]*" | tr "\n" " " >> $LOG
  echo >> $LOG
done
cat $LOG
