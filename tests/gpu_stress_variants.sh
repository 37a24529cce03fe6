#!/bin/bash
# Ad-hoc GPU stress (run it on the GPU box, under `timeout`): builds of the library with extreme parameters that force
# the rarely used paths of the sorter, each checked against libbz2 by the fuzzers, piece by piece on two file corpora
# and on three bulk inputs.  Round 2: all five variants without a mismatch (gpurun_out/r2_stress.log).
#   sA  two rank rounds           -> buckets stay open after the rounds, the general sorter finishes every such block
#   sB  give up after one round   -> tiny give-up depths, long chains of doubling rounds
#   sC  split limits 39 bits / 4  -> oversized bins left as one group in almost every block of real data
#   sD  ring of 4 oversized bins  -> blocks refused by the split kernel, sorted from scratch on the side stream
#   sE  eight rank arrays         -> blocks with and without a rank array in one batch
cd "$(dirname "$0")/.."
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -I include"
S=bzip2-rust_amd/csrc/*.hip
[ -f bzip2-rust_amd/libbzx_sA.so ] || $H -DRK_ROUNDS=2 $S -o bzip2-rust_amd/libbzx_sA.so
[ -f bzip2-rust_amd/libbzx_sB.so ] || $H -DBS_ROUNDS=1 -DRK_ROUNDS=3 $S -o bzip2-rust_amd/libbzx_sB.so
[ -f bzip2-rust_amd/libbzx_sC.so ] || $H -DBS_MAX_DEPTH=39 -DBS_MAX_SPLITS=4 $S -o bzip2-rust_amd/libbzx_sC.so
[ -f bzip2-rust_amd/libbzx_sD.so ] || $H -DBS_MAX_BIG=4 $S -o bzip2-rust_amd/libbzx_sD.so
[ -f bzip2-rust_amd/libbzx_sE.so ] || $H -DBZX_STRESS_FEW_RANK_ARRAYS $S -o bzip2-rust_amd/libbzx_sE.so
mkdir -p gpurun_out
LOG=gpurun_out/stress.log
rm -f $LOG
for v in sA sB sC sD sE; do
  export BZX_LIB=bzip2-rust_amd/libbzx_$v.so
  echo "== $v" >> $LOG
  timeout -k 10 150 python tests/gpu_probe_fuzz.py ${SEED:-31} ${NSMALL:-1200} 2>&1 | tail -n 1 >> $LOG
  timeout -k 10 150 python tests/gpu_probe_fuzz_big.py $((${SEED:-31}+1)) ${NBIG:-200} 2>&1 | tail -n 1 >> $LOG
  for c in hdr py; do timeout -k 10 200 python tests/gpu_probe_pieces.py $c 128 2>&1 | tail -n 1 >> $LOG; done
  timeout -k 10 150 python tests/gpu_probe_bsort.py 128 so,zeros,text 2>&1 | grep -o "^[a-z-]* \|blocks-left [0-9]* from-scratch [0-9]*\|parity=[A-Z]*" | tr "\n" " " >> $LOG
  echo >> $LOG
done
cat $LOG
