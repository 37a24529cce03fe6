#!/bin/bash
# TEST INFRASTRUCTURE: builds the kernels for the CPU fiber emulator (logic debugging only).
set -e
cd "$(dirname "$0")/../.."
SRCS=$(ls bzip2-rust_amd/csrc/*.hip)
ARGS=""
for s in $SRCS; do ARGS="$ARGS -x c++ $s"; done
g++ -O1 -g -std=c++17 -fPIC -shared -w -DSCAN_SEG=16 -DBS_C=2048 -I tests/emu $ARGS -x c++ tests/emu/hip_emu.cpp \
    -Wl,--unresolved-symbols=ignore-all -o tests/emu/libbzx_emu.so
echo built tests/emu/libbzx_emu.so
# second build with only two rank rounds: buckets stay open after the rounds although some of their groups were
# resolved in them, and the general sorter has to finish such blocks (with the shipped 14 rounds that takes a repeat
# of more than half a million symbols); see test_emu_short_rank_rounds
if [ "$1" != "fast" ]; then
g++ -O1 -g -std=c++17 -fPIC -shared -w -DSCAN_SEG=16 -DBS_C=2048 -DRK_ROUNDS=2 -I tests/emu $ARGS -x c++ tests/emu/hip_emu.cpp \
    -Wl,--unresolved-symbols=ignore-all -o tests/emu/libbzx_emu_rk2.so
echo built tests/emu/libbzx_emu_rk2.so
fi
