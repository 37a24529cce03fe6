#!/bin/bash
# Builds libbzx.so (HIP kernels + C ABI) for gfx950, in-tree.
#   build.sh          the product
#   build.sh diag     also libbzx_diag.so: the same sources with -DBZX_DIAG (phase timers, bzx_dbg_* helpers) for the
#                     tests/gpu_probe_* scripts (select it with BZX_LIB=.../libbzx_diag.so); never loaded by tests or bench
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRCS=$(ls csrc/*.hip)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -I ../include"
if [ "$1" = "diag" ]; then
    shift
    $HIPCC $FLAGS -DBZX_DIAG $SRCS -o libbzx_diag.so "$@"
    echo "built $(pwd)/libbzx_diag.so"
    exit 0
fi
$HIPCC $FLAGS $SRCS -o libbzx.so "$@"
echo "built $(pwd)/libbzx.so"
# thin command line over the C ABI; host code only
g++ -O2 -std=c++17 -Wall ../tools/bzx.cpp -I ../include -L . -lbzx -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -o bzx
echo "built $(pwd)/bzx"
