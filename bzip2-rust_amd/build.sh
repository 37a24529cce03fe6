#!/bin/bash
# Builds libbzx.so (HIP kernels + C ABI) for gfx950, in-tree.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRCS=$(ls csrc/*.hip)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    -I ../include $SRCS -o libbzx.so "$@"
echo "built $(pwd)/libbzx.so"
