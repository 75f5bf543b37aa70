#!/bin/bash
# A/B builds beside the product library: tools/build_variant.sh <name> "<extra hipcc flags>"
#   -> spaghettisearch_amd/libspaghetti_rank_<name>.so ; use it with SS_LIB_PATH=... (experiments only, never shipped)
set -e
name=$1; shift
cd "$(dirname "$0")/../spaghettisearch_amd/csrc"
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    -I/opt/rocm/include $@ -shared -Wl,-soname,libspaghetti_rank.so -o ../libspaghetti_rank_$name.so *.hip -L/opt/rocm/lib -lrccl
