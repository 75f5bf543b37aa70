#!/bin/bash
# A/B library that differs from the product in ONE translation unit: tools/build_variant_one.sh <name> <file.hip> "<extra hipcc flags>"
#   -> spaghettisearch_amd/libspaghetti_rank_<name>.so (the other objects are the product's: run make first); SS_LIB_PATH=... selects it
set -e
name=$1; f=$2; shift; shift
cd "$(dirname "$0")/../spaghettisearch_amd/csrc"
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I/opt/rocm/include $@ -c $f -o /tmp/var_${name}.o
objs=$(ls *.o | grep -v "^${f%.hip}.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-soname,libspaghetti_rank.so -o ../libspaghetti_rank_$name.so $objs /tmp/var_${name}.o -L/opt/rocm/lib -lrccl
