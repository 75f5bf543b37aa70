#!/bin/bash
# Diagnostic build of the library (-DSS_DIAG: in-kernel counters/stamps, printed by ss_scorer_destroy) beside the product
# build: spaghettisearch_amd/libspaghetti_rank_diag.so; use it with SS_LIB_PATH=... (never timed, never shipped).
set -e
cd "$(dirname "$0")/../spaghettisearch_amd/csrc"
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    -I/opt/rocm/include -DSS_DIAG $EXTRA -shared -Wl,-soname,libspaghetti_rank.so -o ../libspaghetti_rank_diag.so *.hip -L/opt/rocm/lib -lrccl
