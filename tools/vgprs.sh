#!/bin/bash
# tools/vgprs.sh <file.hip> [regex] [extra flags]: registers / scratch / occupancy of the file's kernels (CPU only: hipcc cross-compiles)
f=$1; re=${2:-.}; shift; shift
cd "$(dirname "$0")/../spaghettisearch_amd/csrc"
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I/opt/rocm/include "$@" \
    -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/vgprs_$$.o 2>&1 |
  awk '/Function Name:/{n=$5} / VGPRs:/{v=$4} /ScratchSize/{s=$5} /TotalSGPRs/{g=$4} /LDS Size/{l=$6} /Occupancy/{o=$5; print n, "sgpr", g, "vgpr", v, "scratch", s, "occ", o}' |
  c++filt | grep -E "$re"
rm -f /tmp/vgprs_$$.o
