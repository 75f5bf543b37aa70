#!/bin/bash
# wave-kernel build variants (tools/build_variant_one.sh w<name> score_wave.hip ...) at config 3: wall ms per batch back to back (tools/score_wall.py)
for v in ${VARIANTS:-product wse256 wse272 wse288 wse320 wse352 product}; do
  if [ $v = product ]; then unset SS_LIB_PATH; else export SS_LIB_PATH=spaghettisearch_amd/libspaghetti_rank_$v.so; fi
  echo "== $v"; timeout -k 10 200 python3 tools/score_wall.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-220
done
