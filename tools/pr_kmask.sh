#!/bin/bash
# time each work class of the sweep alone (experiment builds with -DSS_PR_EXP_KINDMASK, tools/build_variant.sh):
#   LIBS="kmask km_nost" MASKS="0x800 0x400" tools/pr_kmask.sh out.log
out=${1:-gpurun_out/pr_kmask.log}
for lib in ${LIBS:-kmask}; do
export SS_LIB_PATH=$PWD/spaghettisearch_amd/libspaghetti_rank_$lib.so
for m in ${MASKS:-0xFFFFFFFF 0x0 0x100 0x200 0x400 0x800 0x1000}; do
  SS_PR_KIND_MASK=$m R=3 timeout -k 10 200 python tools/pr_exp.py 2>&1 | grep -E "lib=" | sed "s/^/mask $m: /" | cut -c1-130 >> $out || exit 1
done
done
