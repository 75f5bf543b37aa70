#!/bin/bash
# round 5, first GPU trip: new k_pr_sweep (4 waves/SIMD under the stagger), stagger vectors with four real rounds, config-2 classes
set -o pipefail
out=gpurun_out/r05a; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_pagerank.py -x -q -m gpu > $out/pytest_pr.log 2>&1 || { tail -20 $out/pytest_pr.log; exit 1; }
tail -2 $out/pytest_pr.log
code() { echo $((10 + $1 + 6*$2 + 36*$3 + 216*$4)); }
sets="pr.stagger=1;pr.stagger=0;pr.blocks_per_cu=3"
for v in "0 1 2 3" "0 1 2 4" "0 1 2 5" "0 2 1 3" "0 3 1 4" "0 1 3 5" "0 2 4 5" "0 2 3 5" "1 2 3 4" "0 1 3 4" "0 3 1 5" "0 4 1 3" "0 4 2 5" "0 1 4 5" "0 4 1 2"; do sets="$sets;pr.stagger=$(code $v)"; done
OPTSETS="$sets" R=5 timeout -k 10 500 python tools/pr_exp.py > $out/stagger.log 2>&1 || { tail -5 $out/stagger.log; exit 1; }
cat $out/stagger.log | cut -c1-150
# config 2: classes of k_pr_sweep_n<1> alone
export SS_LIB_PATH=$PWD/spaghettisearch_amd/libspaghetti_rank_kmask.so
for m in 0xFFFFFFFF 0x0 0x100 0x200 0x400 0x2000 0x300 0x700; do
  SS_PR_KIND_MASK=$m N=1048576 E=5000000 K=1 R=5 timeout -k 10 100 python tools/pr_exp.py 2>&1 | grep lib= | sed "s/^/mask $m: /" | cut -c1-140 >> $out/c2_classes.log || exit 1
done
cat $out/c2_classes.log
