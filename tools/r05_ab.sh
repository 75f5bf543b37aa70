#!/bin/bash
# A/B on one box: the product library against spaghettisearch_amd/libspaghetti_rank_old.so (the previous commit's scoring files), config-3 wall per batch and the
# tail / mixed / half-half batches, twice each in alternation
for rep in 1 2; do
  for v in old product; do
    if [ $v = product ]; then unset SS_LIB_PATH; else export SS_LIB_PATH=spaghettisearch_amd/libspaghetti_rank_$v.so; fi
    echo "== $v (run $rep)"; timeout -k 10 200 python3 tools/score_wall.py 2>&1 | tail -1 | cut -c1-200
    timeout -k 10 300 python3 tools/small_v2.py 2>&1 | grep "routing off"
  done
done
