#!/bin/bash
out=gpurun_out/r05j; mkdir -p $out; rm -f $out/*.log
for lib in "" t512p8 t512p8w t1024p4 t512p16 t256p16; do
  if [ -n "$lib" ]; then export SS_LIB_PATH=$PWD/spaghettisearch_amd/libspaghetti_rank_$lib.so; else unset SS_LIB_PATH; fi
  CFG="4096:13 2048:13" timeout -k 10 300 python tools/tfidf_exp.py 2>&1 | grep blocks= | sed "s/^/lib=${lib:-product} /" | cut -c1-140 >> $out/geom.log
done
cat $out/geom.log
