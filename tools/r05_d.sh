#!/bin/bash
set -o pipefail
out=gpurun_out/r05d; mkdir -p $out
for lib in "" sc4 sc1; do
  if [ -n "$lib" ]; then export SS_LIB_PATH=$PWD/spaghettisearch_amd/libspaghetti_rank_$lib.so; else unset SS_LIB_PATH; fi
  CFG="4096:13" timeout -k 10 300 python tools/tfidf_exp.py 2>&1 | grep blocks= | sed "s/^/lib=${lib:-product} /" >> $out/tfidf_abl.log || exit 1
done
unset SS_LIB_PATH
cat $out/tfidf_abl.log
timeout -k 10 600 python -m pytest tests/test_gpu_bench_rehearsal.py tests/test_gpu_pagerank.py tests/test_gpu_comm.py -x -q -m gpu > $out/pytest.log 2>&1; tail -15 $out/pytest.log
