#!/bin/bash
set -o pipefail
out=gpurun_out/r05i; mkdir -p $out; rm -f $out/*.log
timeout -k 10 600 python -m pytest tests/test_gpu_tfidf.py tests/test_gpu_index_update.py tests/test_gpu_shard_index.py -x -q -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
grep -q " passed" $out/pytest.log || exit 1
grep -q "failed" $out/pytest.log && exit 1
CFG="4096:13" timeout -k 10 300 python tools/tfidf_exp.py 2>&1 | grep blocks= | tee -a $out/tfidf.log
export SS_LIB_PATH=$PWD/spaghettisearch_amd/libspaghetti_rank_scph.so
CFG="4096:13" timeout -k 10 300 python tools/tfidf_exp.py > $out/scph.log 2>&1; grep -E "phases|blocks=" $out/scph.log | tail -2
