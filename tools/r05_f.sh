#!/bin/bash
set -o pipefail
out=gpurun_out/r05f; mkdir -p $out; rm -f $out/*.log
timeout -k 10 400 python -m pytest tests/test_gpu_pagerank.py tests/test_gpu_topic_sensitive.py -x -q -m gpu > $out/pytest.log 2>&1; tail -5 $out/pytest.log
grep -q passed $out/pytest.log || exit 1
for k in 1 2; do
N=1048576 E=5000000 K=$k R=5 OPTSETS="pr.big_blocks=0;pr.big_blocks=1;pr.big_blocks=1,pr.blocks_per_cu=4;pr.big_blocks=1,pr.blocks_per_cu=6;pr.big_blocks=0,pr.blocks_per_cu=3" timeout -k 10 200 python tools/pr_exp.py 2>&1 | grep lib= | cut -c1-150 >> $out/big.log || exit 1
N=10000000 E=50000000 K=$k R=3 OPTSETS="pr.big_blocks=0;pr.big_blocks=1" timeout -k 10 200 python tools/pr_exp.py 2>&1 | grep lib= | cut -c1-150 >> $out/big.log
done
cat $out/big.log
