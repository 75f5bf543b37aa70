#!/bin/bash
set -o pipefail
out=gpurun_out/r05e; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_pagerank.py -x -q -m gpu -k "inside_one_launch" > $out/pytest.log 2>&1; tail -8 $out/pytest.log
grep -q passed $out/pytest.log || exit 1
for k in 1 2; do
N=1048576 E=5000000 K=$k R=5 OPTSETS="pr.persistent=0;pr.persistent=1;pr.persistent=2;pr.persistent=2,pr.persistent_blocks=2;pr.persistent=2,pr.persistent_blocks=3;pr.persistent=2,pr.persistent_blocks=1" timeout -k 10 200 python tools/pr_exp.py 2>&1 | grep lib= | cut -c1-150 >> $out/persist.log || exit 1
done
N=10000000 E=50000000 K=1 R=3 OPTSETS="pr.persistent=0;pr.persistent=2" timeout -k 10 200 python tools/pr_exp.py 2>&1 | grep lib= | cut -c1-150 >> $out/persist.log
cat $out/persist.log
