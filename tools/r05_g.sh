#!/bin/bash
set -o pipefail
out=gpurun_out/r05g; mkdir -p $out; rm -f $out/*.log
timeout -k 10 700 python -m pytest tests/test_gpu_score.py -x -q -m gpu > $out/pytest.log 2>&1; tail -4 $out/pytest.log
grep -q " passed" $out/pytest.log || exit 1
grep -q "failed" $out/pytest.log && exit 1
OPTS="score.small=0" timeout -k 10 400 python tools/score_mixed.py > $out/mixed.log 2>&1; grep -v "score.wave=0" $out/mixed.log | cut -c1-170
