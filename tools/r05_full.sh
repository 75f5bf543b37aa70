#!/bin/bash
# full GPU test suite, then the driver-shaped bench line and the default one
set -o pipefail
tag=${1:-r05a}
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $out/${tag}_gpu_tests.log 2>&1; tail -5 $out/${tag}_gpu_tests.log
grep -q " passed" $out/${tag}_gpu_tests.log || exit 1
grep -q "failed" $out/${tag}_gpu_tests.log && exit 1
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver_shape.json 2> $out/${tag}_bench_driver_shape.err || { tail -5 $out/${tag}_bench_driver_shape.err; exit 1; }
python - $out/${tag}_bench_driver_shape.json <<'PY'
import json, sys
o = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver shape:", o["value"], o["ms_per_step"], o["roofline"]["frac"], json.dumps(o["summary"]))
print("tail of the line:", json.dumps(o)[-600:])
PY
