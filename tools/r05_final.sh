#!/bin/bash
# final evidence of the round: full GPU suite, the default bench line, the driver-shaped bench line
tag=${1:-r05d}
out=gpurun_out; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -q -m gpu > $out/${tag}_gpu_tests.log 2>&1; tail -4 $out/${tag}_gpu_tests.log
timeout -k 10 400 python bench.py > $out/${tag}_bench_full.json 2> $out/${tag}_bench_full.err || { tail -5 $out/${tag}_bench_full.err; exit 1; }
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver_shape.json 2> $out/${tag}_bench_driver_shape.err || { tail -5 $out/${tag}_bench_driver_shape.err; exit 1; }
python - $out/${tag}_bench_full.json $out/${tag}_bench_driver_shape.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    o = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, json.dumps(o["summary"]))
PY
