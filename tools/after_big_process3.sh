#!/bin/bash
# bench.py's settle wait behind a large exiting process: with it (default 8 s) and without (--settle 0)
big() { python3 - <<'PY'
import torch
x = [torch.empty(8 << 30, dtype=torch.uint8, device="cuda") for _ in range(16)]
for t in x: t.fill_(1)
torch.cuda.synchronize()
PY
}
digest() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('settle_s', d['settle_s'], 'ms_per_sweep', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'gather probe', round(d['roofline']['gather_ceiling_ms'],4))"; }
for s in 8 0 4 8 0; do
  big; echo "== behind a 128 GB process, --settle $s"
  python3 bench.py --workload pagerank --no-cpu-baseline --no-config2 --steps 20 --warmup 5 --settle $s 2>/dev/null | digest
done
