#!/bin/bash
# tools/after_big_process.sh, second part: can a process that starts behind a large exiting one avoid the slow state?
big() { python3 - <<'PY'
import torch
x = [torch.empty(8 << 30, dtype=torch.uint8, device="cuda") for _ in range(16)]
for t in x: t.fill_(1)
torch.cuda.synchronize()
PY
}
run() { echo "== $1"; shift; env "$@" R=2 PROBE=1 python3 tools/pr_exp.py 2>&1 | grep lib= | cut -c1-170; }
run "fresh box" X=1
big; run "right behind a 128 GB process" X=1
big; run "behind one, sleeping 8 s before the first GPU call" PRE_SLEEP=8
big; run "behind one, 200 GB taken and returned first" RINSE_GB=200
big; run "behind one, graph and state rebuilt twice later in the process" REBUILD=2 REBUILD_SLEEP=3
big; run "behind one (control)" X=1
run "the next" X=1
