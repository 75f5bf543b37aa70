#!/bin/bash
# does a process that starts right behind a large exiting one get slower memory?  (tools/pr_exp.py: sweep ms and gather probe)
big() { python3 - <<'PY'
import torch
x = [torch.empty(8 << 30, dtype=torch.uint8, device="cuda") for _ in range(16)]     # 128 GB touched and dropped at exit
for t in x: t.fill_(1)
torch.cuda.synchronize()
PY
}
echo "fresh box:";            R=2 python3 tools/pr_exp.py 2>&1 | grep lib= | cut -c1-150
big; echo "right behind a 128 GB process:"; R=2 python3 tools/pr_exp.py 2>&1 | grep lib= | cut -c1-150
echo "the next one:";         R=2 python3 tools/pr_exp.py 2>&1 | grep lib= | cut -c1-150
big; sleep 5; echo "5 s behind a 128 GB process:"; R=2 python3 tools/pr_exp.py 2>&1 | grep lib= | cut -c1-150
