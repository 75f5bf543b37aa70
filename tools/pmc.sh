#!/bin/bash
# usage: tools/pmc.sh "<counters>" tag  -- runs lat2.py (NQ=1024) under rocprofv3 --pmc, prints per-kernel counter sums for k_score_*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$2
rm -rf $out
NQ=1024 timeout -k 10 250 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $out -o pmc -- python3 tools/lat2.py > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"]
    if "k_score" not in k and "k_merge_topk" not in k: continue
    k = "waves" if "k_score_waves" in k else "slices" if "k_score_slices" in k else "merge"
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[(k, row["Counter_Name"])] += 1
for k in acc:
    for c, v in acc[k].items():
        print(k, c, v / max(n[(k, c)], 1))
PY
