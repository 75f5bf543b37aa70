#!/bin/bash
# usage: tools/pmc_pr.sh "<counters>" tag   -- rocprofv3 --pmc over the PageRank half of bench.py, per-dispatch averages for k_pr_sweep<16>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmcpr_$2
rm -rf $out
timeout -k 10 250 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $out -o pmc -- python3 bench.py --workload pagerank --no-cpu-baseline --steps 10 --warmup 2 > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for row in csv.DictReader(open(f[0])):
    if "k_pr_sweep<16>" not in row["Kernel_Name"]: continue
    acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]].add(row["Dispatch_Id"])
for c, v in acc.items():
    print(c, round(v / max(len(n[c]), 1)))
PY
