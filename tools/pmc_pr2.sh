#!/bin/bash
# usage: tools/pmc_pr2.sh "<counters>" tag   -- rocprofv3 --pmc over tools/pr_exp.py (R=3), per-dispatch medians for k_pr_sweep<16, false> and k_pr_probe<16, .>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmcpr_$2
rm -rf $out
R=3 timeout -k 10 250 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $out -o pmc -- python3 tools/pr_exp.py > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
per = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"]
    if "k_pr_sweep<16" in k: k = "step"
    elif "k_pr_probe<16" in k: k = "probe"
    else: continue
    per[(k, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
for (k, c), d in sorted(per.items()):
    v = sorted(d.values())
    print(k, c, "median", v[len(v) // 2], "n", len(v))
PY
