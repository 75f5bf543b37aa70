#!/bin/bash
# usage: tools/pmc2.sh "<counters>" tag <script> <kernel regex> -- rocprofv3 --pmc over a tools/ script, per-kernel counter averages
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$2
rm -rf $out
timeout -k 10 400 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $out -o pmc -- python3 $3 > $out.log 2>&1
python3 - "$out" "$4" <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
pat = re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"]
    m = pat.search(k)
    if not m: continue
    k = m.group(0)
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[(k, row["Counter_Name"])] += 1
for k in acc:
    for c, v in acc[k].items():
        print(k, c, f"{v / max(n[(k, c)], 1):.4g}", f"(n={n[(k,c)]})")
PY
