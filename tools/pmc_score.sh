#!/bin/bash
# usage: tools/pmc_score.sh "<counters>" tag   -- tools/score_exp.py (R=8) under rocprofv3 --pmc; per-kernel counter medians for the scoring kernels (MODES=wave by default: k_wave_prep / k_score_wave / k_merge_flat)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$2
rm -rf $out
MODES=${MODES:-wave} R=8 timeout -k 10 250 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $out -o pmc -- python3 tools/score_exp.py > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
per = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"]
    if "k_score_slices" in k: k = "slices"
    elif "k_merge_topk" in k: k = "merge"
    elif "k_score_wave" in k: k = "wave"
    elif "k_merge_flat" in k: k = "merge_flat"
    elif "k_wave_prep" in k: k = "wave_prep"
    else: continue
    per[(k, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
for (k, c), d in sorted(per.items()):
    v = sorted(d.values())
    print(k, c, "median", v[len(v) // 2], "n", len(v))
PY
