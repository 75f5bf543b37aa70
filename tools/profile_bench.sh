#!/bin/bash
# Regenerates the committed profile summaries for the command the bench line comes from (run on the GPU box):
#   tools/profile_bench.sh r01e     ->  gpurun_out/r01e_{kernel_stats_bench_full.csv, pmc_hbm_bytes.json, bench_line_under_rocprof.json}
# Three separate rocprofv3 runs (kernel trace; FETCH_SIZE; WRITE_SIZE): PMC passes never share a run with tracing options
# other than --kernel-trace.  Copy the three files into profiles/ afterwards.
set -u
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
args="bench.py"
pmc_args="bench.py --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -o kt -- python3 $args > $out/${tag}_bench_line_under_rocprof.json 2> $out/${tag}_kt.err || exit 1
find $out/${tag}_kt -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats_bench_full.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_$c -o pmc -- python3 $pmc_args > /dev/null 2> $out/${tag}_$c.err || exit 1
done
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, re, statistics, sys, collections
out, tag = sys.argv[1], sys.argv[2]
rows = []
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{tag}_{c}/**/*counter_collection.csv", recursive=True)
    per = collections.defaultdict(lambda: collections.defaultdict(float))       # kernel -> dispatch -> sum over instances
    for r in csv.DictReader(open(f[0])):
        m = re.search(r"(k_\w+(?:<[\w, ]+>)?)", r["Kernel_Name"])
        if not m or r["Counter_Name"] != c:
            continue
        per[m.group(1)][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, d in per.items():
        v = list(d.values())
        rows.append({"counter": c, "kernel": k, "dispatches": len(v), "median_KB": statistics.median(v), "max_KB": max(v)})
json.dump(rows, open(f"{out}/{tag}_pmc_hbm_bytes.json", "w"), indent=1)
print("wrote", f"{out}/{tag}_pmc_hbm_bytes.json", len(rows), "rows")
PY
