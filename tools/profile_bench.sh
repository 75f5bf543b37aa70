#!/bin/bash
# Regenerates the committed profile summaries for the command the bench line comes from (run on the GPU box):
#   tools/profile_bench.sh r01e     ->  gpurun_out/r01e_{kernel_stats_bench_full.csv, kernel_trace_headline_kernels.txt, pmc_hbm_bytes.json, bench_line_under_rocprof.json}
# Three separate rocprofv3 runs (kernel trace; FETCH_SIZE; WRITE_SIZE): PMC passes never share a run with tracing options
# other than --kernel-trace.  Copy the three files into profiles/ afterwards.
set -u
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
args="bench.py"
pmc_args="bench.py --no-cpu-baseline"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -o kt -- python3 $args > $out/${tag}_bench_line_under_rocprof.json 2> $out/${tag}_kt.err || exit 1
find $out/${tag}_kt -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats_bench_full.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_$c -o pmc -- python3 $pmc_args > /dev/null 2> $out/${tag}_$c.err || exit 1
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

# per-dispatch summary of the headline kernels from the same kernel trace: the --stats average of the sweep is diluted by the ~4 us
# no-op launches after convergence; "full" = the dispatches above 100 us
python3 - "$out" "$tag" <<'PY'
import csv, glob, re, statistics, sys, collections
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(f"{out}/{tag}_kt/**/*kernel_trace.csv", recursive=True)
if f:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        m = re.search(r"(k_\w+(?:<[\w, ]+>)?)", r["Kernel_Name"])
        if m:
            per[m.group(1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines = ["# per-dispatch durations (us) from the kernel trace behind %s_kernel_stats_bench_full.csv (same rocprofv3 run of `python3 bench.py`)" % tag,
             "# 'full' (k_pr_sweep*) = dispatches above 100 us: sweeps that did a sweep's work on the 10M / 50M graph; config 2's sweeps (~55 us) and the",
             "# no-op launches after convergence (~4 us) fall under it",
             "kernel, dispatches, avg_all, dispatches_full, avg_full, median_full, min_full, max_full"]
    for k in sorted(per):
        if not re.match(r"k_(pr_sweep|score_wave|merge_flat|wave_prep|score_slices|merge_topk|scatter|bucket_sum|weight_count|aff_)", k):
            continue
        v = per[k]
        full = [x for x in v if x > 100] if "pr_sweep" in k else v
        if not full:
            full = v
        lines.append(f"{k}, {len(v)}, {sum(v) / len(v):.1f}, {len(full)}, {sum(full) / len(full):.1f}, {statistics.median(full):.1f}, {min(full):.1f}, {max(full):.1f}")
    open(f"{out}/{tag}_kernel_trace_headline_kernels.txt", "w").write("\n".join(lines) + "\n")
    print("wrote", f"{out}/{tag}_kernel_trace_headline_kernels.txt")
PY
