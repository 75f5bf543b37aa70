#!/bin/bash
# usage: tools/kt.sh tag <python script> [env assignments are inherited]  -- kernel-trace stats of a tools/ script (per-kernel count / avg / max us)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/kt_$1
rm -rf $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 $2 > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no stats file"); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} max_us {float(r['MaxNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
