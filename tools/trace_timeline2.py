"""Timeline of the last N launches of the scoring kernels in a rocprofv3 kernel trace (tools/kt.sh), whichever they are:
    python tools/trace_timeline2.py gpurun_out/kt_<tag> [N]"""
import csv, glob, re, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if re.search(r"k_(score_slices|merge_topk|score_wave|merge_flat|wave_prep|score_small)", r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    nm = re.search(r"k_\w+", r["Kernel_Name"]).group(0)
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{nm:16s} q{r.get('Queue_Id', '?'):>3s} start {s:9.1f} end {e:9.1f} dur {e - s:7.1f} us")
