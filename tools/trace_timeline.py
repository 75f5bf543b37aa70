"""Timeline of the scoring kernels from a rocprofv3 kernel trace (tools/kt.sh tag script): per batch the start / end of k_wave_prep,
k_score_wave and k_merge_flat relative to the first of the last N batches, the gaps and the overlaps.
    python tools/trace_timeline.py gpurun_out/kt_<tag> [N]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("k_score_wave", "k_merge_flat", "k_wave_prep"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
waves = [r for r in rows if "k_score_wave" in r["Kernel_Name"]][-n:]
t0 = int(waves[0]["Start_Timestamp"])
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0 - 50000]
for r in sel:
    nm = "wave " if "k_score_wave" in r["Kernel_Name"] else ("merge" if "k_merge_flat" in r["Kernel_Name"] else "prep ")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{nm} q{r.get('Queue_Id', '?'):>3s} start {s:9.1f} end {e:9.1f} dur {e - s:7.1f} us")
ws = [int(r["Start_Timestamp"]) for r in waves]
print("wave start-to-start us:", ["%.1f" % ((b - a) / 1e3) for a, b in zip(ws, ws[1:])])
