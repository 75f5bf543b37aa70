"""Every kernel of the last stretch of a rocprofv3 kernel trace, in start order: start / duration / gap to the previous kernel's end.
    python tools/trace_all.py gpurun_out/kt_<tag> <first kernel name fragment> [occurrence from the end, default 1]"""
import csv, glob, sys
d, frag = sys.argv[1], sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 1
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if frag in r["Kernel_Name"]]
i0 = starts[-occ]
t0 = int(rows[i0]["Start_Timestamp"]); prev = t0
busy = 0
for r in rows[i0:(starts[-occ + 1] if occ > 1 else len(rows))]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:7.1f}  {r['Kernel_Name'][:90]}")
    prev = max(prev, e); busy += e - s
print(f"span {(prev - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
