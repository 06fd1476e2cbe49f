"""Device timeline of the last timed repeat of a profiled bench run (rocprofv3 --kernel-trace CSV):
    python tools/timeline.py <dir>/<prefix>_kernel_trace.csv"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "nuts3_kernel" in r["Kernel_Name"] or "nuts_kernel" in r["Kernel_Name"]]
last = idx[-1]
a = last
while a > 0 and "prep" not in rows[a]["Kernel_Name"]:
    a -= 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:last + 16]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:46]:46s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}")
    prev_end = e
