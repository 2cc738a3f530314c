"""Print the kernel timeline (start offset, duration, name) around one layer's backward from a rocprofv3
--kernel-trace CSV directory: python tools/timeline.py <dir> [anchor substring]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
anchor = sys.argv[2] if len(sys.argv) > 2 else "dkdv"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, nm in enumerate(names) if anchor in nm]
i0 = idx[len(idx) * 3 // 4]
t0 = int(rows[i0 - 30]["Start_Timestamp"])
prev_end = t0
for r in rows[i0 - 30:i0 + 40]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} gap {(s - prev_end) / 1e3:6.1f} dur {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:100]}")
    prev_end = e
