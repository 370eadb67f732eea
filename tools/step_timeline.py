"""Print the kernel timeline of the last optimizer step in a rocprofv3 --kernel-trace CSV
(usage: python tools/step_timeline.py <..._kernel_trace.csv> [marker-substring, default 'adam'] [substring a kernel of the step
must contain: the last step that has one is printed])."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "adam"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
if len(sys.argv) > 3:
    for j in range(len(idx) - 1, 0, -1):
        if any(sys.argv[3] in r["Kernel_Name"] for r in rows[idx[j - 1] + 1:idx[j]]):
            a, b = idx[j - 1], idx[j]
            break
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %8.1f  %-62s grid=%s,%s,%s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:62], r["Grid_Size_X"],
                                                 r["Grid_Size_Y"], r["Grid_Size_Z"]))
