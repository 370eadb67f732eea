"""Mean duration per kernel name from a rocprofv3 --kernel-trace CSV (usage: python tools/kernel_means.py <kernel_trace.csv> [substr])."""
import collections
import csv
import sys

acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
        continue
    acc[r["Kernel_Name"][:80]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)[len(v) // 4:]      # drop the warm-up quartile
    print("%-80s n=%-4d mean %.1f us  min %.1f" % (k, len(v), sum(v2) / len(v2), min(v)))
