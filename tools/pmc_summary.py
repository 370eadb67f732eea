"""Mean counter value per dispatch and kernel from rocprofv3 --pmc CSV output
(usage: python tools/pmc_summary.py <dir-or-counter_collection.csv> [kernel-substring])."""
import collections
import csv
import glob
import os
import sys

path = sys.argv[1]
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0.0, set()])
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if want not in k:
            continue
        key = (k[:70], r.get("Grid_Size", ""), r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"])
        acc[key][1].add(r["Dispatch_Id"])
for (k, gsz, c), (v, d) in sorted(acc.items()):
    print("%-70s grid=%-9s %-28s n=%-4d mean=%.4g" % (k, gsz, c, len(d), v / max(len(d), 1)))
