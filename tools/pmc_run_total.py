"""Total of every counter over ALL dispatches of a rocprofv3 --pmc run, divided by a step count:
usage: python tools/pmc_run_total.py <dir> <steps> [kernel-substring]  -> per-step totals and the per-kernel breakdown."""
import collections
import csv
import glob
import os
import sys

path, steps = sys.argv[1], float(sys.argv[2])
want = sys.argv[3] if len(sys.argv) > 3 else ""
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            tot[r["Counter_Name"]][r["Kernel_Name"][:70]] += float(r["Counter_Value"])
for c, ks in sorted(tot.items()):
    print("%-28s per step (run total / %g steps): %.6g" % (c, steps, sum(ks.values()) / steps))
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1])[:10]:
        print("    %-72s %.6g" % (k, v / steps))
