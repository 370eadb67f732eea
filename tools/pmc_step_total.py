"""Counter total PER STEP over all kernels of a multi-launch step, from rocprofv3 --pmc CSV output:
usage: python tools/pmc_step_total.py <dir-or-counter_collection.csv> <anchor-kernel-substring> [kernel-substring]
One step = the dispatches between two consecutive dispatches of the anchor kernel (e.g. `adam_kernel`: once per optimizer
step; `mix_decoder` for the inference call); the mean over the steps of the run (the first two are skipped: warm-up,
allocation) is printed per counter, with the per-kernel breakdown of that mean."""
import collections
import csv
import glob
import os
import sys

path, anchor = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else ""
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])))
rows.sort()
by_counter = collections.defaultdict(list)
for d, k, c, v in rows:
    by_counter[c].append((d, k, v))
for c, lst in sorted(by_counter.items()):
    steps, cur = [], collections.defaultdict(float)
    for d, k, v in lst:
        if want in k:
            cur[k[:60]] += v
        if anchor in k:
            steps.append(cur)
            cur = collections.defaultdict(float)
    steps = steps[2:] if len(steps) > 4 else steps
    if not steps:
        print("%s: anchor kernel %r not found" % (c, anchor))
        continue
    tot = sum(sum(s.values()) for s in steps) / len(steps)
    print("%-28s per step (mean of %d steps): %.6g" % (c, len(steps), tot))
    keys = sorted({k for s in steps for k in s}, key=lambda k: -sum(s.get(k, 0.0) for s in steps))
    for k in keys[:12]:
        print("    %-62s %.6g" % (k, sum(s.get(k, 0.0) for s in steps) / len(steps)))
