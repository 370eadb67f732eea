#!/usr/bin/env python3
"""Rewrite profiles/traffic.json - the one source of bench.py's roofline.traffic - from the PMC summaries of a measurement round:
    python3 tools/traffic_refresh.py <round dir, e.g. gpurun_out/r04_final2> <prefix for the copies under profiles/, e.g. r04>
Reads <dir>/bench_pmc_summary.txt (headline kernel, KiB per launch) and <dir>/pmcstep_*.txt (KiB per step), copies them to
profiles/<prefix>_pmc_bench.txt / profiles/<prefix>_pmcstep_*.txt and points traffic.json at the copies."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, prefix = sys.argv[1], sys.argv[2]
tj = os.path.join(ROOT, "profiles", "traffic.json")
d = json.load(open(tj))


def per_step(path):
    out = {}
    for line in open(path):
        m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+per step \(.*?\):\s*([0-9.e+]+)", line)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


head = os.path.join(src, "bench_pmc_summary.txt")
if os.path.exists(head):
    vals = {}
    for line in open(head):
        m = re.match(r"void fov::lstm_cluster_fused_kernel<256, 0>.*?(FETCH_SIZE|WRITE_SIZE)\s+n=\d+\s+mean=([0-9.e+]+)", line)
        if m:
            vals[m.group(1)] = float(m.group(2))
    if len(vals) == 2:
        dst = "profiles/%s_pmc_bench.txt" % prefix
        shutil.copy(head, os.path.join(ROOT, dst))
        d["headline"].update({"fetch_kib": vals["FETCH_SIZE"], "write_kib": vals["WRITE_SIZE"], "source": dst})
for name, key in (("train_mixing_f32", "train_mixing/f32"), ("train_mixing_bf16", "train_mixing/bf16"), ("infer_mixing_f32", "infer_mixing/f32"),
                  ("infer_mixing_bf16", "infer_mixing/bf16"), ("train_f32", "train/f32"), ("config1", "config1/f32"), ("a10", "a10/f32"),
                  ("convlstm", "convlstm/f32")):
    f = os.path.join(src, "pmcstep_%s.txt" % name)
    if not os.path.exists(f):
        continue
    v = per_step(f)
    if len(v) != 2:
        continue
    dst = "profiles/%s_pmcstep_%s.txt" % (prefix, name)
    shutil.copy(f, os.path.join(ROOT, dst))
    d["modes"][key] = {"fetch_kib": v["FETCH_SIZE"], "write_kib": v["WRITE_SIZE"], "source": dst}
json.dump(d, open(tj, "w"), indent=1)
print(json.dumps(d, indent=1))
