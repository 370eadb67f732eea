"""Diagnostic: hipcc pads no hazards around inline-asm MFMAs.  Reports every v_mfma whose A/B source VGPR was written by a
VALU instruction fewer than 2 wait states earlier (VALU write -> MFMA SrcA/B read, cdna_hip_programming.md 5.7), and every
VALU / LDS-store / VMEM-store read of an MFMA result fewer than N wait states after the MFMA.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --offload-device-only -o k.s kernel.hip; python tools/isa_mfma_hazard.py k.s [substr]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2] if len(sys.argv) > 2 else ""


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    if m:
        return [int(m.group(1))]
    return []


name = None
last_valu_write = {}
last_mfma_write = {}
slot = 0
bad = 0
for ln, l in enumerate(lines):
    t = l.strip()
    if re.match(r"^_Z\w+:", t):
        name = t.split(":")[0]
        last_valu_write = {}
        last_mfma_write = {}
        slot = 0
        continue
    if name is None or (sub and sub not in name):
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        if t.endswith(":"):
            last_valu_write = {}   # label: unknown predecessors (conservative reset)
        continue
    parts = t.split(None, 1)
    op = parts[0]
    args = parts[1].split(";")[0] if len(parts) > 1 else ""
    toks = [a.strip() for a in args.split(",")]
    if op == "s_nop":
        slot += int(toks[0]) + 1
        continue
    slot += 1
    if not op.startswith("v_mfma") and not op.startswith("s_"):
        # MFMA write -> read by anything else (VALU, LDS / memory store, ...): 8-pass fp32 MFMA needs >= 11 wait states
        srcs = toks[1:] if (op.startswith("v_") or op.startswith("ds_read") or "load" in op) else toks
        for src in srcs:
            for r in regs(src):
                w = last_mfma_write.get(r)
                if w is not None and slot - w[0] < 12:
                    print("%s line %d: %s reads v%d, written %d slot(s) earlier by an MFMA" % (name[:60], ln + 1, t[:50], r, slot - w[0]))
                    bad += 1
    if op.startswith("v_mfma"):
        for r in regs(toks[0]):
            last_mfma_write[r] = (slot, t[:70])
        for src in toks[1:4]:   # A, B and the accumulator input
            for r in regs(src):
                w = last_valu_write.get(r)
                if w is not None and slot - w[0] <= 2:
                    print("%s line %d: %s reads v%d written %d slot(s) earlier by: %s" % (name[:60], ln + 1, op, r, slot - w[0], w[1]))
                    bad += 1
        continue
    if op.startswith("v_") and not op.startswith("v_mfma") and toks:
        for r in regs(toks[0]):
            last_valu_write[r] = (slot, t[:70])
print("hazards found:", bad)
