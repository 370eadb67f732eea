"""Diagnostic: where does a kernel wait for global loads?  Prints, per kernel whose mangled name contains <substr>, the
sequence of `s_waitcnt vmcnt(N)` with the number of VMEM loads issued so far, loop headers and barriers - a run of
vmcnt(0) waits that each follow ONE new load is a chain of serialised memory round trips (typically loads that sit in
a branch: the compiler waits for them at the merge).
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s kernel.hip; python tools/isa_waits.py k.s <substr>"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2]
starts = [i for i, l in enumerate(lines) if l.startswith("_Z") and l.rstrip().endswith(":") is False and ":" in l and sub in l.split(":")[0]]
for i in starts:
    name = lines[i].split(":")[0]
    j, nl, ns, seq = i, 0, 0, []
    while j < len(lines) and "s_endpgm" not in lines[j]:
        t = lines[j]
        if re.search(r"\b(buffer|global|flat)_load", t):
            nl += 1
        if re.search(r"\b(buffer|global|flat)_store", t):
            ns += 1
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if m:
            seq.append("%d:%s" % (nl, m.group(1)))
        if "s_barrier" in t:
            seq.append("BAR")
        if "v_mfma" in t and (not seq or seq[-1] != "M"):
            seq.append("M")
        if re.match(r"\.LBB\d+_\d+:.*Loop Header", t) or ("Loop Header" in t):
            seq.append("<loop>")
        j += 1
    print(name[:110])
    print("  loads %d stores %d" % (nl, ns))
    print("  " + " ".join(seq))
