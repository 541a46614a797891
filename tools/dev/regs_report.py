"""tools/dev/regs_report.py RESOURCE_REMARKS [substring ...]: registers, spills, scratch and occupancy per kernel from the remarks
hipcc prints with -Rpass-analysis=kernel-resource-usage (kernels whose demangled name holds any of the substrings; all if none)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
want = sys.argv[2:]
seen = set()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split()[0].strip()
    if name in seen:
        continue
    seen.add(name)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if want and not any(w in dn for w in want):
        continue
    f = lambda pat: (re.search(pat, b) or [None, "?"])[1]
    print("%-90s VGPR %s spill %s/%s scratch %s SGPR %s occ %s" % (dn[:90], f(r" VGPRs: (\d+)"), f(r"VGPRs Spill: (\d+)"), f(r"SGPRs Spill: (\d+)"),
          f(r"ScratchSize \[bytes/lane\]: (\d+)"), f(r"SGPRs: (\d+)"), f(r"Occupancy \[waves/SIMD\]: (\d+)")))
