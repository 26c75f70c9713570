"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (csrc/resource_usage.txt)."""
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "resource_usage.txt"
txt = open(path).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
keys = [("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("SGPR", r"SGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("LDS", r"LDS Size \[bytes/block\]")]
for b in blocks:
    name = b.split("\n")[0].split(" ")[0]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:64]
    vals = []
    for label, k in keys:
        m = re.search(k + r": (\d+)", b)
        vals.append(f"{label} {m.group(1) if m else '?':>5s}")
    print(f"{dn:66s} " + "  ".join(vals))
