"""Prints a one-line summary per kernel from hipcc's -Rpass-analysis=kernel-resource-usage log."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "raytracer-rust_amd/csrc/kernel_resources.txt"
txt = open(path).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
KEYS = [("SGPR", r"TotalSGPRs"), ("VGPR", r" VGPRs"), ("AGPR", r"AGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("LDS", r"LDS Size \[bytes/block\]")]
for b in blocks:
    name = b.split()[0].replace("_ZN3rtk", "")[:58]
    vals = []
    for label, key in KEYS:
        m = re.search(key + r": (\d+)", b)
        vals.append(f"{label} {m.group(1) if m else '?':>5s}")
    print(f"{name:60s} " + "  ".join(vals))
