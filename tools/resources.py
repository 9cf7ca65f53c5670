"""Table of the compiler's resource remarks (csrc/*_resources.txt): VGPRs, scratch, occupancy, spills per kernel."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "raytracer-rust_amd", "csrc")
for f in ("opaque", "kernel", "wavefront"):
    cur = None
    for line in open(os.path.join(d, f + "_resources.txt")):
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = subprocess.run(["c++filt", t.split(": ")[1]], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name)}
        elif cur is not None:
            k, v = t.split(": ")
            cur[k] = v
            if k.startswith("LDS"):
                print(f"{cur['name']:60s} VGPR {cur['VGPRs']:>4s} scratch {cur['ScratchSize [bytes/lane]']:>4s} occ {cur['Occupancy [waves/SIMD]']} "
                      f"sgpr-spill {cur['SGPRs Spill']:>3s} vgpr-spill {cur['VGPRs Spill']:>3s} LDS {v}")
                cur = None
