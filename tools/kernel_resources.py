"""Per-kernel register / LDS / occupancy table of a HIP source, from hipcc -Rpass-analysis=kernel-resource-usage.
    python tools/kernel_resources.py relativepose_amd/csrc/scnet.hip [substring filter ...]"""
import re
import subprocess
import sys

src, filt = sys.argv[1], sys.argv[2:]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
KEYS = (("vgpr", "VGPRs"), ("agpr", "AGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"),
        ("lds", r"LDS Size \[bytes/block\]"), ("sgpr-spill", "SGPRs Spill"), ("vgpr-spill", "VGPRs Spill"))
for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
    name = b.split("\n")[0].strip().split(" ")[0]
    if filt and not any(f in name for f in filt):
        continue
    vals = []
    for label, key in KEYS:
        m = re.search(key + r": (\d+)", b)
        vals.append(f"{label} {int(m.group(1)) if m else -1}")
    print(f"{name[:90]:90s} " + " ".join(vals))
if r.returncode:
    print(r.stderr[-3000:])
