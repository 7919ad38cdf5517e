"""Per-kernel register / LDS / occupancy table of a HIP source, from hipcc -Rpass-analysis=kernel-resource-usage.
    python tools/kernel_resources.py relativepose_amd/csrc/scnet.hip [substring filter ...]"""
import re
import subprocess
import sys

src, filt = sys.argv[1], sys.argv[2:]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
t = r.stderr
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].strip()
    if filt and not any(f in name for f in filt):
        continue

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    print(f"{name[:100]:100s} vgpr {g('VGPRs'):4d} agpr {g('AGPRs'):4d} scratch {g(r'ScratchSize \[bytes/lane\]'):5d} "
          f"occ {g(r'Occupancy \[waves/SIMD\]')} lds {g(r'LDS Size \[bytes/block\]'):6d} sgpr-spill {g('SGPRs Spill')} vgpr-spill {g('VGPRs Spill')}")
