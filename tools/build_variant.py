"""Build relativepose_amd/librelpose_hip_<name>.so with extra flags on every source (experiments / ablations; e.g. -DRP_EXPERIMENTS
turns the RELPOSE_* environment switches of the experiment log back on).  Use it with RELPOSE_LIB_PATH=<that file>.
    python tools/build_variant.py xp -DRP_EXPERIMENTS"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relativepose_amd import build as b
name, flags = sys.argv[1], sys.argv[2:]
objs = []
for src, extra in b.SOURCES:
    o = f"/tmp/{src[:-4]}_{name}.o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", f"--offload-arch={b.ARCH}", "-O3", "-std=c++17", "-fPIC", *extra, *flags, "-c", os.path.join(b.CSRC, src), "-o", o],
                          stderr=subprocess.DEVNULL)
    objs.append(o)
out = os.path.join(b.HERE, f"librelpose_hip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-o", out] + objs)
print(out)
