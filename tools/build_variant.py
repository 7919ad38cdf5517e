"""Build relativepose_amd/librelpose_hip_<name>.so with extra -D flags on scnet.hip (experiments / ablations)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relativepose_amd import build as b
name, flags = sys.argv[1], sys.argv[2:]
o = f"/tmp/scnet_{name}.o"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags, "-c", os.path.join(b.CSRC, "scnet.hip"), "-o", o],
                      stderr=subprocess.DEVNULL)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(b.HERE, f"librelpose_hip_{name}.so"), o,
                       os.path.join(b.CSRC, "matcher.o"), os.path.join(b.CSRC, "geometry.o")])
