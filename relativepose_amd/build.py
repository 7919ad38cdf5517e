"""Build librelpose_hip.so (gfx950) in-tree with hipcc.  No torch involved: the
library is a plain C-ABI shared object (include/relpose.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librelpose_hip.so")
ARCH = "gfx950"

# (source, extra flags).  matcher/geometry must round like numpy: no FMA contraction.
SOURCES = [
    ("matcher.hip", ["-ffp-contract=off"]),
    ("affinity.hip", ["-ffp-contract=off"]),
    ("geometry.hip", ["-ffp-contract=off"]),
    ("keypoints.hip", ["-ffp-contract=off"]),
    ("scnet.hip", []),
]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
              [os.path.join(os.path.dirname(HERE), "include", "relpose.h")]
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _newer(s, o) or any(_newer(h, o) for h in headers):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o] + extra + \
                  os.environ.get("RELPOSE_HIPCC_FLAGS", "").split()       # experiments only (e.g. -DRP_ABLATE=3)
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or any(_newer(o, LIB) for o in objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
