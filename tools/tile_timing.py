"""Experiment (round 6): where the waves of deconv_tile_kernel spend their cycles.  Needs the timing build:
    python tools/build_variant.py tt -DRP_TILE_TIMING=1 && RELPOSE_LIB_PATH=relativepose_amd/librelpose_hip_tt.so python tools/tile_timing.py [precision]
One full forward at 64 images; s_memtime stamps around the segments of the main loop of the non-paired deconv_tile_kernel instantiations (deconv2),
summed over all waves (csrc/scnet.hip, RP_TILE_TIMING)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace

import torch

from relativepose_amd import _lib, weights
from relativepose_amd.model import SCNet

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
net = SCNet(SimpleNamespace(batchnorm=1, useTanh=1, skipLayer=1, outputType="rgbdnsf", snumclass=15))
net.load_state_dict(weights.make_state_dict(7, 15))
net.set_precision(prec)
torch.manual_seed(0)
x = torch.randn(64, 16, 160, 640, device="cuda")
lib = C.CDLL(_lib.LIB_PATH)
fn = lib.relpose_debug_tile_timing
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
net(x); torch.cuda.synchronize()
fn(None, 1)
net(x); torch.cuda.synchronize()
out = (C.c_uint64 * 8)()
fn(out, 0)
v = [int(a) for a in out]
names = ["load issue", "MFMAs + fragment reads", "barrier 1", "LDS stores (+ A transform)", "barrier 2", "prologue", "whole kernel", "waves"]
w = max(v[7], 1)
print(f"precision {prec}: {w} waves of deconv_tile_kernel (non-paired instantiations: deconv2), mean cycles per wave")
for n_, a in zip(names[:7], v[:7]):
    print(f"  {n_:28s} {a / w:10.0f}  ({100.0 * a / max(v[6], 1):5.1f} % of the kernel)")
print(f"  epilogue + rest              {(v[6] - sum(v[:6])) / w:10.0f}  ({100.0 * (v[6] - sum(v[:6])) / max(v[6], 1):5.1f} %)")
