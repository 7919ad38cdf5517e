"""The three forwards of one pipeline step ALONE on the GPU -- level 0 (zero-warp plan, fills the self-stream cache), then two self-cached forwards
-- with nothing on any other stream: the contention-free time of the SCNet part of a step.  bench.py's in-loop figure divides the same executed work
by the step time of the running pipeline (matcher / geometry / heads / resize of the other batch alongside); the difference is what the overlap costs.
  python tools/loop_forwards.py [n_images] [steps]        (also a target for rocprofv3 --kernel-trace: per-kernel times of the cached plans)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace

import torch

from relativepose_amd import weights
from relativepose_amd.model import SCNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
S = 15
net = SCNet(SimpleNamespace(batchnorm=1, useTanh=1, skipLayer=1, outputType="rgbdnsf", snumclass=S))
net.load_state_dict(weights.make_state_dict(7, S))
if len(sys.argv) > 3:
    net.set_precision(sys.argv[3])
torch.manual_seed(0)
x0 = torch.randn(n, 16, 160, 640, device="cuda")
x0[:, 8:] = 0
x1 = x0.clone()
x1[:, 8:] = torch.randn(n, 8, 160, 640, device="cuda")
out = torch.empty(n, net.out_channels, 160, 640, device="cuda")


def step():
    tag = net.new_self_tag()
    net.forward(x0, out=out, zero_warp=True, self_tag=tag)
    net.forward(x1, out=out, self_tag=tag)
    net.forward(x1, out=out, self_tag=tag)


for _ in range(2):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    step()
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / steps
full = net.plan_macs(n)
ex = net.plan_macs(n, SCNet.FLAG_ZERO_WARP) + 2 * net.plan_macs(n, 0, self_cached=True)
gflop = 36.14 * n * ex / full
print(f"forwards of one step alone: {ms:.3f} ms per step ({n} images), executed {gflop:.1f} GFLOP -> {gflop / ms:.1f} TFLOP/s = {gflop / ms / 157.3:.3f} of the fp32 MFMA peak "
      f"(= {n // 2 * 1e3 / ms:.1f} pairs/s if nothing else ran)")
