"""Run a few SCNet forwards at the bench batch (32 pairs = 64 images) -- target for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from relativepose_amd import weights
from relativepose_amd.model import SCNet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
net = SCNet(SimpleNamespace(batchnorm=1, useTanh=1, skipLayer=1, outputType="rgbdnsf", snumclass=15))
net.load_state_dict(weights.make_state_dict(7, 15))
if len(sys.argv) > 3:
    net.set_precision(sys.argv[3])
torch.manual_seed(0)
x = torch.randn(n, 16, 160, 640, device='cuda')
for _ in range(reps):
    y = net(x)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
