"""fp32 vs an opt-in split-precision mode (argv[2]: f16x3 | bf16x3) of SCNet: output difference and forward time at the bench batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from relativepose_amd import weights
from relativepose_amd.model import SCNet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = SCNet(SimpleNamespace(batchnorm=1, useTanh=1, skipLayer=1, outputType="rgbdnsf", snumclass=15))
net.load_state_dict(weights.make_state_dict(7, 15))
torch.manual_seed(0)
x = torch.randn(n, 16, 160, 640, device='cuda')
def run(mode):
    net.set_precision(mode)
    y = net(x); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        y = net(x)
    torch.cuda.synchronize()
    return y, (time.perf_counter() - t0) / 3 * 1e3
y32, t32 = run("f32")
y16, t16 = run(sys.argv[2] if len(sys.argv) > 2 else "bf16x3")
y32b, _ = run("f32")
d = (y16 - y32).abs()
print(f"forward ms: f32 {t32:.2f}  {sys.argv[2] if len(sys.argv) > 2 else chr(98)+chr(102)+'16x3'} {t16:.2f}")
print(f"{sys.argv[2] if len(sys.argv) > 2 else chr(98)+chr(102)+chr(49)+chr(54)+chr(120)+chr(51)} vs f32: max abs {d.max().item():.3e}  mean abs {d.mean().item():.3e}  (output abs mean {y32.abs().mean().item():.3f}, max {y32.abs().max().item():.2f})")
for name, sl in (("rgb", slice(0, 3)), ("normal", slice(3, 6)), ("depth", slice(6, 7)), ("sem", slice(7, 22)), ("feat", slice(22, 54))):
    print(f"   {name:7s} max abs {d[:, sl].max().item():.3e}  ref scale {y32[:, sl].abs().max().item():.3f}")
print("f32 reproducible after switching back:", torch.equal(y32, y32b))
