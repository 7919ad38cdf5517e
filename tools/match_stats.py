"""Graph sizes and power-iteration counts of the matcher on the bench workload (per recurrent level)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
from relativepose_amd import synth, weights, rpmodule
from relativepose_amd.model import SCNet
from relativepose_amd.pipeline import RelativePosePipeline
from relativepose_amd.params import FINAL_PARAMS
SUNCG_SIGMAS = FINAL_PARAMS["suncg"]
B, N, S = 16, 200, 15
dev = torch.device("cuda", 0)
data = synth.make_pairs(B, 2000, "suncg"); pts, ptw = synth.make_keypoints(B, N, 2000, "second")
net = SCNet(SimpleNamespace(batchnorm=1, useTanh=1, skipLayer=1, outputType="rgbdnsf", snumclass=S))
net.load_state_dict(weights.make_state_dict(7, S))
Cc = N * 5
pipe = RelativePosePipeline(net, "suncg", "second", SUNCG_SIGMAS, max_edges=min(Cc * (Cc - 1), 1 << 20))
st = pipe.prepare(data["rgb"], data["norm"], data["depth"], pts, ptw, dev)
keep = []
pose, status, tr = pipe.run(st, keep=keep)
for lvl, k in enumerate(keep):
    pc, nn, ft = k["pc"], k["nn"], k["ft"]
    para = rpmodule.opts(*pipe.sigmas[min(lvl, len(pipe.sigmas) - 1)])
    res = rpmodule.match_pairs(pc[:, 0].contiguous(), nn[:, 0].contiguous(), ft[:, 0].contiguous(), st["w_s"],
                               pc[:, 1].contiguous(), nn[:, 1].contiguous(), ft[:, 1].contiguous(), st["w_t"],
                               st["ns"], st["nt"], para, debug=True, max_edges=pipe.max_edges)
    c = res.counts.cpu().numpy(); it = res.eig_iters.cpu().numpy()
    print("level", lvl, "method", para.method, "counts[min,mean,max] dist-pass", c[:, 0].min(), c[:, 0].mean(), c[:, 0].max(),
          "| surviving pairs M", c[:, 1].min(), c[:, 1].mean(), c[:, 1].max(), "| nonzero w", c[:, 2].mean(), "| K_eff", c[:, 3].mean())
    print("   eig iters per round (mean over pairs)", it.mean(0), "max", it.max(0))
