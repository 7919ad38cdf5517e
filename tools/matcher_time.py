"""Time relpose_match_pairs alone (HIP events, no other work on the GPU) on the bench workload's own level-0 primitives, with the fit's
vectors in LDS (default) and in global scratch (the layout of pairs with more than 4500 correspondences).
    python tools/matcher_time.py [config 1|2|3] [pairs] [variant indices, e.g. 1 or 0,3]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
from relativepose_amd import params, synth, weights, rpmodule
from relativepose_amd.model import SCNet
from relativepose_amd.pipeline import RelativePosePipeline
from bench import CONFIGS
cfg = CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N, S, ds, mm = cfg["N"], cfg["S"], cfg["dataset"], cfg["mask"]
dev = torch.device("cuda", 0)
data = synth.make_pairs(B, 2000, ds); pts, ptw = synth.make_keypoints(B, N, 2000, mm)
net = SCNet(SimpleNamespace(batchnorm=1, useTanh=1, skipLayer=1, outputType="rgbdnsf", snumclass=S))
net.load_state_dict(weights.make_state_dict(7, S))
Cc = N * 5
sig = params.final_params(ds)
pipe = RelativePosePipeline(net, ds, mm, sig, max_edges=min(Cc * (Cc - 1), (1 << 20) * (N // 200) ** 2), alter_steps=1)
st = pipe.prepare(data["rgb"], data["norm"], data["depth"], pts, ptw, dev)
keep = []
pipe.run(st, keep=keep)
k = keep[0]
args = (k["pc"][:, 0].contiguous(), k["nn"][:, 0].contiguous(), k["ft"][:, 0].contiguous(), st["w_s"],
        k["pc"][:, 1].contiguous(), k["nn"][:, 1].contiguous(), k["ft"][:, 1].contiguous(), st["w_t"], st["ns"], st["nt"])
para = rpmodule.opts(*sig[0])
from relativepose_amd import _lib
ONLY = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else None
for vi, (name, tune) in enumerate((("fit, default", {}), ("fit, 1 workgroup per pair", {"fit_cluster": 1}),
                   ("fit, 1 workgroup per pair, convergence test every 8 products (the round-2/3 rule)", {"fit_cluster": 1, "fit_fixed_checks": 1}), ("fit, leader + 3 helper workgroups per pair", {"fit_cluster": 4}),
                   ("fit, leader + 7 helper workgroups per pair", {"fit_cluster": 8}),
                   ("fit, vectors in global scratch (the > 4500-correspondence layout)", {"fit_global_vectors": 1}))):
    if ONLY is not None and vi not in ONLY:
        continue
    with _lib.tuning(**tune):
        res = rpmodule.match_pairs(*args, para, debug=True, max_edges=pipe.max_edges)
        for _ in range(2):
            rpmodule.match_pairs(*args, para, max_edges=pipe.max_edges)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            rpmodule.match_pairs(*args, para, max_edges=pipe.max_edges)
        e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    c = res.counts.cpu().numpy(); it = res.eig_iters.cpu().numpy()
    if os.environ.get("RELPOSE_POSE_DUMP"):          # (A/B of experiment builds: poses of this variant for a later comparison)
        pth = os.environ["RELPOSE_POSE_DUMP"]
        if os.path.exists(pth):
            ref = np.load(pth)
            print(f"   max |pose - {pth}| = {np.abs(res.pose.cpu().numpy() - ref).max():.3e}")
        else:
            np.save(pth, res.pose.cpu().numpy())
    print(f"{name}: {ms:.3f} ms per relpose_match_pairs (B={B}, N={N}, {ds}); status {np.bincount(res.status.cpu().numpy(), minlength=7).tolist()}; "
          f"edges/pair mean {2 * c[:, 1].mean():.0f} max {2 * c[:, 1].max()}; matrix-vector products per round mean {it.mean(0).round(1).tolist()} max {it.max(0).tolist()}")
