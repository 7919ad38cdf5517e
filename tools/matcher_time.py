"""Time relpose_match_pairs alone (HIP events, no other work on the GPU) on the bench workload's own level-0 primitives, for the
default single-workgroup Lanczos fit and the round-1 launch-sequence fit (RELPOSE_LEGACY_FIT=1), and report the result difference.
    python tools/matcher_time.py [config 1|2|3] [pairs]"""
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
out = {}
for name, env in (("single-workgroup Lanczos fit (default)", {}), ("default fit, row-per-wave pair kernels", {"RELPOSE_LEGACY_PAIRS": "1"}), ("round-1 launch-sequence fit", {"RELPOSE_LEGACY_FIT": "1", "RELPOSE_LEGACY_AFFINITY": "1"})):
    for kk in ("RELPOSE_LEGACY_FIT", "RELPOSE_LEGACY_AFFINITY", "RELPOSE_LEGACY_PAIRS"):
        os.environ.pop(kk, None)
    os.environ.update(env)
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
    out[name] = res.pose.cpu().numpy()
    print(f"{name}: {ms:.3f} ms per relpose_match_pairs (B={B}, N={N}, {ds}); status {np.bincount(res.status.cpu().numpy(), minlength=7).tolist()}; "
          f"edges/pair mean {2 * c[:, 1].mean():.0f} max {2 * c[:, 1].max()}; matrix-vector products per round mean {it.mean(0).round(1).tolist()} max {it.max(0).tolist()}")
a, b = list(out.values())[0], list(out.values())[-1]
print('tiled vs row-per-wave pair kernels: poses bit-equal =', np.array_equal(list(out.values())[0], list(out.values())[1]))
print("rotation difference between the two fits (Frobenius): max %.3e median %.3e" % (np.linalg.norm((a - b)[:, :3, :3], axis=(1, 2)).max(),
      np.median(np.linalg.norm((a - b)[:, :3, :3], axis=(1, 2)))))
# which fit is right where they differ most?  the CPU oracle (scipy ARPACK like the reference) on that pair
from oracle import rp_oracle as M
worst = int(np.argmax(np.linalg.norm((a - b)[:, :3, :3], axis=(1, 2))))
cpu = lambda t: t.cpu().numpy()
S_ = {"pc": cpu(args[0][worst]), "normal": cpu(args[1][worst]), "feat": cpu(args[2][worst]), "weight": cpu(args[3][worst])}
T_ = {"pc": cpu(args[4][worst]), "normal": cpu(args[5][worst]), "feat": cpu(args[6][worst]), "weight": cpu(args[7][worst])}
ref = M.relative_pose_helper(S_, T_, M.Params(*sig[0]))
print("pair %d (largest difference): rotation error vs the oracle (ARPACK): Lanczos fit %.3e, launch-sequence (64-step power iteration) fit %.3e"
      % (worst, np.linalg.norm(a[worst, :3, :3] - ref[:3, :3]), np.linalg.norm(b[worst, :3, :3] - ref[:3, :3])))
