"""Soak test of the fit's helper workgroups (RELPOSE_TUNE_FIT_CLUSTER): B random pairs of N keypoints, the fit with G workgroups per
pair `reps` times; every run must equal the single-workgroup fit bitwise (poses, traces, product counts).
    python tools/fit_cluster_soak.py N B G reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from relativepose_amd import rpmodule, _lib, params
N, B, G, reps = (int(a) for a in sys.argv[1:5])
dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
def mk(n):
    pc = rng.standard_normal((B, n, 3)); nn = rng.standard_normal((B, n, 3)); nn /= np.linalg.norm(nn, axis=-1, keepdims=True)
    ft = rng.standard_normal((B, n, 32)).astype(np.float32) * 0.3
    return pc, nn, ft
ps, ns_, fs = mk(N)
R = np.eye(3); t = np.array([0.1, -0.2, 0.05])
pt = ps @ R.T + t + 0.05 * rng.standard_normal(ps.shape); nt_ = ns_ + 0.05 * rng.standard_normal(ps.shape); nt_ /= np.linalg.norm(nt_, axis=-1, keepdims=True)
ft = fs + 0.1 * rng.standard_normal(fs.shape).astype(np.float32)
T = lambda a, dt=torch.float64: torch.tensor(a, dtype=dt, device=dev)
w = torch.ones(B, N, dtype=torch.float64, device=dev)
n = torch.full((B,), N, dtype=torch.int32, device=dev)
args = (T(ps), T(ns_), T(fs, torch.float32), w, T(pt), T(nt_), T(ft, torch.float32), w, n, n)
para = rpmodule.opts(*params.final_params("suncg")[0])
print("setup done", flush=True)
with _lib.tuning(fit_cluster=1):
    ref = rpmodule.match_pairs(*args, para, debug=True)
    torch.cuda.synchronize()
print("G=1 status", ref.status.cpu().tolist()[:4], "products", ref.eig_iters.cpu().numpy()[0].tolist(), flush=True)
bad = 0; worst = 0.0; itdiff = 0
with _lib.tuning(fit_cluster=G):
    for r in range(reps):
        res = rpmodule.match_pairs(*args, para, debug=True)
        torch.cuda.synchronize()
        d = float((res.pose - ref.pose).abs().max())
        if d != 0.0 or not torch.equal(res.trace, ref.trace): bad += 1; worst = max(worst, d)
        if not torch.equal(res.eig_iters, ref.eig_iters): itdiff += 1
print(f"N={N} B={B} G={G}: {reps} runs, {bad} differ from G=1 (worst {worst:.3g}), {itdiff} with different product counts; status", res.status.cpu().tolist()[:4], flush=True)
