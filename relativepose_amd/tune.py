"""sigma tuning of the pose module: the finite-difference
objective of the reference's trainRelativePoseModuleRecFD.py:215-298, with every
objective evaluation done as ONE batched `match_pairs` call on the GPU instead of
a Python loop over cached primitives.

A cached primitive is the reference's dict (trainRelativePoseModuleRecFD.py:207-208):
  pc_src/normal_src/feat_src/weight_src, pc_tgt/..., R_gt.
"""
import numpy as np

from . import _lib, rpmodule
from .evaluation import angular_distance_np  # noqa: F401  (util.py:176-187)


class PrimitiveSet:
    """Cached primitives resident on the device (uploaded once, reused by every objective call)."""

    def __init__(self, primitives):
        dev = _lib.require_gpu()
        cases = [({'pc': p['pc_src'], 'normal': p['normal_src'], 'feat': p['feat_src'], 'weight': p['weight_src']},
                  {'pc': p['pc_tgt'], 'normal': p['normal_tgt'], 'feat': p['feat_tgt'], 'weight': p['weight_tgt']}) for p in primitives]
        self.args = rpmodule.pack_keypoints(cases, dev)
        self.R_gt = np.stack([np.asarray(p['R_gt'], dtype=np.float64) for p in primitives])

    def poses(self, para):
        return rpmodule.match_pairs(*self.args, para).pose.cpu().numpy()


def cache_primitives(pipe, batches, device=None):
    """The producer of the primitive cache (trainRelativePoseModuleRecFD.py:129-212): run the recurrent loop over `batches` (dicts in the
    DataLoader layout of evaluation.evaluate_pairs: rgb / norm [B,2,3,h,4h], depth [B,2,h,4h], R [B,2,4,4] and pts [B,2,N,2], ptw [B,2,N] -- or, for a
    keypoints="reference" pipeline, sift = [(source detections, target detections)] * B --) and keep
    the LAST level's matching primitives of every scan pair in the reference's dict format (:207-208)

        {'pc_src' [n,3], 'normal_src' [n,3], 'feat_src' [n,32] f32, 'weight_src' [n], 'pc_tgt', 'normal_tgt', 'feat_tgt', 'weight_tgt', 'R_gt' [4,4]}

    -- what `PrimitiveSet` / `objective` / `tune_step` consume (the reference np.save's the list, :212).  The reference runs the matcher on
    the levels BEFORE the last one only (:197-205: the last level's pose is what the tuning optimises); the pipeline's last-level match is
    computed and dropped.  Pairs a level leaves without keypoints (doCompletion == 0) keep their n = 0 arrays."""
    import torch
    dev = device if device is not None else _lib.require_gpu()
    out = []
    for batch in batches:
        # either keypoint mode: "given" batches carry pts / ptw, keypoints="reference" batches carry the views' SIFT detections.
        # Note for the kinect convention: the per-level sets hold up to 560 keypoints per view -- beyond the 512 targets of the affinity tile /
        # pool kernels, so that mode's affinity build runs on the (slower) LDS kernel (csrc/affinity.hip)
        from .evaluation import _prepare_batch
        st = _prepare_batch(pipe, batch, dev)
        prim = {}
        pipe.run(st, primitives=prim)
        pc, nn, ft = (prim[k].cpu().numpy() for k in ("pc", "nn", "ft"))
        ns, nt = prim["ns"].cpu().numpy(), prim["nt"].cpu().numpy()          # (the last level's own keypoint set: keypoints="reference" derives one per level)
        ws, wt = prim["w_s"].cpu().numpy(), prim["w_t"].cpu().numpy()
        for b in range(st["B"]):
            R_gt = np.matmul(batch["R"][b, 1], np.linalg.inv(batch["R"][b, 0]))          # evaluation.py:213 / trainRelativePoseModuleRecFD.py:118
            out.append({'pc_src': pc[b, 0, :ns[b]].copy(), 'normal_src': nn[b, 0, :ns[b]].copy(), 'feat_src': ft[b, 0, :ns[b]].copy(),
                        'weight_src': ws[b, :ns[b]].copy(),
                        'pc_tgt': pc[b, 1, :nt[b]].copy(), 'normal_tgt': nn[b, 1, :nt[b]].copy(), 'feat_tgt': ft[b, 1, :nt[b]].copy(),
                        'weight_tgt': wt[b, :nt[b]].copy(), 'R_gt': R_gt})
        del st, prim
    return out


def objective(prims, para):
    """trainRelativePoseModuleRecFD.py:215-233: (mean squared Frobenius rotation error, mean angular distance)."""
    if not isinstance(prims, PrimitiveSet):
        prims = PrimitiveSet(prims)
    R_hat = prims.poses(para)
    loss = np.power(R_hat[:, :3, :3] - prims.R_gt[:, :3, :3], 2).sum((1, 2))
    ad = angular_distance_np(R_hat[:, :3, :3], prims.R_gt[:, :3, :3])
    return float(loss.sum() / len(loss)), float(ad.sum() / len(ad))


def make_para(sig):
    p = rpmodule.opts()
    p.sigmaAngle1, p.sigmaAngle2, p.sigmaDist, p.sigmaFeat = [float(v) for v in sig]
    return p


def tune_step(prims, sigmas, rng, n_probe=10, objective_fn=objective, info=None):
    """One outer iteration of trainRelativePoseModuleRecFD.py:245-298: finite-difference gradient from
    `n_probe` random relative perturbations (least squares), normalised step, halving line search.
    `rng.uniform(size=4)` replaces the reference's global np.random.uniform (the same stream for the same seed).  Returns
    (new sigmas [4], loss, ad, found_descent); `info` (a dict) receives the probes: eps [n_probe,4], losses, ads, grad."""
    if not isinstance(prims, PrimitiveSet) and objective_fn is objective:
        prims = PrimitiveSet(prims)
    sig = np.asarray(sigmas, dtype=np.float64)
    eps = np.zeros((n_probe, 4))
    losses, ads = np.zeros(n_probe), np.zeros(n_probe)
    for j in range(n_probe):
        if j >= 1:
            eps[j] = (rng.uniform(size=4) - 0.5) / 5
        losses[j], ads[j] = objective_fn(prims, make_para(sig * (1 + eps[j])))
    grad = np.linalg.lstsq(eps[1:], losses[1:] - losses[0], rcond=None)[0]
    grad = grad / max(np.abs(grad / sig))
    if info is not None:
        info.update(eps=eps, losses=losses, ads=ads, grad=grad)
    alpha = 1.0
    for _ in range(10):
        cand = sig * (1 + -1 * grad * alpha)
        loss, ad = objective_fn(prims, make_para(cand))
        if loss < losses[0]:
            return cand, loss, ad, True
        alpha /= 2
    return sig, float(losses[0]), float(ads[0]), False
