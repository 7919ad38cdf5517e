"""ORACLE (test infrastructure, not product code): the per-pair recurrent loop of
the reference's evaluation driver, with keypoints injected.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this.  Follows /root/reference/evaluation.py:203-284 and getMatchingPrimitive
(RPModule/rpmodule.py:511-538) with the cv2-SIFT keypoint detector
(rputil.getKeypoint, not installable here -- parity unpinned for detection)
replaced by caller-supplied pixel coordinates and weights.
"""
import time

import numpy as np

from . import geom_oracle as G
from . import rp_oracle as M


def sample_primitives(depth, normal, feat, pts, dataset):
    """getMatchingPrimitive after keypoint detection: 3-D point, normal,
    descriptor per keypoint.  pts [k,2] pixel coords (float64)."""
    h = depth.shape[0]
    pc, nn = G.get_pixel(depth, normal, pts, dataset)
    ptn = pts.copy().astype('float')
    ptn[:, 0] /= 4 * h
    ptn[:, 1] /= h
    des = G.interpolate(feat, ptn.astype(np.float32)).T
    return pc.T, nn, des


def run_pair(net, rgb, norm, depth, pts, ptw, sigmas, dataset='suncg', mask_method='second',
             S=15, alter_steps=3, completion=1, timing=None, detail=None, R_forced=None):
    """One scan pair.  rgb/norm [2,3,h,4h], depth [2,h,4h] float32, pts
    [2,N,2], ptw [2,N], sigmas [steps,4] = (sigmaAngle1, sigmaAngle2, sigmaDist,
    sigmaFeat) per step.  Returns (R_hat 4x4, [R_hat after each step]).
    ``R_forced`` (list of 4x4, one per step) teacher-forces the pose fed to the
    warp of each step -- used by parity tests because with random-init weights
    the matching problem is ill-conditioned and 1e-16 differences in the fit
    grow to 1e-6 within one step (tests/test_oracle_golden.py)."""
    R_hat = np.eye(4)
    views, masks = [], []
    for v in range(2):
        vw, m = G.build_view(rgb[v], norm[v], depth[v], mask_method)
        views.append(vw)
        masks.append(m)
    obs_n = [norm[v].transpose(1, 2, 0) for v in range(2)]
    trace = []
    tm = timing if timing is not None else {}
    for step in range(alter_steps):
        t0 = time.time()
        if R_forced is not None:
            R_hat = R_forced[step]
        t2s = G.warping(views[1], np.linalg.inv(R_hat), dataset).astype(np.float32)
        s2t = G.warping(views[0], R_hat, dataset).astype(np.float32)
        x = np.concatenate((np.concatenate((views[0], t2s), 1), np.concatenate((views[1], s2t), 1)))
        t1 = time.time()
        f = net.forward_pairs(x).numpy()
        t2 = time.time()
        prim = []
        for v in range(2):
            n, d = G.compose(f[v], masks[v], obs_n[v], depth[v])
            pc, nn, des = sample_primitives(d, n, f[v, 7 + S:7 + S + 32], pts[v], dataset)
            w = ptw[v]
            if not completion:
                k = w == 1
                pc, nn, des, w = pc[k], nn[k], des[k], w[k]
            prim.append({'pc': pc, 'normal': nn, 'feat': des, 'weight': w})
        t3 = time.time()
        p = M.Params(*sigmas[step])
        det = {}
        # evaluation.py:280 tests pts3d.shape[0] (always 3) so the helper is
        # always called; its own <3-keypoint exit covers the degenerate case.
        R_hat = M.relative_pose_helper(prim[0], prim[1], p, det)
        t4 = time.time()
        for k, dt in (("warp", t1 - t0), ("scnet", t2 - t1), ("sample", t3 - t2), ("match", t4 - t3)):
            tm[k] = tm.get(k, 0.0) + dt
        trace.append(R_hat.copy())
        if detail is not None:
            detail.append({'x': x, 'f': f, 'prim': prim, 'match': det})
    return R_hat, trace
