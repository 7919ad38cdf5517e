"""ORACLE (test infrastructure, not product code): evaluation-side statistics of the
reference -- util.parse_data (util.py:42-92, the point clouds of the observed
block), util.point_cloud_overlap (util.py:21-40, sklearn KDTree like the reference)
and the pose error metrics of evaluation.py:291-297."""
import numpy as np
from sklearn.neighbors import KDTree

from . import geom_oracle as G


def observed_clouds(depth, dataset):
    """depth [2,h,4h] -> (pc_src, pc_tgt) of the observed face / kinect crop (util.py:43-72, method 'ours')."""
    h = depth.shape[1]
    out = []
    for v in range(2):
        if 'scannet' in dataset:
            y0, y1, x0, x1 = G.kinect_box(h)
            crop = depth[v, y0:y1, x0:x1]
        else:
            crop = depth[v, :, h:2 * h]
        out.append(G.depth2pc(crop, dataset)[0])
    return out


def point_cloud_overlap(pc_src, pc_tgt, R_gt_44):
    src_t = np.matmul(R_gt_44[:3, :3], pc_src.T) + R_gt_44[:3, 3:4]
    d1, _ = KDTree(pc_tgt).query(src_t.T, k=1)
    tgt_t = np.matmul(np.linalg.inv(R_gt_44), np.concatenate((pc_tgt.T, np.ones([1, pc_tgt.shape[0]]))))[:3, :]
    d2, _ = KDTree(pc_src).query(tgt_t.T, k=1)
    ov = max((d1 < 0.08).sum() / pc_src.shape[0], (d2 < 0.08).sum() / pc_tgt.shape[0])
    return ov, np.linalg.norm(R_gt_44[:3, 3]), np.linalg.norm(src_t.mean(1) - pc_tgt.T.mean(1)), (np.min(d1) + np.min(d2)) / 2


def pose_errors(R_hat44, R_gt44, pc_src):
    """evaluation.py:291-297: (angular error deg, translation error, blind angular, blind translation)."""
    R_hat, t_hat = R_hat44[:3, :3], R_hat44[:3, 3]
    R_gt = R_gt44[:3, :3]
    ang = lambda A, B: np.arccos(((np.trace(A @ B.T) - 1) / 2).clip(-1, 1)) / np.pi * 180.0
    ad = ang(R_hat, R_gt)
    ad_blind = ang(R_gt, np.eye(3))
    tr = np.linalg.norm(np.matmul(R_hat - R_gt, pc_src.mean(0).reshape(3)) + t_hat - R_gt44[:3, 3])
    return ad, tr, ad_blind, np.linalg.norm(t_hat - R_gt44[:3, 3])
