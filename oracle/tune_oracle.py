"""ORACLE (test infrastructure, not product code): the sigma-tuning objective of
/root/reference/trainRelativePoseModuleRecFD.py:215-233 as a Python loop over the
cached primitives with the numpy matcher oracle."""
import numpy as np

from . import rp_oracle as M


def objective(primitives, para):
    loss, ad = 0.0, 0.0
    for p in primitives:
        S = {'pc': p['pc_src'], 'normal': p['normal_src'], 'feat': p['feat_src'], 'weight': p['weight_src']}
        T = {'pc': p['pc_tgt'], 'normal': p['normal_tgt'], 'feat': p['feat_tgt'], 'weight': p['weight_tgt']}
        q = M.Params(para.sigmaAngle1, para.sigmaAngle2, para.sigmaDist, para.sigmaFeat)
        R_hat = M.relative_pose_helper(S, T, q)
        loss += np.power(R_hat[:3, :3] - p['R_gt'][:3, :3], 2).sum()
        tr = np.trace(R_hat[:3, :3] @ p['R_gt'][:3, :3].T)
        ad += np.arccos(np.clip((tr - 1) / 2, -1, 1)) / np.pi * 180.0
    return loss / len(primitives), ad / len(primitives)
