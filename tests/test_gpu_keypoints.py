"""GPU: descriptor-to-map distance + NMS Sampling vs the oracle and the reference's Sampling."""
import os

import numpy as np
import pytest

from gpu_util import log
from oracle import geom_oracle as G
from test_oracle_golden import _kp_inputs

pytestmark = pytest.mark.gpu


def test_distance_map_and_sampling(golden_dir):
    import torch
    from relativepose_amd import rputil
    gk = np.load(os.path.join(golden_dir, "keypoints.npz"))
    dev = torch.device("cuda:0")
    for tag, fs, featt in _kp_inputs():
        dist = rputil.feature_distance_map_dev(torch.from_numpy(np.ascontiguousarray(fs.T)).to(dev), torch.from_numpy(featt).to(dev))
        d_o = G.feature_distance_map(fs, featt)
        rel = float(np.abs(dist.cpu().numpy() - d_o).max() / d_o.max())
        pts = rputil.sampling_dev(dist, 2).cpu().numpy()
        pts_o = G.sampling(d_o.copy(), 2)
        log("keypoint_sampling", case=tag, dist_rel_err=rel, pts_equal_oracle=bool(np.array_equal(pts, pts_o)),
            pts_equal_reference=bool(np.array_equal(pts, gk[f"kp_{tag}_pts"])))
        assert rel < 1e-6
        assert np.array_equal(pts, gk[f"kp_{tag}_pts"])
        # Sampling on the oracle's own maps through the numpy-signature mirror
        assert np.array_equal(rputil.Sampling(d_o, 2), pts_o)
