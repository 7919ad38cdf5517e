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


@pytest.mark.parametrize("ci", range(4))
def test_getkeypoint_assembly_matches_reference(golden_dir, ci):
    """rputil.getKeypoint / getKeypoint_kinect with the SIFT detections given (rputil.set_sift_detector returning the fixture's fixed
    points -- the same ones the reference was fed through a cv2 stub, make_golden.gen_getkeypoint) and np.random seeded like the
    golden run: descriptor sampling at the detections, the [<=30, H, W] cross-view distance maps + NMS, the random fill with its own
    augmentation and the observed-region weights equal the reference's output (coordinates exactly, normalised coordinates to
    round-off)."""
    import torch
    from cases import GK_CASES
    from relativepose_amd import rputil, synth
    g = np.load(os.path.join(golden_dir, "getkeypoint.npz"))
    kind, seed = GK_CASES[ci]
    rs, rt, feats, featt, det_s, det_t, rs_full, rt_full = synth.make_keypoint_case(seed, kind)
    queue = [det_s, det_t]
    seen = []

    def detector(gray):
        seen.append(gray.shape)
        return queue.pop(0)
    old = rputil.set_sift_detector(detector)
    try:
        np.random.seed(seed)
        if kind == "kinect":
            res = rputil.getKeypoint_kinect(rs, rt, torch.from_numpy(feats), torch.from_numpy(featt), rs_full, rt_full)
        else:
            res = rputil.getKeypoint(rs, rt, torch.from_numpy(feats), torch.from_numpy(featt))
    finally:
        rputil.set_sift_detector(old)
    assert seen == ([(480, 640)] * 2 if kind == "kinect" else [(160, 160)] * 2)       # SIFT sees the full kinect frame / the observed face
    names = ("pts", "ptsNorm", "ptsW", "ptt", "pttNorm", "pttW")
    diffs = {}
    for name, a in zip(names, res):
        ref = g[f"gk_{ci}_{name}"]
        assert a.shape == ref.shape, (name, a.shape, ref.shape)
        diffs[name] = float(np.abs(np.asarray(a, dtype=np.float64) - ref).max())
    log("getkeypoint_assembly", case=ci, kind=kind, n_source=len(res[0]), n_target=len(res[3]), max_abs_diff=diffs)
    assert diffs["pts"] == 0 and diffs["ptt"] == 0 and diffs["ptsW"] == 0 and diffs["pttW"] == 0
    assert diffs["ptsNorm"] < 1e-15 and diffs["pttNorm"] < 1e-15


def test_getkeypoint_without_detections_returns_nones():
    import torch
    from relativepose_amd import rputil, synth
    rs, rt, feats, featt, *_ = synth.make_keypoint_case(31, "second")
    old = rputil.set_sift_detector(lambda gray: np.zeros((0, 2)))
    try:
        assert rputil.getKeypoint(rs, rt, torch.from_numpy(feats), torch.from_numpy(featt)) == (None,) * 6
    finally:
        rputil.set_sift_detector(old)
