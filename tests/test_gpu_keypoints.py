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


# ---- the batched, device-resident derivation (csrc/keypoints.hip; RelativePosePipeline(keypoints="reference")) -----------------------------
@pytest.mark.parametrize("kind,cis", [("second", (0, 1)), ("kinect", (2, 3))])
def test_batched_keypoints_equal_the_reference_goldens(golden_dir, kind, cis):
    """relpose_keypoints_reference on a BATCH of two scan pairs (4 views) against the reference's own getKeypoint / getKeypoint_kinect
    outputs (getkeypoint.npz: the reference run with a cv2 stub returning the fixture's detections, np.random seeded): keypoint
    coordinates and weights of both views, exactly -- descriptors at the selected detections, the fused distance-map + NMS picks (the maps
    are never materialised), validity filter, random fill, concatenation order.  The host half (rputil.keypoint_plan) draws the reference's
    np.random stream from RandomState(seed).  Reference: rputil.py:141-237, :240-353, :355-371."""
    import torch
    from cases import GK_CASES
    from relativepose_amd import rputil, synth
    g = np.load(os.path.join(golden_dir, "getkeypoint.npz"))
    dev = torch.device("cuda:0")
    H, W, S = 160, 640, 15
    off = 7 + S
    f = torch.randn(4, off + 32, H, W, device=dev)                 # the other channels of a network output: ignored
    plans = []
    for b, ci in enumerate(cis):
        k, seed = GK_CASES[ci]
        assert k == kind
        _, _, feats, featt, det_s, det_t, _, _ = synth.make_keypoint_case(seed, kind)
        f[2 * b, off:] = torch.from_numpy(feats).to(dev)
        f[2 * b + 1, off:] = torch.from_numpy(featt).to(dev)
        plans.append(rputil.keypoint_plan(rputil.map_detections(det_s, kind, H), rputil.map_detections(det_t, kind, H), kind, H, W,
                                          np.random.RandomState(seed)))
    tab = rputil.upload_keypoint_tables(rputil.keypoint_tables(plans, H, W), dev)
    pts, w, npts = rputil.keypoints_reference_dev(f, off, tab, kind)
    # the same call through the custom-op surface
    import relativepose_amd.ops  # noqa: F401
    from relativepose_amd.util import MASKS
    o = torch.ops.relpose.keypoints_reference(f, off, tab["q_src"], tab["q_pt"], tab["q_map"], tab["q_off"], tab["nq_view_max"], tab["topk"], 15,
                                              tab["slot_kind"], tab["slot_xy"], MASKS[kind])
    assert torch.equal(o[0], pts) and torch.equal(o[1], w) and torch.equal(o[2], npts)
    pts, w, npts = pts.cpu().numpy(), w.cpu().numpy(), npts.cpu().numpy()
    for b, ci in enumerate(cis):
        for v, (pn, wn) in enumerate((("pts", "ptsW"), ("ptt", "pttW"))):
            ref_p, ref_w = g[f"gk_{ci}_{pn}"], g[f"gk_{ci}_{wn}"]
            n = int(npts[2 * b + v])
            assert n == len(ref_p), (ci, pn, n, len(ref_p))
            assert np.array_equal(pts[2 * b + v, :n], ref_p), (ci, pn)
            assert np.array_equal(w[2 * b + v, :n], ref_w), (ci, wn)
            assert not w[2 * b + v, n:].any()
    log("batched_keypoints_vs_reference_golden", kind=kind, keypoints_per_view=npts.tolist(), queries=int(tab["nq"]))


def _sift_like(rs_, kind, n, h=160):
    if kind == "kinect":
        return np.stack((rs_.uniform(2, 636, n), rs_.uniform(2, 476, n)), 1)
    return np.stack((rs_.uniform(1, h - 3, n), rs_.uniform(1, h - 3, n)), 1)


@pytest.mark.parametrize("ds,kind,S,tanh", [("suncg", "second", 15, 1), ("scannet", "kinect", 21, 1)])
def test_pipeline_reference_keypoints_equal_per_pair_getkeypoint_at_every_level(ds, kind, S, tanh):
    """RelativePosePipeline(keypoints="reference"): 4 scan pairs x 3 free-running levels -- at every level the batched device derivation gives
    each pair exactly the keypoints (coordinates, weights, counts, order) that the per-pair shim rputil.getKeypoint / getKeypoint_kinect
    (pinned to the reference by getkeypoint.npz) computes from THAT level's feature maps of THAT pair with the same np.random seed, and
    the poses are those of a keypoints="given" pipeline fed the same per-level sets.  Reference call site: evaluation.py:278 ->
    rpmodule.getMatchingPrimitive :511-533."""
    import torch
    from types import SimpleNamespace
    from relativepose_amd import params, rputil, synth, weights
    from relativepose_amd.model import SCNet
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    B, h = 4, 160
    d = synth.make_pairs(B, 8100, ds)
    net = SCNet(SimpleNamespace(batchnorm=1, useTanh=tanh, skipLayer=1, outputType="rgbdnsf", snumclass=S))
    net.load_state_dict(weights.make_state_dict(9, S))
    rs_ = np.random.RandomState(55)
    dets = [(_sift_like(rs_, kind, 60 + 7 * b), _sift_like(rs_, kind, 45 + 5 * b)) for b in range(B)]
    sift = [(rputil.map_detections(a, kind, h), rputil.map_detections(c, kind, h)) for a, c in dets]
    seeds = [[1000 * b + lvl + 17 for lvl in range(3)] for b in range(B)]
    pipe = RelativePosePipeline(net, ds, kind, params.final_params(ds), keypoints="reference")
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], None, None, dev, sift=sift, kp_seeds=seeds)
    keep = []
    pose, status, trace = pipe.run(st, keep=keep)
    off = 7 + S
    img = np.zeros((h, 4 * h, 3), np.uint8)
    full = np.zeros((480, 640, 3), np.uint8)
    worst = 0
    for lvl in range(3):
        f = keep[lvl]["f"]
        P, ns, nt = keep[lvl]["pts"].cpu().numpy(), keep[lvl]["ns"].cpu().numpy(), keep[lvl]["nt"].cpu().numpy()
        ws, wt = keep[lvl]["w_s"].cpu().numpy(), keep[lvl]["w_t"].cpu().numpy()
        for b in range(B):
            queue = [dets[b][0], dets[b][1]]
            old = rputil.set_sift_detector(lambda gray: queue.pop(0))
            try:
                np.random.seed(seeds[b][lvl])
                fs, ft = f[2 * b, off:off + 32].contiguous(), f[2 * b + 1, off:off + 32].contiguous()
                res = rputil.getKeypoint_kinect(img, img, fs, ft, full, full) if kind == "kinect" else rputil.getKeypoint(img, img, fs, ft)
            finally:
                rputil.set_sift_detector(old)
            pts, _, ptsW, ptt, _, pttW = res
            assert ns[b] == len(pts) and nt[b] == len(ptt), (lvl, b, ns[b], len(pts), nt[b], len(ptt))
            assert np.array_equal(P[b, 0, :ns[b]], pts) and np.array_equal(P[b, 1, :nt[b]], ptt), (lvl, b)
            assert np.array_equal(ws[b, :ns[b]], ptsW) and np.array_equal(wt[b, :nt[b]], pttW), (lvl, b)
            worst = max(worst, int(ns[b]), int(nt[b]))
    # the poses: the matcher shim on each level's kept primitives at the pair's own (ragged) counts and weights reproduces the level's pose bitwise
    from relativepose_amd import rpmodule
    sig = params.final_params(ds)
    for lvl in range(3):
        k = keep[lvl]
        for b in (0, B - 1):
            n_s, n_t = int(k["ns"][b]), int(k["nt"][b])
            S_ = {"pc": k["pc"][b, 0, :n_s].cpu().numpy(), "normal": k["nn"][b, 0, :n_s].cpu().numpy(), "feat": k["ft"][b, 0, :n_s].cpu().numpy(),
                  "weight": k["w_s"][b, :n_s].cpu().numpy()}
            T_ = {"pc": k["pc"][b, 1, :n_t].cpu().numpy(), "normal": k["nn"][b, 1, :n_t].cpu().numpy(), "feat": k["ft"][b, 1, :n_t].cpu().numpy(),
                  "weight": k["w_t"][b, :n_t].cpu().numpy()}
            helper = rpmodule.RelativePoseEstimation_helper(S_, T_, rpmodule.opts(*sig[lvl]))
            assert np.array_equal(helper, trace[lvl][b].cpu().numpy()), (lvl, b)
    log("pipeline_reference_keypoints", dataset=ds, mask=kind, pairs=B, levels=3, max_keypoints_per_view=worst, status=status.cpu().tolist())
    assert torch.isfinite(pose).all()


# ---- round 6: doCompletion = 0 (the 'ours_nc' method) and views without detections ---------------------------------------------------------
@pytest.mark.parametrize("ci", [0, 2])
@pytest.mark.parametrize("comp", [0, 1])
def test_getmatchingprimitive_shim_equals_the_reference_with_and_without_completion(golden_dir, ci, comp):
    """rpmodule.getMatchingPrimitive (same-named shim over the C ABI) against the REFERENCE's own getMatchingPrimitive(…, doCompletion = 0 | 1)
    (tests/golden/gmp_nc.npz, make_golden.gen_gmp_nc: reference run with the cv2 stub's fixed detections and np.random seeded): doCompletion = 0
    keeps the observed-region keypoints only (rpmodule.py:534-537; the 'ours_nc' method, evaluation.py:74).  All eight outputs."""
    import torch
    from cases import GK_CASES
    from relativepose_amd import rpmodule, rputil, synth
    g = np.load(os.path.join(golden_dir, "gmp_nc.npz"))
    kind, seed = GK_CASES[ci]
    ds, dS, dT, det_s, det_t = synth.make_matching_primitive_case(seed, kind)
    queue = [det_s, det_t]
    old = rputil.set_sift_detector(lambda gray: queue.pop(0))
    try:
        np.random.seed(seed)
        tS = dict(dS, feat=torch.from_numpy(dS["feat"])); tT = dict(dT, feat=torch.from_numpy(dT["feat"]))
        res = rpmodule.getMatchingPrimitive(tS, tT, ds, "skybox", comp)
    finally:
        rputil.set_sift_detector(old)
    err = {}
    for name, a in zip(("pts3d", "ptt3d", "ptsns", "ptsnt", "dess", "dest", "ptsW", "pttW"), res):
        ref = g[f"gmp_{ci}_c{comp}_{name}"]
        assert np.asarray(a).shape == ref.shape, (name, np.asarray(a).shape, ref.shape)
        err[name] = float(np.abs(np.asarray(a, dtype=np.float64) - ref).max()) if ref.size else 0.0
    log("getmatchingprimitive_vs_reference", case=ci, kind=kind, doCompletion=comp, n_source=int(res[0].shape[1]), n_target=int(res[1].shape[1]), max_abs_diff=err)
    assert err["ptsW"] == 0 and err["pttW"] == 0 and err["dess"] == 0 and err["dest"] == 0          # descriptors bit-exact
    assert max(err["pts3d"], err["ptt3d"], err["ptsns"], err["ptsnt"]) < 1e-12
    if not comp:
        assert (np.asarray(res[6]) == 1).all() and (np.asarray(res[7]) == 1).all()


@pytest.mark.parametrize("kind,cis", [("second", (0, 1)), ("kinect", (2, 3))])
def test_batched_keypoints_observed_only_equal_the_reference_selection(golden_dir, kind, cis):
    """RELPOSE_KP_OBSERVED_ONLY (RelativePosePipeline(keypoints="reference", completion=0)): the batched device derivation keeps exactly the
    keypoints the reference keeps with doCompletion = 0 -- its own keypoint lists (getkeypoint.npz) filtered by its own weights == 1, in order
    (rpmodule.py:534-537) -- and as many as the reference's getMatchingPrimitive(…, 0) returned (gmp_nc.npz)."""
    import torch
    from cases import GK_CASES
    from relativepose_amd import rputil, synth
    g = np.load(os.path.join(golden_dir, "getkeypoint.npz"))
    gn = np.load(os.path.join(golden_dir, "gmp_nc.npz"))
    dev = torch.device("cuda:0")
    H, W, S = 160, 640, 15
    off = 7 + S
    f = torch.randn(4, off + 32, H, W, device=dev)
    plans = []
    for b, ci in enumerate(cis):
        k, seed = GK_CASES[ci]
        _, _, feats, featt, det_s, det_t, _, _ = synth.make_keypoint_case(seed, kind)
        f[2 * b, off:] = torch.from_numpy(feats).to(dev)
        f[2 * b + 1, off:] = torch.from_numpy(featt).to(dev)
        plans.append(rputil.keypoint_plan(rputil.map_detections(det_s, kind, H), rputil.map_detections(det_t, kind, H), kind, H, W, np.random.RandomState(seed)))
    tab = rputil.upload_keypoint_tables(rputil.keypoint_tables(plans, H, W), dev)
    pts, w, npts = (t.cpu().numpy() for t in rputil.keypoints_reference_dev(f, off, tab, kind, observed_only=True))
    for b, ci in enumerate(cis):
        for v, (pn, wn, gname) in enumerate((("pts", "ptsW", "pts3d"), ("ptt", "pttW", "ptt3d"))):
            keep = g[f"gk_{ci}_{wn}"] == 1
            n = int(npts[2 * b + v])
            assert n == int(keep.sum()) and np.array_equal(pts[2 * b + v, :n], g[f"gk_{ci}_{pn}"][keep]), (ci, pn)
            assert (w[2 * b + v, :n] == 1).all() and not w[2 * b + v, n:].any()
            if f"gmp_{ci}_c0_{gname}" in gn:
                assert n == gn[f"gmp_{ci}_c0_{gname}"].shape[1], (ci, gname)
    log("batched_keypoints_observed_only", kind=kind, keypoints_per_view=npts.tolist())


def test_pipeline_pair_without_detections_returns_identity_and_leaves_the_batch_alone():
    """a view without SIFT detections must not abort the batch.  The reference's getKeypoint returns None, that level's pose is
    the identity and the loop goes on (rputil.py:156-166, rpmodule.py:522-523, evaluation.py:280-282).  Batched: the pair gets an empty keypoint
    plan at every level -> 0 keypoints -> the matcher's "return identity" status; the other pairs' poses are bitwise what they are without it;
    completion = 0 runs the same way."""
    import torch
    from types import SimpleNamespace
    from relativepose_amd import params, rputil, synth, weights
    from relativepose_amd.model import SCNet
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    ds, kind, S, h = "suncg", "second", 15, 160
    d = synth.make_pairs(3, 8300, ds)
    net = SCNet(SimpleNamespace(batchnorm=1, useTanh=1, skipLayer=1, outputType="rgbdnsf", snumclass=S))
    net.load_state_dict(weights.make_state_dict(9, S))
    rs_ = np.random.RandomState(56)
    dets = [(_sift_like(rs_, kind, 50 + 5 * b), _sift_like(rs_, kind, 40 + 5 * b)) for b in range(3)]
    sift = [(rputil.map_detections(a, kind, h), rputil.map_detections(c, kind, h)) for a, c in dets]
    seeds = [[100 * b + lvl + 3 for lvl in range(3)] for b in range(3)]
    for completion in (1, 0):
        pipe = RelativePosePipeline(net, ds, kind, params.final_params(ds), keypoints="reference", completion=completion)
        pose_all, status_all, _ = pipe.run(pipe.prepare(d["rgb"], d["norm"], d["depth"], None, None, dev, sift=sift, kp_seeds=seeds))
        sift_gap = [sift[0], (np.zeros((0, 2)), sift[1][1]), sift[2]]           # pair 1: the source view has no detections
        keep = []
        pose, status, _ = pipe.run(pipe.prepare(d["rgb"], d["norm"], d["depth"], None, None, dev, sift=sift_gap, kp_seeds=seeds), keep=keep)
        assert int(status[1]) == 1 and torch.equal(pose[1], torch.eye(4, dtype=torch.float64, device=dev))      # RELPOSE_FEW_KEYPOINTS: identity
        assert all(int(k["ns"][1]) == 0 and int(k["nt"][1]) == 0 for k in keep)
        for b in (0, 2):
            assert torch.equal(pose[b], pose_all[b]) and int(status[b]) == int(status_all[b]), (completion, b)
        if not completion:
            assert all(bool((k["w_s"][b, :int(k["ns"][b])] == 1).all()) and bool((k["w_t"][b, :int(k["nt"][b])] == 1).all()) for k in keep for b in (0, 2))
    log("pipeline_pair_without_detections", status=status.cpu().tolist())
