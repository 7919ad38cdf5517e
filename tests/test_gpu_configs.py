"""BASELINE.json configs[2..4] at their stated sizes through the device pipeline:

* configs[2]  Matterport conventions, N=400 keypoints (160 000-entry affinity, C=2000 correspondences), 32 pairs
* configs[3]  ScanNet conventions ('kinect' mask, 66x88 observed crop), 32 pairs per GPU
  property checks: batch == single pair bitwise (sampled pairs), all status 0, run-to-run deterministic, and the matcher
  on the batch's own level-0 primitives of a sampled pair vs the oracle helper (<1e-4 Frobenius on the rotation).
* configs[4]  SUNCG 320x1280, f16x3 conv arithmetic ("fp16 MFMA conv path"), 3 recurrent levels vs the parameterised
  oracle (teacher-forced per level; the oracle is validated against the reference at h=160 only: parity unpinned at h=320).
"""
from types import SimpleNamespace

import numpy as np
import pytest

from gpu_util import log
from oracle import pipeline_oracle as P
from oracle import rp_oracle as M
from oracle.scnet_oracle import SCNetOracle
from relativepose_amd import params, synth, weights

pytestmark = pytest.mark.gpu


def _net(S, tanh, sd, prec="f32"):
    from relativepose_amd.model import SCNet
    net = SCNet(SimpleNamespace(batchnorm=1, useTanh=tanh, skipLayer=1, outputType="rgbdnsf", snumclass=S))
    net.load_state_dict(sd)
    net.set_precision(prec)
    return net


@pytest.mark.parametrize("ds,mm,S,N,seed", [("matterport", "second", 21, 400, 3000), ("scannet", "kinect", 21, 200, 4000)])
def test_config_batch32_properties_and_matcher_vs_oracle(ds, mm, S, N, seed):
    import torch
    from relativepose_amd import rpmodule
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    B = 32
    d = synth.make_pairs(B, seed, ds)
    pts, ptw = synth.make_keypoints(B, N, seed, mm)
    sig = params.final_params(ds)
    Cc = N * 5
    pipe = RelativePosePipeline(_net(S, 1, weights.make_state_dict(7, S)), ds, mm, sig, max_edges=min(Cc * (Cc - 1), (1 << 20) * (N // 200) ** 2))
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    keep = []
    pose, status, trace = pipe.run(st, keep=keep)
    prim0 = [{k: keep[0][k][0, v].cpu().numpy() for k in ("pc", "nn", "ft")} for v in range(2)]
    lvl0_pose = trace[0][0].cpu().numpy()
    del keep
    pose2, status2, _ = pipe.run(st)
    assert torch.equal(pose, pose2) and torch.equal(status, status2)
    assert (status == 0).all(), status.cpu().tolist()
    for b in (3, 29):
        st1 = pipe.prepare(d["rgb"][b:b + 1], d["norm"][b:b + 1], d["depth"][b:b + 1], pts[b:b + 1], ptw[b:b + 1], dev)
        p1, _, _ = pipe.run(st1)
        assert torch.equal(p1[0], pose[b]), b
    # matcher of pair 0, level 0, on the GPU's own primitives vs the oracle helper
    S_ = {"pc": prim0[0]["pc"], "normal": prim0[0]["nn"], "feat": prim0[0]["ft"], "weight": ptw[0, 0]}
    T_ = {"pc": prim0[1]["pc"], "normal": prim0[1]["nn"], "feat": prim0[1]["ft"], "weight": ptw[0, 1]}
    det = {}
    ref = M.relative_pose_helper(S_, T_, M.Params(*sig[0]), det)
    err = float(np.linalg.norm(lvl0_pose[:3, :3] - ref[:3, :3]))
    helper = rpmodule.RelativePoseEstimation_helper(S_, T_, rpmodule.opts(*sig[0]))
    log("config_batch32", ds=ds, N=N, matcher_rot_err_vs_oracle=err, oracle_status=int(det.get("status", 0)))
    assert np.array_equal(helper, lvl0_pose)          # same kernels, same inputs: batch of 32 == batch of 1
    assert err < 1e-4


def test_config4_320x1280_f16x3_three_levels_vs_parameterised_oracle():
    import torch
    from relativepose_amd import rpmodule
    from relativepose_amd.pipeline import RelativePosePipeline
    ds, mm, h, S, tanh, N = "suncg", "second", 320, 15, 1, 64
    d = synth.make_pairs(1, 5200, ds, h=h)
    pts, ptw = synth.make_keypoints(1, N, 5200, mm, h=h)
    sd = weights.make_state_dict(11, S)
    sig = np.array(params.final_params(ds))
    rs = np.random.RandomState(2)
    forced = [np.eye(4), synth.random_rigid(rs, 0.8, 0.5), synth.random_rigid(rs, 0.5, 0.3)]
    det = []
    _, otrace = P.run_pair(SCNetOracle(sd, S, tanh), d["rgb"][0], d["norm"][0], d["depth"][0], pts[0], ptw[0], sig, ds, mm, S, alter_steps=3,
                           detail=det, R_forced=forced)
    dev = torch.device("cuda:0")
    pipe = RelativePosePipeline(_net(S, tanh, sd, "f16x3"), ds, mm, sig, alter_steps=3)
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    keep = []
    pose, status, trace = pipe.run(st, R_forced=[torch.from_numpy(f[None]).to(dev) for f in forced], keep=keep)
    for step in range(3):
        nbad = int((keep[step]["x"].cpu().numpy() != det[step]["x"]).any(1).sum())
        f_err = np.abs(keep[step]["f"].cpu().numpy() - det[step]["f"]).max()
        prim = det[step]["prim"]
        pc_err = max(np.abs(keep[step]["pc"][0, v].cpu().numpy() - prim[v]["pc"]).max() for v in range(2))
        ft_err = max(np.abs(keep[step]["ft"][0, v].cpu().numpy() - prim[v]["feat"]).max() for v in range(2))
        iso = rpmodule.RelativePoseEstimation_helper(prim[0], prim[1], rpmodule.opts(*sig[step]))
        iso_err = float(np.linalg.norm(iso[:3, :3] - otrace[step][:3, :3]))
        log("config4_320_f16x3", step=step, net_input_pixels_differ=nbad, net_out_abs_err=f_err, pc_err=pc_err, feat_err=ft_err,
            matcher_iso_rot_err=iso_err)
        assert nbad <= 8 and f_err < 1e-3 and pc_err < 1e-3 and ft_err < 1e-3 and iso_err < 1e-4
    assert int(status[0]) == 0


def test_config4_full_size_batch32_320x1280_f16x3_properties():
    """BASELINE configs[4] at its BENCH size (32 scan pairs x 320x1280, f16x3 conv arithmetic, 3 free-running levels, N=200, max_edges as the
    bench sets it): run-to-run deterministic (bitwise), every status 0, proper rotations, and bitwise equal to single-pair runs for sampled
    pairs -- the same properties tests/test_gpu_e2e.py asserts for configs[1] and the test above for configs[2-3] (VERDICT r4 next #1b).
    (Values are unpinnable at h=320 -- the reference asserts 160x640; the arithmetic itself is pinned at h=160 by
    test_free_running_well_conditioned_16bit_conv_arithmetic.)"""
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    ds, mm, h, S, N, B = "suncg", "second", 320, 15, 200, 32
    d = synth.make_pairs(B, 5000, ds, h=h)
    pts, ptw = synth.make_keypoints(B, N, 5000, mm, h=h)
    Cc = N * 5
    pipe = RelativePosePipeline(_net(S, 1, weights.make_state_dict(7, S), "f16x3"), ds, mm, params.final_params(ds), max_edges=min(Cc * (Cc - 1), 1 << 20))
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    pose, status, _ = pipe.run(st)
    pose2, status2, _ = pipe.run(st)
    assert torch.equal(pose, pose2) and torch.equal(status, status2)
    assert (status == 0).all(), status.cpu().tolist()
    assert torch.isfinite(pose).all()
    R = pose[:, :3, :3]
    orth = (R @ R.transpose(1, 2) - torch.eye(3, dtype=torch.float64, device=dev)).abs().max().item()
    for b in (0, 13, 31):
        st1 = pipe.prepare(d["rgb"][b:b + 1], d["norm"][b:b + 1], d["depth"][b:b + 1], pts[b:b + 1], ptw[b:b + 1], dev)
        p1, s1, _ = pipe.run(st1)
        assert torch.equal(p1[0], pose[b]) and int(s1[0]) == 0, b
    log("config4_batch32_320x1280_f16x3", orthogonality=orth, status_ok=int((status == 0).sum().item()))
    assert orth < 1e-9
