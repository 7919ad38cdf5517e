"""Free-running recurrent loop on the GPU (the pose estimate of level k is fed back into the warp of level k+1,
evaluation.py:232-284) against the REFERENCE's poses after each level.

Three kinds of fixtures (all captured from the reference by tests/golden/make_golden.py):

* ``e2e.npz``      random-init weights, 6 scan pairs (4 SUNCG = BASELINE configs[0], 1 Matterport, 1 ScanNet).  With
                   random weights the descriptors carry no information and the loop is CHAOTIC: ``e2e_env.npz`` holds the
                   reference's own response to uniform(+-3e-5) noise on the network output (the size of the float32
                   kernel-vs-reference difference) -- ~1e-3 after level 0 and O(1) (an unrelated pose) after levels 1-2.
                   The GPU loop is asserted to stay inside that measured envelope; both numbers are logged.
* ``e2e_wc.npz``   WELL-CONDITIONED pairs (synth.make_wc_pair + weights.make_descriptor_state_dict: keypoints are
                   projections of common world points, descriptors follow the view-invariant texture).  The reference's
                   envelope there is ~1e-7, and the free-running GPU pose is asserted within the north-star bar,
                   1e-4 Frobenius on the rotation, after EVERY level.
* BASELINE configs[1] size (32 pairs, N=200): batch result bitwise == single-pair runs, all status 0.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from cases import E2E_CASES, E2E_N, E2E_WEIGHT_SEED, ENV_AMP, WC2_CASES, WC_CASES, WC_KW, WC_S, WC_SIGMAS, WC_WEIGHT_SEED
from gpu_util import log
from relativepose_amd import synth, weights

pytestmark = pytest.mark.gpu

ENV_SLACK = 3.0      # the GPU run is one more sample of a heavy-tailed distribution of which the envelope holds 8


def _net(S, tanh, sd, prec="f32"):
    from relativepose_amd.model import SCNet
    net = SCNet(SimpleNamespace(batchnorm=1, useTanh=tanh, skipLayer=1, outputType="rgbdnsf", snumclass=S))
    net.load_state_dict(sd)
    net.set_precision(prec)
    return net


def _rot_err(a, b):
    return float(np.linalg.norm(a[:3, :3] - b[:3, :3]))


@pytest.mark.parametrize("ci", range(len(E2E_CASES)))
def test_free_running_random_weights_inside_reference_envelope(ci, golden_dir):
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    ge = np.load(os.path.join(golden_dir, "e2e.npz"))
    gm = np.load(os.path.join(golden_dir, "matcher.npz"))
    env = np.load(os.path.join(golden_dir, "e2e_env.npz"))
    assert float(env["amp"]) == ENV_AMP
    ds, mm, S, tanh, seed = E2E_CASES[ci]
    d = synth.make_pairs(1, seed, ds)
    pts, ptw = synth.make_keypoints(1, E2E_N, seed, mm)
    dev = torch.device("cuda:0")
    pipe = RelativePosePipeline(_net(S, tanh, weights.make_state_dict(E2E_WEIGHT_SEED, S)), ds, mm, gm[f"params_{ds}"])
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    pose, status, trace = pipe.run(st)                     # free running: no R_forced
    e = env[f"env_{ci}"]                                   # [seeds, 3]
    errs = [_rot_err(trace[s][0].cpu().numpy(), ge[f"e2e_{ci}_R{s}"]) for s in range(3)]
    log("e2e_free_running", case=ci, ds=ds, gpu_rot_err_vs_reference=errs, reference_envelope_max=e.max(0), reference_envelope_median=np.median(e, 0),
        noise_amp=ENV_AMP)
    assert int(status[0]) == 0
    # level 0 is asserted against the reference's own perturbation envelope; after levels 1-2 that envelope is O(1) (an unrelated
    # pose: no bound on a rotation difference could fail there), so those levels are logged above and only checked to be proper poses
    assert errs[0] <= ENV_SLACK * e[:, 0].max(), (errs[0], ENV_SLACK * e[:, 0].max())
    for s in range(3):
        Rg = trace[s][0].cpu().numpy()[:3, :3]
        assert np.allclose(Rg @ Rg.T, np.eye(3), atol=1e-9) and np.linalg.det(Rg) > 0.999, s


@pytest.mark.parametrize("ci", range(len(WC_CASES)))
def test_free_running_well_conditioned_within_1e4(ci, golden_dir):
    """The north-star parity bar on the whole loop: rotation within 1e-4 Frobenius of the reference after every level,
    with the GPU's own pose fed back (pose_inverse -> warp_pairs -> SCNet -> sample -> match)."""
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    g = np.load(os.path.join(golden_dir, "e2e_wc.npz"))
    d, pts, ptw, T = synth.make_wc_pair(WC_CASES[ci], **WC_KW)
    assert np.array_equal(T, g[f"wc_{ci}_T"])
    dev = torch.device("cuda:0")
    sig = np.tile(np.array([WC_SIGMAS]), (3, 1))
    pipe = RelativePosePipeline(_net(WC_S, 1, weights.make_descriptor_state_dict(WC_WEIGHT_SEED, WC_S)), "suncg", "second", sig)
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    pose, status, trace = pipe.run(st)
    errs = [_rot_err(trace[s][0].cpu().numpy(), g[f"wc_{ci}_R{s}"]) for s in range(3)]
    terr = [float(np.linalg.norm(trace[s][0].cpu().numpy()[:3, 3] - g[f"wc_{ci}_R{s}"][:3, 3])) for s in range(3)]
    e = g[f"wc_env_{ci}"]
    log("e2e_wc_free_running", case=ci, seed=WC_CASES[ci], gpu_rot_err_vs_reference=errs, gpu_trans_err_vs_reference=terr,
        reference_envelope_max=e.max(0), rot_err_vs_true_motion=[_rot_err(trace[s][0].cpu().numpy(), T) for s in range(3)])
    assert int(status[0]) == 0
    assert e.max() < 1e-5, "fixture is not well-conditioned"
    for s in range(3):
        assert errs[s] < 1e-4, (s, errs)
        assert terr[s] < 1e-4, (s, terr)


@pytest.mark.parametrize("ci", range(len(WC2_CASES)))
def test_free_running_well_conditioned_other_conventions_within_1e4(ci, golden_dir):
    """The 1e-4 bar on the whole free-running loop under the OTHER two dataset conventions (e2e_wc2.npz): Matterport ('second' mask,
    S=21, face rotations Rs[(i-1)%4], util.py:119-158) and ScanNet ('kinect' mask: 66x88 observed crop with the intrinsics scaling of
    util.py:468-523, no tanh on the descriptors) -- the GPU's own pose fed back through pose_inverse -> warp_pairs -> SCNet ->
    sample -> match, against the reference's pose after every level."""
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    g = np.load(os.path.join(golden_dir, "e2e_wc2.npz"))
    ds, mm, S, tanh, seed, kw = WC2_CASES[ci]
    d, pts, ptw, T = synth.make_wc_pair(seed, dataset=ds, mask_method=mm, **kw)
    assert np.array_equal(T, g[f"wc2_{ci}_T"])
    dev = torch.device("cuda:0")
    sig = np.tile(np.array([WC_SIGMAS]), (3, 1))
    pipe = RelativePosePipeline(_net(S, tanh, weights.make_descriptor_state_dict(WC_WEIGHT_SEED, S)), ds, mm, sig)
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    pose, status, trace = pipe.run(st)
    errs = [_rot_err(trace[s][0].cpu().numpy(), g[f"wc2_{ci}_R{s}"]) for s in range(3)]
    terr = [float(np.linalg.norm(trace[s][0].cpu().numpy()[:3, 3] - g[f"wc2_{ci}_R{s}"][:3, 3])) for s in range(3)]
    e = g[f"wc2_env_{ci}"]
    log("e2e_wc2_free_running", case=ci, dataset=ds, mask=mm, seed=seed, gpu_rot_err_vs_reference=errs, gpu_trans_err_vs_reference=terr,
        reference_envelope_max=e.max(0), rot_err_vs_true_motion=[_rot_err(trace[s][0].cpu().numpy(), T) for s in range(3)])
    assert int(status[0]) == 0
    assert e.max() < 1e-5, "fixture is not well-conditioned"
    for s in range(3):
        assert errs[s] < 1e-4, (s, errs)
        assert terr[s] < 1e-4, (s, terr)


# ---- the 16-bit conv arithmetic of BASELINE configs[4] on the reference-pinned fixtures (VERDICT r4 next #1a) ------------------------------
# configs[4] itself (320x1280) is unpinnable: the reference asserts 160x640 panoramas.  Its ARITHMETIC is pinnable: the seven well-conditioned
# fixtures above are reference outputs at h=160, and the conv kernels do not depend on the panorama size (every input is resized to 224x224).
#   f16x3  (3 fp16 MFMA terms per fp32 product: what `bench.py --config 4` runs)  -> the north-star bar, 1e-4, after every level
#   f16    (one fp16 MFMA per product, SURVEY 8d config 5's literal "fp16 MFMA convs": an accuracy trade the caller opts into)
#          -> asserted against F16_ROT_BOUND / F16_TRANS_BOUND, measured on the MI355X (profiles/r05_parity_full_suite.jsonl) with 3x slack;
#          it does NOT meet the 1e-4 parity bar on every fixture and bench.py's `dtype` string says so.
# measured (MI355X, round 5): f16x3 <= 7.9e-8 rotation / 1.1e-7 translation on all seven fixtures and levels; plain f16 <= 8.7e-6 / 2.4e-5 after
# level 0 but up to 3.9e-3 / 7.8e-3 after level 2 (the feedback loop amplifies the 2^-11 product error): the bounds below are 3x that
F16_ROT_BOUND = 1.2e-2
EXACT_BAR = 1e-6
F16_TRANS_BOUND = 2.5e-2
_WC_ALL = [("wc", i) for i in range(len(WC_CASES))] + [("wc2", i) for i in range(len(WC2_CASES))]


def _wc_fixture(kind, ci, golden_dir):
    if kind == "wc":
        g = np.load(os.path.join(golden_dir, "e2e_wc.npz"))
        d, pts, ptw, T = synth.make_wc_pair(WC_CASES[ci], **WC_KW)
        return ("suncg", "second", WC_S, 1, d, pts, ptw, T, [g[f"wc_{ci}_R{s}"] for s in range(3)], g[f"wc_{ci}_T"], g[f"wc_env_{ci}"])
    g = np.load(os.path.join(golden_dir, "e2e_wc2.npz"))
    ds, mm, S, tanh, seed, kw = WC2_CASES[ci]
    d, pts, ptw, T = synth.make_wc_pair(seed, dataset=ds, mask_method=mm, **kw)
    return (ds, mm, S, tanh, d, pts, ptw, T, [g[f"wc2_{ci}_R{s}"] for s in range(3)], g[f"wc2_{ci}_T"], g[f"wc2_env_{ci}"])


@pytest.mark.parametrize("prec", ["bf16x9", "bf16x6", "f16x3", "f16"])
@pytest.mark.parametrize("kind,ci", _WC_ALL)
def test_free_running_well_conditioned_16bit_conv_arithmetic(kind, ci, prec, golden_dir):
    """The whole free-running loop (the GPU's own pose fed back, three levels) with the conv stack in the 16-bit MFMA modes against the
    REFERENCE's pose after every level, on all seven reference-pinned fixtures (SUNCG, Matterport, ScanNet conventions).
    Reference: model/mymodel.py:15-39 (conv2d / deconv2d blocks), evaluation.py:232-284."""
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    ds, mm, S, tanh, d, pts, ptw, T, Rref, Tg, env = _wc_fixture(kind, ci, golden_dir)
    assert np.array_equal(T, Tg)
    dev = torch.device("cuda:0")
    sig = np.tile(np.array([WC_SIGMAS]), (3, 1))
    pipe = RelativePosePipeline(_net(S, tanh, weights.make_descriptor_state_dict(WC_WEIGHT_SEED, S), prec), ds, mm, sig)
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    pose, status, trace = pipe.run(st)
    errs = [_rot_err(trace[s][0].cpu().numpy(), Rref[s]) for s in range(3)]
    terr = [float(np.linalg.norm(trace[s][0].cpu().numpy()[:3, 3] - Rref[s][:3, 3])) for s in range(3)]
    log("e2e_wc_free_running_16bit", fixture=kind, case=ci, dataset=ds, mask=mm, conv_precision=prec, gpu_rot_err_vs_reference=errs,
        gpu_trans_err_vs_reference=terr, reference_envelope_max=env.max(0), bar=EXACT_BAR if prec in ("bf16x9", "bf16x6") else 1e-4 if prec == "f16x3" else F16_ROT_BOUND)
    assert int(status[0]) == 0
    # bf16x9 / bf16x6 (exact-product emulation of the fp32 contraction, round 6: the headline's arithmetic): 1e-6, the bar the round-5 verdict set
    rb, tb = (EXACT_BAR, EXACT_BAR) if prec in ("bf16x9", "bf16x6") else (1e-4, 1e-4) if prec == "f16x3" else (F16_ROT_BOUND, F16_TRANS_BOUND)
    for s in range(3):
        assert errs[s] < rb, (prec, s, errs)
        assert terr[s] < tb, (prec, s, terr)
    if prec == "f16":
        assert errs[0] < 1e-4 and terr[0] < 1e-4, (errs, terr)       # level 0 (no feedback yet) still meets the bar


def test_wc_teacher_forcing_changes_nothing_but_feedback_matters(golden_dir):
    """The feedback path is live on the well-conditioned fixture: forcing a WRONG pose into level 1 moves the level-1
    result (so the <1e-4 agreement above does pin pose_inverse -> warp_pairs with the GPU's own pose)."""
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    d, pts, ptw, T = synth.make_wc_pair(WC_CASES[0], **WC_KW)
    dev = torch.device("cuda:0")
    sig = np.tile(np.array([WC_SIGMAS]), (3, 1))
    pipe = RelativePosePipeline(_net(WC_S, 1, weights.make_descriptor_state_dict(WC_WEIGHT_SEED, WC_S)), "suncg", "second", sig, alter_steps=2)
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    _, _, free = pipe.run(st)
    wrong = synth.random_rigid(np.random.RandomState(3), 0.6, 0.5)
    forced = [torch.eye(4, dtype=torch.float64, device=dev)[None], torch.from_numpy(wrong[None]).to(dev)]
    _, _, tf = pipe.run(st, R_forced=forced)
    assert torch.equal(free[0], tf[0])
    moved = _rot_err(free[1][0].cpu().numpy(), tf[1][0].cpu().numpy())
    log("e2e_wc_feedback_sensitivity", level1_rot_change_with_wrong_pose=moved)
    assert moved > 1e-6


def test_baseline_config1_batch_equals_single_pairs():
    """BASELINE configs[1] size: 32 SUNCG pairs x 200 keypoints in one batch (workspace sizing, the 2048-entry LDS
    scale/shift table at 64 images, max_edges as the bench sets it): bitwise equal to single-pair runs for sampled pairs,
    every status 0, deterministic run to run."""
    import torch
    from relativepose_amd import params
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    B, N, S = 32, 200, 15
    d = synth.make_pairs(B, 2000, "suncg")
    pts, ptw = synth.make_keypoints(B, N, 2000, "second")
    net = _net(S, 1, weights.make_state_dict(7, S))
    Cc = N * 5
    pipe = RelativePosePipeline(net, "suncg", "second", params.final_params("suncg"), max_edges=min(Cc * (Cc - 1), 1 << 20))
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    pose, status, _ = pipe.run(st)
    pose2, status2, _ = pipe.run(st)
    assert torch.equal(pose, pose2) and torch.equal(status, status2)
    assert (status == 0).all(), status.cpu().tolist()
    assert torch.isfinite(pose).all()
    for b in (0, 7, 18, 31):
        st1 = pipe.prepare(d["rgb"][b:b + 1], d["norm"][b:b + 1], d["depth"][b:b + 1], pts[b:b + 1], ptw[b:b + 1], dev)
        p1, s1, _ = pipe.run(st1)
        assert torch.equal(p1[0], pose[b]) and int(s1[0]) == 0, b
    R = pose[:, :3, :3]
    orth = (R @ R.transpose(1, 2) - torch.eye(3, dtype=torch.float64, device=dev)).abs().max().item()
    log("config1_batch32", orthogonality=orth)
    assert orth < 1e-9
