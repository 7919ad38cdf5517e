"""Host-side formats and bookkeeping (no GPU): the reference's checkpoint container, the tuned-parameter files, the pose
error metrics and the .result.npy record list, and the torch.ops.relpose schema registration."""
import os

import numpy as np

from oracle import stats_oracle as SO
from relativepose_amd import evaluation as E
from relativepose_amd import params, synth, weights


def test_load_checkpoint_reference_container(tmp_path):
    """evaluation.py:143-153: torch.load(path)['state_dict'], optionally with DataParallel 'module.' keys."""
    import torch
    sd = weights.make_state_dict(3, 15)
    ck = {"epoch": 7, "state_dict": {"module." + k: torch.from_numpy(v) for k, v in sd.items()}, "optimizer": {"lr": 1e-3}}
    p = str(tmp_path / "suncg.comp.pth.tar")
    torch.save(ck, p)
    got = weights.load_checkpoint(p)
    assert list(got) == list(sd)
    assert all(np.array_equal(got[k], sd[k]) and got[k].dtype == np.float32 for k in sd)
    torch.save({k: torch.from_numpy(v).double() for k, v in sd.items()}, p)        # a bare state dict, float64 tensors
    got = weights.load_checkpoint(p)
    assert all(np.array_equal(got[k], sd[k]) and got[k].dtype == np.float32 for k in sd)


def test_final_params_equal_reference_files(golden_dir):
    g = np.load(os.path.join(golden_dir, "matcher.npz"))          # params_* were read from the reference's txt files
    for ds in ("suncg", "matterport", "scannet"):
        assert np.array_equal(np.array(params.final_params(ds)), g[f"params_{ds}"])
    assert params.final_params("suncg") != params.final_params("matterport")


def test_load_param_file(tmp_path):
    p = tmp_path / "final_param_x_rlevel_3.txt"
    rows = params.final_params("scannet")
    p.write_text("\n".join(" ".join(repr(v) for v in r) for r in rows) + "\n")
    assert params.load_param_file(str(p)) == rows


def test_pose_error_metrics_equal_reference_and_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    ad = E.angular_distance_np(g["R_hat"], g["R_gt"])
    assert np.array_equal(ad, g["angular_distance"])                        # the reference's util.angular_distance_np
    assert E.angular_distance_np(g["R_hat"][0], g["R_gt"][0]).shape == (1,)
    rs = np.random.RandomState(5)
    for i in range(len(g["R_hat"])):
        A, B = np.eye(4), np.eye(4)
        A[:3, :3], B[:3, :3] = g["R_hat"][i], g["R_gt"][i]
        A[:3, 3], B[:3, 3] = rs.randn(3), rs.randn(3)
        pc = rs.randn(100, 3)
        assert np.allclose(E.pose_errors(A, B, pc), SO.pose_errors(A, B, pc), rtol=0, atol=1e-12)


def test_result_records_roundtrip(tmp_path):
    rs = np.random.RandomState(1)
    stats = []
    for i in range(5):
        A, B = synth.random_rigid(rs), synth.random_rigid(rs)
        stats.append(E.result_record(f"a/{i}", f"b/{i}", A, B, rs.randn(50, 3), float(rs.rand()), 0.4, 0.3, 0.01))
    keys = {'img_src', 'img_tgt', 'err_ad', 'err_t', 'err_blind', 'err_t_blind', 'overlap', 'pc_dist', 'cam_dist', 'pc_nearest', 'R_gt', 'R_pred_44'}
    assert set(stats[0]) == keys                                             # evaluation.py:303-305
    p = E.save_results(str(tmp_path / "exp.result"), stats)
    assert p.endswith(".result.npy")
    back = E.load_results(p)
    assert len(back) == 5 and all(np.array_equal(a['R_pred_44'], b['R_pred_44']) and a['err_ad'] == b['err_ad'] for a, b in zip(stats, back))
    s = E.summarize(stats)
    assert sum(v["nobs"] for v in s.values()) == 5
    assert E.overlap_bucket(0.05) == '0-0.1' and E.overlap_bucket(0.3) == '0.1-0.5' and E.overlap_bucket(0.9) == '0.5-1.0'


def test_torch_ops_registered_and_cuda_only():
    import pytest
    import torch
    from relativepose_amd import ops
    for name in ops.OPS:
        assert hasattr(torch.ops.relpose, name), name
    with pytest.raises(Exception):                                            # no CPU kernel: loud failure, no fallback
        torch.ops.relpose.pose_inverse(torch.eye(4, dtype=torch.float64)[None])
    from relativepose_amd import rpmodule
    assert ops.params_list(rpmodule.opts())[6] == 0.01 and len(ops.PARAM_ORDER) == 8


def test_torch_ops_have_meta_kernels():
    """Shape functions for every operator (Meta dispatch key): graphs with torch.ops.relpose.* can be traced without a GPU."""
    import torch
    from relativepose_amd import ops  # noqa: F401
    m = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device="meta")
    R = torch.ops.relpose
    f64, i32 = torch.float64, torch.int32
    assert R.pose_inverse(m(3, 4, 4, dt=f64)).shape == (3, 4, 4)
    pc, valid = R.pano2pc(m(2, 160, 640), 0)
    assert pc.shape == (2, 3, 102400) and pc.dtype == f64 and valid.shape == (2, 102400) and valid.dtype == torch.uint8
    assert R.build_view(m(2, 3, 160, 640), m(2, 3, 160, 640), m(2, 160, 640), 0).shape == (2, 8, 160, 640)
    x, mask = R.apply_mask(m(2, 7, 160, 640), 0)
    assert x.shape == (2, 7, 160, 640) and mask.shape == (2, 1, 160, 640)
    assert R.warp(m(2, 8, 160, 640), m(2, 4, 4, dt=f64), 0).shape == (2, 8, 160, 640)
    x16 = m(2, 16, 160, 640)
    assert R.warp_pairs_(x16, m(2, 4, 4, dt=f64), 0) is x16
    pc, nn, ft = R.sample_primitives(m(2, 54, 160, 640), 22, m(2, 3, 160, 640), m(2, 160, 640), m(2, 80, 2, dt=f64), m(2, dt=i32), 0, 0, 0)
    assert pc.shape == nn.shape == (2, 80, 3) and pc.dtype == f64 and ft.shape == (2, 80, 32) and ft.dtype == torch.float32
    kp = (m(4, 80, 3, dt=f64), m(4, 80, 3, dt=f64), m(4, 80, 32), m(4, 80, dt=f64), m(4, 90, 3, dt=f64), m(4, 90, 3, dt=f64), m(4, 90, 32),
          m(4, 90, dt=f64), m(4, dt=i32), m(4, dt=i32))
    pose, status = R.match_pairs(*kp, [0.] * 8, 5, 0, 100000)
    assert pose.shape == (4, 4, 4) and pose.dtype == f64 and status.shape == (4,) and status.dtype == i32
    wij, cj, cw, keff = R.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], [0.] * 8, 5, True)
    assert wij.shape == (4, 80, 90) and cj.shape == cw.shape == (4, 80, 5) and cj.dtype == i32 and cw.dtype == f64 and keff.shape == (4,)
    assert R.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], [0.] * 8, 5, False)[0].numel() == 0
    pts, w, npts = R.keypoints_reference(m(4, 54, 160, 640), 22, m(100, dt=i32), m(100, 2), m(100, dt=i32), m(5, dt=i32), 60, 5, 9,
                                         m(4, 200, dt=i32), m(4, 200, 2, dt=f64), 0)
    assert pts.shape == (4, 200, 2) and w.shape == (4, 200) and npts.shape == (4,)


def test_wc_fixture_generator_is_deterministic_and_geometric():
    from cases import WC_CASES, WC_KW
    d, pts, ptw, T = synth.make_wc_pair(WC_CASES[0], **WC_KW)
    d2, pts2, _, T2 = synth.make_wc_pair(WC_CASES[0], **WC_KW)
    assert np.array_equal(pts, pts2) and np.array_equal(T, T2) and np.array_equal(d["depth"], d2["depth"])
    assert pts.shape == (1, 2, 200, 2) and (ptw == 1).all()
    assert (pts[..., 0] > 160).all() and (pts[..., 0] < 320).all() and (pts[..., 1] > 0).all() and (pts[..., 1] < 160).all()
    assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12)
    sd = weights.make_descriptor_state_dict(21, 15)
    assert not sd["conv4.0.weight"][:, 256:].any() and not sd["deconv4.0.weight"][:256].any()
    assert list(sd) == list(weights.make_state_dict(21, 15))
