"""GPU: batched sigma-tuning objective vs the per-primitive oracle loop, and the tuning step against the reference's
own run of trainRelativePoseModuleRecFD.py:215-298 (tests/golden/tune.npz)."""
import numpy as np
import pytest

from oracle import tune_oracle
from relativepose_amd import synth

pytestmark = pytest.mark.gpu


def _prims(n=6, N=60):
    out = []
    for i in range(n):
        S, T, G = synth.make_match_case(N + 7 * i, 300 + i, inlier=0.5 + 0.05 * i, noise=0.01)
        out.append({'pc_src': S['pc'], 'normal_src': S['normal'], 'feat_src': S['feat'], 'weight_src': S['weight'],
                    'pc_tgt': T['pc'], 'normal_tgt': T['normal'], 'feat_tgt': T['feat'], 'weight_tgt': T['weight'], 'R_gt': G})
    return out


def test_objective_matches_oracle_loop():
    from relativepose_amd import tune
    prims = _prims()
    for sig in ([0.2615, 0.2615, 0.04, 0.01], [0.3, 0.2, 0.03, 0.012]):
        para = tune.make_para(sig)
        lg, ag = tune.objective(prims, para)
        lo, ao = tune_oracle.objective(prims, para)
        assert abs(lg - lo) < 1e-9 * max(1, abs(lo)) and abs(ag - ao) < 1e-6


def test_tune_step_equals_the_reference_script(golden_dir):
    """Two outer iterations of the tuning loop with the batched GPU objective against what the reference's own lines produced on the
    same primitives and random stream: identical probes, loss differences to 1e-10, the accepted sigmas to 1e-9 relative."""
    from relativepose_amd import tune
    from test_tune_cpu import run_against_golden
    run_against_golden(golden_dir, tune.objective)


def test_cache_primitives_feeds_the_objective(golden_dir):
    """tune.cache_primitives (the producer of trainRelativePoseModuleRecFD.py:129-212's cache: the LAST level's primitives of every scan pair
    in the :207-208 dict format) -> tune.objective: the matcher run on the cached primitives with the last level's sigmas reproduces the
    pipeline's own final poses, the dict keys / dtypes / shapes are the reference's, R_gt = R_tgt inv(R_src) (:101), and the objective is the
    mean squared Frobenius error of those poses (VERDICT r4 missing #2)."""
    import torch
    from types import SimpleNamespace
    from cases import WC_CASES, WC_KW, WC_S, WC_SIGMAS, WC_WEIGHT_SEED
    from relativepose_amd import tune, weights
    from relativepose_amd.model import SCNet
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    net = SCNet(SimpleNamespace(batchnorm=1, useTanh=1, skipLayer=1, outputType="rgbdnsf", snumclass=WC_S))
    net.load_state_dict(weights.make_descriptor_state_dict(WC_WEIGHT_SEED, WC_S))
    sig = np.tile(np.array([WC_SIGMAS]), (3, 1))
    pipe = RelativePosePipeline(net, "suncg", "second", sig)
    batches, Ts = [], []
    for seeds in (WC_CASES[:2], WC_CASES[2:]):                   # two batches (2 pairs, 1 pair)
        parts = [synth.make_wc_pair(sd, **WC_KW) for sd in seeds]
        b = {k: np.concatenate([p[0][k] for p in parts]) for k in ("rgb", "norm", "depth", "R")}
        b["pts"], b["ptw"] = np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])
        # extrinsics in the reference's convention (world -> camera: R_gt = R_tgt inv(R_src) is the source -> target motion; the fixture's
        # "R" holds the renderer's camera poses)
        b["R"] = np.stack([np.stack((np.eye(4), p[3])) for p in parts])
        batches.append(b)
        Ts += [p[3] for p in parts]
    prims = tune.cache_primitives(pipe, batches, dev)
    assert len(prims) == 3
    keys = {'pc_src', 'normal_src', 'feat_src', 'weight_src', 'pc_tgt', 'normal_tgt', 'feat_tgt', 'weight_tgt', 'R_gt'}
    for p, b_i in zip(prims, ((0, 0), (0, 1), (1, 0))):
        assert set(p) == keys
        assert p['pc_src'].shape == (200, 3) and p['normal_tgt'].shape == (200, 3) and p['feat_src'].shape == (200, 32)
        assert p['feat_src'].dtype == np.float32 and p['pc_src'].dtype == np.float64 and p['weight_src'].shape == (200,)
        R = batches[b_i[0]]["R"][b_i[1]]
        assert np.array_equal(p['R_gt'], np.matmul(R[1], np.linalg.inv(R[0])))
    # the pipeline's own final poses = the matcher on the cached (last-level) primitives with the last level's sigmas
    finals = []
    for b in batches:
        st = pipe.prepare(b["rgb"], b["norm"], b["depth"], b["pts"], b["ptw"], dev)
        finals.append(pipe.run(st)[0].cpu().numpy())
    finals = np.concatenate(finals)
    ps = tune.PrimitiveSet(prims)
    para = tune.make_para(sig[2])
    got = ps.poses(para)
    assert np.abs(got - finals).max() < 1e-12, np.abs(got - finals).max()
    loss, ad = tune.objective(ps, para)
    want = float(np.mean([np.power(finals[i][:3, :3] - prims[i]['R_gt'][:3, :3], 2).sum() for i in range(3)]))
    assert abs(loss - want) < 1e-12
    # well-conditioned fixtures: the cached primitives' pose is the true motion to matcher accuracy
    from gpu_util import log
    log("cache_primitives", objective_loss=loss, objective_ad=ad, rot_err_vs_true_motion=[float(np.linalg.norm(got[i][:3, :3] - Ts[i][:3, :3])) for i in range(3)])
    assert loss < 1e-3
