"""GPU parity of the whole recurrent loop vs the oracle pipeline and the
reference golden poses (tests/golden/e2e.npz, captured from the reference).

With random-init weights the matching problem is ill-conditioned (roundoff in
the oracle alone moves the pose by 1e-6 per step, see test_oracle_golden), and
the float32 network adds ~1e-4 to the feature maps, so the loop is compared
teacher-forced, stage by stage:
  * network output and sampled primitives within float32 tolerance,
  * the matcher, fed the ORACLE's primitives of that step, within 1e-4,
  * the level's end-to-end pose against the reference's, inside the reference's own one-level perturbation envelope (e2e_env_tf.npz)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from cases import E2E_CASES, E2E_N, E2E_WEIGHT_SEED, ENV_AMP, TF_CASES
from gpu_util import log
from oracle import pipeline_oracle as P
from oracle.scnet_oracle import SCNetOracle
from relativepose_amd import synth, weights

pytestmark = pytest.mark.gpu

ENV_SLACK = 3.0      # as tests/test_gpu_e2e.py: the GPU run is one more sample of a heavy-tailed distribution of which the envelope holds 8


def _gpu_net(S, tanh, seed):
    from relativepose_amd.model import SCNet
    net = SCNet(SimpleNamespace(batchnorm=1, useTanh=tanh, skipLayer=1, outputType="rgbdnsf", snumclass=S))
    net.load_state_dict(weights.make_state_dict(seed, S))
    return net


@pytest.mark.parametrize("ci", TF_CASES)
def test_pipeline_teacher_forced_vs_oracle(ci, golden_dir):
    import torch
    from relativepose_amd import rpmodule
    from relativepose_amd.pipeline import RelativePosePipeline
    ge = np.load(os.path.join(golden_dir, "e2e.npz"))
    gm = np.load(os.path.join(golden_dir, "matcher.npz"))
    env_tf = np.load(os.path.join(golden_dir, "e2e_env_tf.npz"))
    assert float(env_tf["amp"]) == ENV_AMP and ci in TF_CASES
    ds, mm, S, tanh, seed = E2E_CASES[ci]
    d = synth.make_pairs(1, seed, ds)
    pts, ptw = synth.make_keypoints(1, E2E_N, seed, mm)
    sig = gm[f"params_{ds}"]
    forced = [np.eye(4)] + [ge[f"e2e_{ci}_R{s}"] for s in range(2)]
    # oracle, teacher-forced with the reference's poses
    onet = SCNetOracle(weights.make_state_dict(E2E_WEIGHT_SEED, S), S, tanh)
    det = []
    _, otrace = P.run_pair(onet, d["rgb"][0], d["norm"][0], d["depth"][0], pts[0], ptw[0], sig, ds, mm, S, detail=det, R_forced=forced)
    # HIP pipeline
    dev = torch.device("cuda:0")
    pipe = RelativePosePipeline(_gpu_net(S, tanh, E2E_WEIGHT_SEED), ds, mm, sig)
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    keep = []
    Rf = [torch.from_numpy(f[None]).to(dev) for f in forced]
    pose, status, trace = pipe.run(st, R_forced=Rf, keep=keep)
    for step in range(3):
        x_g = keep[step]["x"].cpu().numpy()
        nbad = int((x_g != det[step]["x"]).any(1).sum())
        f_err = np.abs(keep[step]["f"].cpu().numpy() - det[step]["f"]).max()
        prim = det[step]["prim"]
        pc_err = max(np.abs(keep[step]["pc"][0, v].cpu().numpy() - prim[v]["pc"]).max() for v in range(2))
        nn_err = max(np.abs(keep[step]["nn"][0, v].cpu().numpy() - prim[v]["normal"]).max() for v in range(2))
        ft_err = max(np.abs(keep[step]["ft"][0, v].cpu().numpy() - prim[v]["feat"]).max() for v in range(2))
        # matcher on the oracle's primitives of this step (isolates the fit from the fp32 network difference)
        para = rpmodule.opts(*sig[step])
        pose_iso = rpmodule.RelativePoseEstimation_helper(prim[0], prim[1], para)
        iso_err = np.linalg.norm(pose_iso[:3, :3] - otrace[step][:3, :3])
        e2e_err = np.linalg.norm(trace[step][0].cpu().numpy()[:3, :3] - otrace[step][:3, :3])
        ref_err = np.linalg.norm(trace[step][0].cpu().numpy()[:3, :3] - ge[f"e2e_{ci}_R{step}"][:3, :3])
        log("pipeline_step", case=ci, ds=ds, step=step, net_input_pixels_differ=nbad, net_out_abs_err=f_err, pc_err=pc_err,
            normal_err=nn_err, feat_err=ft_err, matcher_iso_rot_err=iso_err, e2e_rot_err_vs_oracle=e2e_err, e2e_rot_err_vs_reference=ref_err)
        assert nbad <= 8
        assert f_err < 1e-3
        assert pc_err < 1e-3 and ft_err < 1e-3
        assert iso_err < 1e-4
        # the level's pose against the REFERENCE's, bounded by the reference's OWN one-level response to +-3e-5 noise on the network output
        # (e2e_env_tf.npz, make_golden.gen_e2e_env_tf: 8 noise seeds per case and level, teacher-forced like this test): the matching problem
        # is ill-conditioned with random weights (module docstring), and the envelope says by how much -- 1e-4 ... 6e-3 depending on case and level
        bound = ENV_SLACK * float(env_tf[f"env_tf_{ci}"][:, step].max())
        log("pipeline_step_envelope", case=ci, step=step, e2e_rot_err_vs_reference=ref_err, reference_envelope_max=float(env_tf[f"env_tf_{ci}"][:, step].max()), bound=bound)
        assert ref_err <= bound and e2e_err <= bound, (ci, step, ref_err, e2e_err, bound)


def test_pipeline_batch_equals_single_and_is_deterministic():
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    ds, mm, S, tanh = "suncg", "second", 15, 1
    net = _gpu_net(S, tanh, E2E_WEIGHT_SEED)
    d = synth.make_pairs(3, 1000, ds)
    pts, ptw = synth.make_keypoints(3, 60, 1000, mm)
    pipe = RelativePosePipeline(net, ds, mm)
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    pose, status, _ = pipe.run(st)
    pose2, _, _ = pipe.run(st)
    assert torch.equal(pose, pose2)
    for b in range(3):
        st1 = pipe.prepare(d["rgb"][b:b + 1], d["norm"][b:b + 1], d["depth"][b:b + 1], pts[b:b + 1], ptw[b:b + 1], dev)
        p1, _, _ = pipe.run(st1)
        assert torch.equal(p1[0], pose[b]), b


@pytest.mark.parametrize("ds,mm", [("suncg", "second"), ("scannet", "kinect")])
def test_pipeline_320x1280_vs_parameterised_oracle(ds, mm):
    """BASELINE config 5 geometry (h=320): the reference asserts 160x640, so the oracle here is the build's own
    parameterised restatement (validated against the reference at h=160 only -- parity unpinned at this size)."""
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    h, S, tanh, N = 320, 15, 1, 48
    d = synth.make_pairs(1, 5100, ds, h=h)
    pts, ptw = synth.make_keypoints(1, N, 5100, mm, h=h)
    sd = weights.make_state_dict(11, S)
    rs = np.random.RandomState(1)
    forced = [np.eye(4), synth.random_rigid(rs, 0.8, 0.5)]
    det = []
    _, otrace = P.run_pair(SCNetOracle(sd, S, tanh), d["rgb"][0], d["norm"][0], d["depth"][0], pts[0], ptw[0], np.tile([[0.26, 0.26, 0.04, 0.01]], (2, 1)),
                           ds, mm, S, alter_steps=2, detail=det, R_forced=forced)
    dev = torch.device("cuda:0")
    from relativepose_amd.model import SCNet
    net = SCNet(SimpleNamespace(batchnorm=1, useTanh=tanh, skipLayer=1, outputType="rgbdnsf", snumclass=S))
    net.load_state_dict(sd)
    pipe = RelativePosePipeline(net, ds, mm, np.tile([[0.26, 0.26, 0.04, 0.01]], (2, 1)), alter_steps=2)
    st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    keep = []
    pipe.run(st, R_forced=[torch.from_numpy(f[None]).to(dev) for f in forced], keep=keep)
    for step in range(2):
        nbad = int((keep[step]["x"].cpu().numpy() != det[step]["x"]).any(1).sum())
        f_err = np.abs(keep[step]["f"].cpu().numpy() - det[step]["f"]).max()
        pc_err = max(np.abs(keep[step]["pc"][0, v].cpu().numpy() - det[step]["prim"][v]["pc"]).max() for v in range(2))
        ft_err = max(np.abs(keep[step]["ft"][0, v].cpu().numpy() - det[step]["prim"][v]["feat"]).max() for v in range(2))
        log("pipeline_320", ds=ds, step=step, net_input_pixels_differ=nbad, net_out_abs_err=f_err, pc_err=pc_err, feat_err=ft_err)
        assert nbad <= 8 and f_err < 1e-3 and pc_err < 1e-3 and ft_err < 1e-3


def test_pipeline_interleaved_streams_equal_single_stream():
    """Two half-batches on two HIP streams (matcher of one overlapping SCNet of the other) give bitwise the same
    poses as the plain single-stream run."""
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    ds, mm, S, tanh = "suncg", "second", 15, 1
    net = _gpu_net(S, tanh, E2E_WEIGHT_SEED)
    d = synth.make_pairs(4, 1200, ds)
    pts, ptw = synth.make_keypoints(4, 60, 1200, mm)
    pipe = RelativePosePipeline(net, ds, mm)
    full = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
    pose, status, _ = pipe.run(full)
    halves = [pipe.prepare(d["rgb"][a:b], d["norm"][a:b], d["depth"][a:b], pts[a:b], ptw[a:b], dev) for a, b in ((0, 2), (2, 4))]
    for _ in range(2):
        res = pipe.run_interleaved(halves)
        torch.cuda.synchronize()
        assert torch.equal(torch.cat([r[0] for r in res]), pose)
        assert torch.equal(torch.cat([r[1] for r in res]), status)


def test_pipeline_batches_in_flight_equal_sequential_runs():
    """run_pipelined (the bench's serving loop: consecutive batches, two in flight, SCNet forwards chained on a dedicated
    stream, matcher of batch k under the forward of batch k+1) returns bitwise what `run` returns batch by batch,
    in order, through on_result as well."""
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    ds, mm, S, tanh = "suncg", "second", 15, 1
    net = _gpu_net(S, tanh, E2E_WEIGHT_SEED)
    pipe = RelativePosePipeline(net, ds, mm)
    states, want = [], []
    for j in range(2):
        d = synth.make_pairs(3, 1300 + 10 * j, ds)
        pts, ptw = synth.make_keypoints(3, 60, 1300 + 10 * j, mm)
        st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
        pose, status, _ = pipe.run(st)
        states.append(st); want.append((pose.clone(), status.clone()))
    torch.cuda.synchronize()
    seen = []
    res = pipe.run_pipelined(states, 5, on_result=lambda k, pose, status: (seen.append(k), (pose.clone(), status.clone()))[1])
    torch.cuda.synchronize()
    assert seen == [0, 1, 2, 3, 4]
    for k, (pose, status) in enumerate(res):
        assert torch.equal(pose, want[k % 2][0]) and torch.equal(status, want[k % 2][1]), k
    res = pipe.run_pipelined(states[:1], 2)                     # depth 1: plain sequential path
    torch.cuda.synchronize()
    assert all(torch.equal(r[0], want[0][0]) for r in res)


def test_pipeline_rotating_states_with_reupload_equal_sequential_runs():
    """The bench's steady state: 4 prepared batches rotate through 2 in-flight slots (a state changes slot stream from step to step) and
    every step re-uploads its inputs from pinned host memory on the slot stream (before_batch = upload_inputs).  The uploaded buffers are
    scrambled first, so a result can only be right if the upload really happened before the batch's first kernel: poses bitwise equal
    to `run` batch by batch, also through the per-run stacking the bench's single gather uses."""
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    ds, mm, S, tanh = "suncg", "second", 15, 1
    net = _gpu_net(S, tanh, E2E_WEIGHT_SEED)
    pipe = RelativePosePipeline(net, ds, mm)
    states, want = [], []
    for j in range(4):
        d = synth.make_pairs(2, 1500 + 10 * j, ds)
        pts, ptw = synth.make_keypoints(2, 60, 1500 + 10 * j, mm)
        st = pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev, keep_host=True)
        pose, status, _ = pipe.run(st)
        states.append(st); want.append((pose.clone(), status.clone()))
    torch.cuda.synchronize()
    for st in states:                                    # whatever is on the device now must not matter
        for k in st["host"]:
            st[k].zero_()
    res = pipe.run_pipelined(states, 7, None, depth=2, before_batch=lambda i, st: pipe.upload_inputs(st, None))
    torch.cuda.synchronize()
    for k, (pose, status) in enumerate(res):
        assert torch.equal(pose, want[k % 4][0]) and torch.equal(status, want[k % 4][1]), k
    stacked = torch.stack([r[0] for r in res], 1).reshape(2 * 7, 4, 4).reshape(2, 7, 4, 4)
    assert torch.equal(stacked[:, -1], want[6 % 4][0])
    # three in flight over the same 4 states (the bench default since round 5): a state changes slot EVERY time it comes round, its re-use is ordered behind
    # the completion event of the batch that last used its buffers (not behind whatever its previous slot stream has queued since)
    for st in states:
        for k in st["host"]:
            st[k].zero_()
    res = pipe.run_pipelined(states, 11, None, depth=3, before_batch=lambda i, st: pipe.upload_inputs(st, None))
    torch.cuda.synchronize()
    for k, (pose, status) in enumerate(res):
        assert torch.equal(pose, want[k % 4][0]) and torch.equal(status, want[k % 4][1]), ("depth 3", k)


def test_pipeline_pose_outputs_option_gives_the_same_poses():
    """RelativePosePipeline(outputs="pose"): SCNet without the rgb / semantic decoder branches -- poses and status bitwise those of the
    default pipeline (which computes every output like the reference), free-running over the three levels."""
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    ds, mm, S, tanh = "suncg", "second", 15, 1
    net = _gpu_net(S, tanh, E2E_WEIGHT_SEED)
    d = synth.make_pairs(3, 1600, ds)
    pts, ptw = synth.make_keypoints(3, 60, 1600, mm)
    full = RelativePosePipeline(net, ds, mm)
    pose, status, trace = full.run(full.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev))
    lean = RelativePosePipeline(net, ds, mm, outputs="pose")
    pose2, status2, trace2 = lean.run(lean.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev))
    assert torch.equal(pose, pose2) and torch.equal(status, status2)
    assert all(torch.equal(a, b) for a, b in zip(trace, trace2))


def test_pipeline_self_stream_cache_gives_the_same_poses():
    """Levels 1-2 reuse level 0's self-view encoder streams (RelativePosePipeline(self_stream_cache=True), the default:
    SCNet.forward(self_tag=...)); with the cache off every level recomputes them like the reference (mymodel.py:266-276 at every
    call of evaluation.py:242).  Poses, status and the pose after every level must be bitwise equal -- in `run`, and in the serving
    loop where four prepared batches rotate through two in-flight slots (a slot's workspace goes from batch to batch)."""
    import torch
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    ds, mm, S, tanh = "suncg", "second", 15, 1
    net = _gpu_net(S, tanh, E2E_WEIGHT_SEED)
    plain = RelativePosePipeline(net, ds, mm, self_stream_cache=False)
    cached = RelativePosePipeline(net, ds, mm)
    assert cached.self_stream_cache
    states, want = [], []
    for j in range(4):
        d = synth.make_pairs(2, 1700 + 10 * j, ds)
        pts, ptw = synth.make_keypoints(2, 60, 1700 + 10 * j, mm)
        st = plain.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev)
        pose, status, trace = plain.run(st)
        pose2, status2, trace2 = cached.run(st)
        assert torch.equal(pose, pose2) and torch.equal(status, status2)
        assert all(torch.equal(a, b) for a, b in zip(trace, trace2))
        states.append(st); want.append((pose.clone(), status.clone()))
    torch.cuda.synchronize()
    res = cached.run_pipelined(states, 9, None, depth=2)
    torch.cuda.synchronize()
    for k, (pose, status) in enumerate(res):
        assert torch.equal(pose, want[k % 4][0]) and torch.equal(status, want[k % 4][1]), k
    res = cached.run_interleaved(states[:3])
    torch.cuda.synchronize()
    for k, (pose, status) in enumerate(res):
        assert torch.equal(pose, want[k][0]) and torch.equal(status, want[k][1]), k


def test_two_pipelines_in_two_threads_with_different_fit_cluster_equal_their_single_thread_runs():
    """the fit's workgroups per pair travel WITH the matcher call (RelposeMatchArgs::fit_cluster), not through the process-wide
    relpose_set_tuning knob round 5 set around the serving loop.  Two pipelines with different settings (1 and 4 workgroups per pair) run their serving
    loops CONCURRENTLY in two threads of one process; each returns bitwise what it returns alone, and the process-wide knob is never touched.
    Reference: the call carries its own `para` (RPModule/rpmodule.py:317-326)."""
    import threading
    import torch
    from relativepose_amd import _lib
    from relativepose_amd.pipeline import RelativePosePipeline
    dev = torch.device("cuda:0")
    ds, mm, S, tanh = "suncg", "second", 15, 1
    pipes, states, want = [], [], []
    for j, cluster in enumerate((1, 4)):
        pipe = RelativePosePipeline(_gpu_net(S, tanh, E2E_WEIGHT_SEED), ds, mm, loop_fit_cluster=cluster)
        sts = []
        for q in range(2):
            d = synth.make_pairs(2, 1700 + 10 * j + q, ds)
            pts, ptw = synth.make_keypoints(2, 120, 1700 + 10 * j + q, mm)      # 600 correspondences per pair: helper workgroups are legal
            sts.append(pipe.prepare(d["rgb"], d["norm"], d["depth"], pts, ptw, dev))
        pipes.append(pipe); states.append(sts)
    for pipe, sts in zip(pipes, states):                                      # alone, one after the other
        res = pipe.run_pipelined(sts, 6)
        torch.cuda.synchronize()
        want.append([(p.clone(), s.clone()) for p, s in res])
    knob0 = _lib.lib().relpose_set_tuning(_lib.TUNE_KEYS["fit_cluster"], 0)
    assert knob0 == 0
    got, errs = [None, None], []

    def work(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                r = pipes[i].run_pipelined(states[i], 6)
                torch.cuda.current_stream().synchronize()
            got[i] = r
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    for rep in range(2):
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        assert not errs, errs
        for i in range(2):
            for k, (pose, status) in enumerate(got[i]):
                assert torch.equal(pose, want[i][k][0]) and torch.equal(status, want[i][k][1]), (rep, i, k)
    assert _lib.lib().relpose_set_tuning(_lib.TUNE_KEYS["fit_cluster"], 0) == 0      # nobody set the process-wide knob
    log("pipeline_two_threads_fit_cluster", bitwise=True)
