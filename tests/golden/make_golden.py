"""Generate the golden fixtures in tests/golden/*.npz by running the UPSTREAM
REFERENCE (imported in place from /root/reference, CPU) on seeded synthetic
inputs.  Runs only in the build container; the fixtures (data: inputs are
regenerated from seeds, expected outputs are stored) travel with the repo.

    python tests/golden/make_golden.py            # all groups
    python tests/golden/make_golden.py matcher    # one group

Groups: matcher, matcher_stages, matcher_stages_big512, tune, matcher_big, geometry, scnet, e2e, e2e_env, e2e_env_tf, e2e_wc, e2e_wc2, stats, keypoints, getkeypoint, gmp_nc, metrics.  See SURVEY.md §8c for the plan.
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_loader  # noqa: E402
from relativepose_amd import synth, weights  # noqa: E402

PARAM_ROWS = {
    # data/relativePoseModule/final_param_*_rlevel_3.txt read at run time from the reference
}


def load_params():
    out = {}
    for ds in ("suncg", "matterport", "scannet"):
        out[ds] = np.loadtxt(os.path.join(ref_loader.REF, "data", "relativePoseModule",
                                          f"final_param_{ds}_rlevel_3.txt")).reshape(-1, 4)
    return out


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sample_idx(n, k, seed):
    return np.random.RandomState(seed).randint(0, n, size=k)


from cases import (MATCH_CASES, MATCH_METHODS, GEOM_CASES, WARP_ANGLES, SCNET_CASES, SCNET_VARIANT_CASES, E2E_CASES, E2E_N, E2E_WEIGHT_SEED,  # noqa: E402
                   ENV_AMP, ENV_SEEDS, WC_CASES, WC_KW, WC_S, WC_SIGMAS, WC_WEIGHT_SEED, WC2_CASES, TUNE_CASE)


def gen_matcher():
    R = ref_loader.load()
    rp, ru = R["rpmodule"], R["rputil"]
    params = load_params()
    out = {}
    for ci, (N, Nt, seed, ds, row, inl) in enumerate(MATCH_CASES):
        S, T, _ = synth.make_match_case(N, seed, inlier=inl, Nt=Nt)
        for method in MATCH_METHODS:
            if method != "irls+sm" and N > 200:
                continue
            para = ru.opts(*params[ds][row])
            para.method = method
            t = time.time()
            pose = rp.RelativePoseEstimation_helper(S, T, para)
            print(f"matcher case {ci} N={N}/{Nt} {method}: {time.time()-t:.2f}s")
            out[f"pose_{ci}_{method}"] = pose
    out["params_suncg"], out["params_matterport"], out["params_scannet"] = (params[k] for k in ("suncg", "matterport", "scannet"))
    np.savez_compressed(os.path.join(HERE, "matcher.npz"), **out)


class _HelperSpy:
    """Captures the intermediates of ONE reference run of RelativePoseEstimation_helper (rpmodule.py:317-508) from the reference run
    itself: the helper's locals when it returns (sys.settrace 'return' event: wij, corres, the surviving pairs and their weights), the
    filter counts from the reference's own log lines (:404, :436), and the pose (R_cur, t_cur) of fit_irls_sm every time the fit
    rebinds it (:270-271 after the initial IRLS, :306-307 after each of the 5 alternations)."""

    def __init__(self, rp):
        self.rp, self.locals, self.trace, self.log = rp, None, [], []
        self._t_id = None

    def info(self, msg, *a, **k):
        self.log.append(str(msg))

    def __getattr__(self, name):            # any other logger method
        return lambda *a, **k: None

    def _tracer(self, frame, event, arg):
        name = frame.f_code.co_name
        if event != "call" or name not in ("RelativePoseEstimation_helper", "fit_irls_sm"):
            return None
        if frame.f_code.co_filename != self.rp.__file__:
            return None

        def local(fr, ev, ar):
            if name == "fit_irls_sm" and ev in ("line", "return"):
                t = fr.f_locals.get("t_cur")
                if t is not None and id(t) != self._t_id:          # a new t_cur: R_cur was assigned on the line before
                    self._t_id = id(t)
                    T = np.eye(4)
                    T[:3, :3], T[:3, 3] = np.asarray(fr.f_locals["R_cur"]).reshape(3, 3), np.asarray(t).reshape(3)
                    self.trace.append(T)
            if name == "RelativePoseEstimation_helper" and ev == "return":
                self.locals = dict(fr.f_locals)
            return local
        return local

    def run(self, S, T, para):
        old_logger = self.rp.logger
        self.rp.logger = self
        sys.settrace(self._tracer)
        try:
            pose = self.rp.RelativePoseEstimation_helper(S, T, para)
        finally:
            sys.settrace(None)
            self.rp.logger = old_logger
        return pose


def stage_record(spy, pose, N, Nt):
    """The stage-level quantities SURVEY 8(c) lists, from a _HelperSpy run: SHA-256 + 64 samples of wij (float64), the per-row
    correspondence sets over wij > 0 (sorted, -1 padded), the filter counts, M, sum(w) and the pose after each alternation."""
    L = spy.locals
    rec = {"pose": pose}
    if L is None or "wij" not in L:
        return rec
    wij = np.asarray(L["wij"])
    rec["wij_sha"] = np.array(digest(wij))
    idx = sample_idx(wij.size, 64, 7)
    rec["wij_idx"], rec["wij_val"] = idx, wij.reshape(-1)[idx]
    rec["wij_rowsum"] = wij.sum(1)
    if "corres" in L:
        corres = np.asarray(L["corres"])
        K = corres.shape[1] // N
        cj = corres[1].reshape(N, K)
        sets = np.full((N, K), -1, dtype=np.int64)
        for i in range(N):
            keep = sorted(int(j) for j in cj[i] if wij[i, j] > 0)
            sets[i, :len(keep)] = keep
        rec["corres_sets"] = sets
        rec["wij_kth"] = np.sort(wij, 1)[:, ::-1][:, K - 1:K + 1]          # K-th and (K+1)-th largest per row: exact ties are visible
    C = N * min(5, Nt - 1)
    n_pairs = C * (C - 1) // 2
    for msg in spy.log:
        if msg.startswith("dist delete:"):
            rec["n_dist"] = np.array(n_pairs - int(msg.split(":")[1]))
        if msg.startswith("angle delete:"):
            rec["n_angle"] = np.array(int(rec["n_dist"]) - int(msg.split(":")[1]))
    if "w_i1i2j1j2" in L:
        w = np.asarray(L["w_i1i2j1j2"])
        rec["M"], rec["w_sum"], rec["w_nonzero"] = np.array(len(w)), np.array(w.sum()), np.array(int((w != 0).sum()))
    if spy.trace:
        rec["trace"] = np.stack(spy.trace)
    return rec


def gen_matcher_stages():
    """matcher_stages.npz: the intermediates of the reference helper for every MATCH_CASES entry and MATCH_BIG, captured from the
    reference run itself (no restatement involved): see _HelperSpy / stage_record."""
    from cases import MATCH_BIG
    R = ref_loader.load()
    rp, ru = R["rpmodule"], R["rputil"]
    params = load_params()
    out = {}
    cases = [(f"{ci}", c + (0.005,)) for ci, c in enumerate(MATCH_CASES)] + [("big", MATCH_BIG)]
    for tag, (N, Nt, seed, ds, row, inl, noise) in cases:
        S, T, _ = synth.make_match_case(N, seed, inlier=inl, noise=noise, Nt=Nt)
        para = ru.opts(*params[ds][row])
        spy = _HelperSpy(rp)
        t = time.time()
        pose = spy.run(S, T, para)
        rec = stage_record(spy, pose, N, Nt)
        print(f"matcher_stages {tag}: N={N}/{Nt} {time.time()-t:.1f}s keys={sorted(rec)} trace={len(spy.trace)}")
        for k, v in rec.items():
            out[f"{tag}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "matcher_stages.npz"), **out)


def gen_matcher_stages_big512():
    """matcher_stages_big512.npz: the stage intermediates of the reference helper (as gen_matcher_stages) for cases.MATCH_BIG512 -- 1000 source
    x 512 target keypoints, the largest shape the affinity tile / pool kernels accept, so that every affinity kernel runs a 5000-correspondence
    stage golden (the N = Nt = 1000 case can only run on the row / LDS kernels).  A file of its own: the other cases are not regenerated."""
    from cases import MATCH_BIG512
    R = ref_loader.load()
    rp, ru = R["rpmodule"], R["rputil"]
    params = load_params()
    N, Nt, seed, ds, row, inl, noise = MATCH_BIG512
    S, T, _ = synth.make_match_case(N, seed, inlier=inl, noise=noise, Nt=Nt)
    spy = _HelperSpy(rp)
    t = time.time()
    pose = spy.run(S, T, ru.opts(*params[ds][row]))
    rec = stage_record(spy, pose, N, Nt)
    print(f"matcher_stages big512: N={N}/{Nt} {time.time()-t:.1f}s keys={sorted(rec)} trace={len(spy.trace)}")
    np.savez_compressed(os.path.join(HERE, "matcher_stages_big512.npz"), **{f"big512_{k}": v for k, v in rec.items()})


def gen_tune():
    """tune.npz: the finite-difference sigma tuning of the reference (trainRelativePoseModuleRecFD.py:215-298: objective(), probe
    perturbations, least-squares gradient, normalised step, halving line search) run BY THE REFERENCE'S OWN LINES: the text of
    lines 215-298 is read from /root/reference at generation time and exec'd (nothing of it is stored) in a namespace that provides
    what the script's earlier lines would have: the primitives (seeded synthetic ones instead of the cached .npy), the reference's
    helper / opts / angular_distance_np, a seeded np.random and args.max_iter.  Stored: per outer iteration the probe perturbations
    and loss differences (as handed to np.linalg.lstsq), and the line the script appends to its log: loss, angular distance, sigmas."""
    import contextlib
    import io
    import tempfile
    import types
    R = ref_loader.load()
    src = open(os.path.join(ref_loader.REF, "trainRelativePoseModuleRecFD.py")).read().split("\n")
    body = "\n".join(src[214:298])                      # lines 215-298
    assert body.lstrip().startswith("def objective(para):") and "f.write(" in src[297]
    records = []
    real_lstsq = np.linalg.lstsq

    def spy_lstsq(a, b, *args, **kw):
        records.append((np.array(a, copy=True), np.array(b, copy=True)))
        return real_lstsq(a, b, *args, **kw)

    tmp = tempfile.mkdtemp()
    ns = {"np": np, "primitives": synth.make_tune_primitives(TUNE_CASE["n_prims"], TUNE_CASE["N"], TUNE_CASE["seed0"]), "RelativePoseEstimation_helper": R["rpmodule"].RelativePoseEstimation_helper,
          "opts": R["rputil"].opts, "angular_distance_np": R["util"].angular_distance_np,
          "args": types.SimpleNamespace(max_iter=TUNE_CASE["iters"], exp=os.path.join(tmp, "tune"))}
    np.random.seed(TUNE_CASE["np_seed"])
    np.linalg.lstsq = spy_lstsq
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            exec(compile(body, "trainRelativePoseModuleRecFD.py[215:298]", "exec"), ns)
    finally:
        np.linalg.lstsq = real_lstsq
    lines = np.loadtxt(os.path.join(tmp, "tune.txt")).reshape(-1, 6)      # loss_best ad_best sigmaAngle1 sigmaAngle2 sigmaDist sigmaFeat
    assert len(records) == len(lines) == TUNE_CASE["iters"]
    out = {"log": lines, "eps": np.stack([r[0] for r in records]), "dloss": np.stack([r[1] for r in records]),
           "sigma_init": np.array([ns["sigmaAngle1_init"], ns["sigmaAngle2_init"], ns["sigmaDist_init"], ns["sigmaFeat_init"]]),
           "found_last": np.array(bool(ns["foundDescentDirection"]))}
    print("tune log:\n", lines)
    np.savez_compressed(os.path.join(HERE, "tune.npz"), **out)


def gen_matcher_big():
    """One pair with N = 1000 keypoints per view (5000 correspondences, 12.5 M candidate pairs -- beyond what the fit keeps in LDS):
    the reference helper's pose (cases.MATCH_BIG)."""
    from cases import MATCH_BIG
    R = ref_loader.load()
    rp, ru = R["rpmodule"], R["rputil"]
    params = load_params()
    N, Nt, seed, ds, row, inl, noise = MATCH_BIG
    S, T, G = synth.make_match_case(N, seed, inlier=inl, noise=noise, Nt=Nt)
    para = ru.opts(*params[ds][row])
    t = time.time()
    pose = rp.RelativePoseEstimation_helper(S, T, para)
    print(f"matcher_big N={N}: {time.time()-t:.1f}s, rot err vs ground truth {np.linalg.norm(pose[:3,:3]-G[:3,:3]):.2e}")
    np.savez_compressed(os.path.join(HERE, "matcher_big.npz"), pose=pose, params=params[ds][row])


def _views(ds, seed, method):
    import torch
    R = ref_loader.load()
    util = R["util"]
    d = synth.make_pairs(1, seed, ds)
    comp = torch.from_numpy(np.concatenate((d["rgb"][0, 0], d["norm"][0, 0], d["depth"][0, 0][None]), 0)[None])
    v, m, _ = util.apply_mask(comp.clone(), method)
    v = torch.cat((v, (v[:, 6:7] != 0).float()), 1)
    return d, v, m


def gen_geometry():
    import torch
    R = ref_loader.load()
    util, ru = R["util"], R["rputil"]
    out = {}
    for ds, mm, seed in GEOM_CASES:
        d, v, m = _views(ds, seed, mm)
        out[f"mask_{ds}"] = np.packbits(m.numpy().astype(np.uint8))
        out[f"view_{ds}_sha"] = digest(v.numpy())
        pc = util.Pano2PointCloud(d["depth"][0, 0], ds)
        idx = sample_idx(pc.shape[1], 2048, 1)
        out[f"pano2pc_{ds}_shape"] = np.array(pc.shape)
        out[f"pano2pc_{ds}_idx"], out[f"pano2pc_{ds}_val"] = idx, pc[:, idx]
        out[f"pano2pc_{ds}_sum"] = pc.sum(1)
        rs = np.random.RandomState(seed + 1)
        for k in range(3):
            T = synth.random_rigid(rs, WARP_ANGLES[k], 0.8)
            w = np.asarray(util.warping(v.numpy(), T, ds))
            w32 = w.astype(np.float32)
            idx = sample_idx(w32.size, 8192, 2 + k)
            out[f"warp_{ds}_{k}_T"] = T
            out[f"warp_{ds}_{k}_idx"], out[f"warp_{ds}_{k}_val"] = idx, w.reshape(-1)[idx]
            out[f"warp_{ds}_{k}_chsum"] = w.sum((0, 2, 3))
            out[f"warp_{ds}_{k}_nnz"] = np.array([(w[0, c] != 0).sum() for c in range(8)])
            out[f"warp_{ds}_{k}_maskbits"] = np.packbits((w[0, 7] != 0).astype(np.uint8))
            out[f"warp_{ds}_{k}_sha32"] = digest(w32)
        wI = util.warping(v.numpy(), np.eye(4), ds)
        out[f"warp_{ds}_I_absmax"] = np.array(float(np.abs(np.asarray(wI)).max()))
        # keypoint sampling on a "completed" map = the full synthetic maps
        pts, _ = synth.make_keypoints(1, 64, seed + 5, mm)
        pts = pts[0, 0]
        depth = d["depth"][0, 0]
        normal = d["norm"][0, 0].transpose(1, 2, 0)
        pc, nn = ru.getPixel(depth, normal, pts, dataset=ds)
        out[f"getpixel_{ds}_pc"], out[f"getpixel_{ds}_nn"] = pc, nn
        feat = np.random.RandomState(seed + 6).randn(32, 160, 640).astype(np.float32)
        ptn = pts.copy()
        ptn[:, 0] /= 640
        ptn[:, 1] /= 160
        out[f"interp_{ds}"] = ru.interpolate(torch.from_numpy(feat), torch.from_numpy(ptn).float()).numpy()
    # depth2pc on the observed face / crop
    for ds in ("suncg", "matterport", "scannet"):
        d = synth.make_pairs(1, 400, ds)
        dep = d["depth"][0, 0]
        crop = dep[47:113, 196:284] if ds == "scannet" else dep[:, 160:320]
        pc, mask = util.depth2pc(crop, ds)
        out[f"depth2pc_{ds}_sum"], out[f"depth2pc_{ds}_n"] = pc.sum(0), np.array(pc.shape[0])
        out[f"depth2pc_{ds}_head"] = pc[:256]
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), **out)


class _Args:
    batchnorm, skipLayer, outputType = 1, 1, "rgbdnsf"

    def __init__(self, S, tanh):
        self.snumclass, self.useTanh = S, tanh


def ref_net(S, tanh, seed, sd=None):
    import torch
    R = ref_loader.load()
    net = R["mymodel"].SCNet(_Args(S, tanh))
    sd = {k: torch.from_numpy(v) for k, v in (sd if sd is not None else weights.make_state_dict(seed, S)).items()}
    net.load_state_dict(sd, strict=True)
    return net


def scnet_input(seed, ds="suncg", mm="second", step_R=None):
    """A realistic [2,16,160,640] net input: masked views + warped other view."""
    R = ref_loader.load()
    util = R["util"]
    import torch
    d = synth.make_pairs(1, seed, ds)
    vs = []
    for v in range(2):
        comp = torch.from_numpy(np.concatenate((d["rgb"][0, v], d["norm"][0, v], d["depth"][0, v][None]), 0)[None])
        x, m, _ = util.apply_mask(comp.clone(), mm)
        vs.append(torch.cat((x, (x[:, 6:7] != 0).float()), 1))
    Rg = d["R"][0]
    T = np.linalg.inv(Rg[1]) @ Rg[0] if step_R is None else step_R
    t2s = torch.from_numpy(np.asarray(util.warping(vs[1].numpy(), np.linalg.inv(T), ds))).float()
    s2t = torch.from_numpy(np.asarray(util.warping(vs[0].numpy(), T, ds))).float()
    return torch.cat((torch.cat((vs[0], t2s), 1), torch.cat((vs[1], s2t), 1))), d


def gen_scnet():
    import torch
    out = {}
    for tag, S, tanh, seed, ds, mm in SCNET_CASES:
        net = ref_net(S, tanh, seed)
        x, _ = scnet_input(500 + seed, ds, mm)
        taps = {}
        hooks = []
        for name, mod in net.named_modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                hooks.append(mod.register_forward_hook(
                    lambda m, i, o, name=name: taps.setdefault(name, []).append(o.detach())))
        with torch.no_grad():
            t = time.time()
            y = net(x)
            print(f"scnet {tag}: {time.time()-t:.2f}s out {tuple(y.shape)}")
        for h in hooks:
            h.remove()
        out[f"{tag}_cfg"] = np.array([S, tanh, seed, 500 + seed])
        out[f"{tag}_ds"] = np.array([ds, mm])
        out[f"{tag}_out_crop"] = y[:, :, 40:72, 300:332].numpy()
        out[f"{tag}_out_chmean"] = y.mean((2, 3)).numpy()
        out[f"{tag}_out_chabsmax"] = y.abs().amax((2, 3)).numpy()
        idx = sample_idx(y.numel(), 20000, 9)
        out[f"{tag}_out_idx"], out[f"{tag}_out_val"] = idx, y.reshape(-1)[idx].numpy()
        for name, lst in taps.items():
            # first call of a shared-weight block = self stream, second = t2s stream
            for ci, o in enumerate(lst):
                out[f"{tag}_tap_{name}_{ci}_mean"] = o.mean((0, 2, 3)).numpy()
                out[f"{tag}_tap_{name}_{ci}_std"] = o.std((0, 2, 3)).numpy()
    np.savez_compressed(os.path.join(HERE, "scnet.npz"), **out)


def gen_scnet_variants():
    """The reference SCNet built with the other constructor switches (mymodel.py:145-149), fed the seeded state dict of the same spec:
    output samples + per-layer statistics, like gen_scnet.  Also records that the two combinations the product rejects fail in the
    reference too ('k' head; skipLayer=0 with an rgb / n / d head)."""
    import torch
    R = ref_loader.load()
    out = {}
    for tag, S, tanh, seed, ds, mm, bn, skip, otype in SCNET_VARIANT_CASES:
        a = _Args(S, tanh)
        a.batchnorm, a.skipLayer, a.outputType = bn, skip, otype
        net = R["mymodel"].SCNet(a)
        sd = weights.make_state_dict(seed, S, bn, skip, otype)
        assert list(sd) == list(net.state_dict()), "seeded spec != the reference's registration order"
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        x, _ = scnet_input(500 + seed, ds, mm)
        taps, hooks = {}, []
        for name, mod in net.named_modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: taps.setdefault(name, []).append(o.detach())))
        with torch.no_grad():
            y = net(x)
        for h in hooks:
            h.remove()
        print(f"scnet variant {tag}: out {tuple(y.shape)} absmax {float(y.abs().max()):.3g}")
        out[f"{tag}_out_crop"] = y[:, :, 40:72, 300:332].numpy()
        out[f"{tag}_out_chmean"] = y.mean((2, 3)).numpy()
        out[f"{tag}_out_chabsmax"] = y.abs().amax((2, 3)).numpy()
        idx = sample_idx(y.numel(), 20000, 9)
        out[f"{tag}_out_idx"], out[f"{tag}_out_val"] = idx, y.reshape(-1)[idx].numpy()
        for name, lst in taps.items():
            for ci, o in enumerate(lst):
                out[f"{tag}_tap_{name}_{ci}_mean"] = o.mean((0, 2, 3)).numpy()
                out[f"{tag}_tap_{name}_{ci}_std"] = o.std((0, 2, 3)).numpy()
    # what the reference cannot run (the product raises ValueError for these instead of building them)
    fails = []
    for bn, skip, otype in ((1, 0, "rgbdnsf"), (1, 1, "rgbdnksf")):
        a = _Args(15, 1)
        a.batchnorm, a.skipLayer, a.outputType = bn, skip, otype
        try:
            with torch.no_grad():
                R["mymodel"].SCNet(a)(scnet_input(531)[0])
            fails.append("ran")
        except Exception as e:
            fails.append(type(e).__name__)
    print("reference on the rejected combinations:", fails)
    out["rejected_fail_types"] = np.array(fails)
    np.savez_compressed(os.path.join(HERE, "scnet_variants.npz"), **out)


def ref_loop(net, d, pts, ptw, ds, mm, S, sigmas, noise_amp=0.0, noise_seed=0, keep_prims=None, R_forced=None):
    """evaluation.py:217-284 for ONE scan pair driven through the reference's own functions (util.apply_mask,
    util.warping, the reference SCNet, rpmodule.getMatchingPrimitive with getKeypoint replaced by the injected
    keypoints, rpmodule.RelativePoseEstimation_helper).  Returns [R_hat after each of the 3 steps].
    noise_amp > 0 adds uniform(-amp, amp) float32 noise to every network output (perturbation envelope).
    R_forced: the pose estimate every level STARTS from (teacher forcing: level k warps with R_forced[k] instead of its own previous estimate)."""
    import torch
    R = ref_loader.load()
    util, rp, ru, top = R["util"], R["rpmodule"], R["rputil"], R["torch_op"]
    rs_noise = np.random.RandomState(noise_seed)

    def fake_kp(rs, rt, fs, ft, *a, **k):
        ps, pt = pts[0, 0], pts[0, 1]
        pn, tn = ps.copy(), pt.copy()
        pn[:, 0] /= 640; pn[:, 1] /= 160; tn[:, 0] /= 640; tn[:, 1] /= 160
        return ps, pn, ptw[0, 0], pt, tn, ptw[0, 1]

    rp.getKeypoint = fake_kp
    rp.getKeypoint_kinect = fake_kp
    data = {k: torch.from_numpy(v) for k, v in d.items()}
    rgb_u8 = (d["rgb"] * 255).clip(0, 255).astype("uint8")
    trace = []
    with torch.no_grad():
        R_hat = np.eye(4)
        comp = [torch.cat((top.v(data["rgb"][:, v]), top.v(data["norm"][:, v]), top.v(data["depth"][:, v:v + 1])), 1)
                for v in range(2)]
        views, masks = [], []
        for v in range(2):
            x, m, _ = util.apply_mask(comp[v].clone(), mm)
            masks.append(top.npy(m[0]).transpose(1, 2, 0))
            views.append(torch.cat((x, (x[:, 6:7] != 0).float()), 1))
        obs = [{"rgb": rgb_u8[0, v].transpose(1, 2, 0), "depth": d["depth"][0, v],
                "normal": d["norm"][0, v].transpose(1, 2, 0)} for v in range(2)]
        fs, fe = 7 + S, 7 + S + 32
        for step in range(3):
            if R_forced is not None:
                R_hat = R_forced[step]
            t2s = top.v(util.warping(top.npy(views[1]), np.linalg.inv(R_hat), ds))
            s2t = top.v(util.warping(top.npy(views[0]), R_hat, ds))
            f = net(torch.cat((torch.cat((views[0], t2s), 1), torch.cat((views[1], s2t), 1))))
            if noise_amp:
                f = f + torch.from_numpy(rs_noise.uniform(-noise_amp, noise_amp, tuple(f.shape)).astype(np.float32))
            dc = []
            for v in range(2):
                fv = top.npy(f[v])
                m = masks[v]
                c = {}
                c["normal"] = ((1 - m) * fv[3:6].transpose(1, 2, 0) + m * obs[v]["normal"]) \
                    / (np.linalg.norm(obs[v]["normal"], axis=2, keepdims=True) + 1e-6)
                c["depth"] = (1 - m[:, :, 0]) * fv[6] + m[:, :, 0] * obs[v]["depth"]
                c["rgb"] = (m * obs[v]["rgb"]).astype("uint8")
                c["rgb_full"] = c["rgb"]
                c["feat"] = f[v, fs:fe]
                dc.append(c)
            para = ru.opts(*sigmas[step])
            prim = rp.getMatchingPrimitive(dc[0], dc[1], ds, "skybox", 1)
            p3s, p3t, ns_, nt_, des, det, ws, wt = prim
            R_hat = rp.RelativePoseEstimation_helper({"pc": p3s.T, "normal": ns_, "feat": des, "weight": ws},
                                                     {"pc": p3t.T, "normal": nt_, "feat": det, "weight": wt}, para)
            trace.append(R_hat)
            if step == 0 and keep_prims is not None:
                keep_prims.update(pc=p3s, n=ns_, des=des, pct=p3t, nt=nt_, dest=det)
    return trace


def gen_e2e():
    """evaluation.py:217-284 driven through the reference's own functions with
    rputil.getKeypoint replaced by injected keypoints (4 SUNCG-shaped pairs,
    config 1 of BASELINE.json, + 1 matterport + 1 scannet pair)."""
    params = load_params()
    out = {}
    nets = {}
    for ci, (ds, mm, S, tanh, seed) in enumerate(E2E_CASES):
        key = (S, tanh)
        if key not in nets:
            nets[key] = ref_net(S, tanh, E2E_WEIGHT_SEED)
        d = synth.make_pairs(1, seed, ds)
        pts, ptw = synth.make_keypoints(1, E2E_N, seed, mm)
        t0 = time.time()
        prims = {}
        trace = ref_loop(nets[key], d, pts, ptw, ds, mm, S, params[ds], keep_prims=prims)
        for step in range(3):
            out[f"e2e_{ci}_R{step}"] = trace[step]
        out[f"e2e_{ci}_prim_pc"], out[f"e2e_{ci}_prim_n"], out[f"e2e_{ci}_prim_des"] = prims["pc"], prims["n"], prims["des"]
        out[f"e2e_{ci}_prim_pct"], out[f"e2e_{ci}_prim_nt"], out[f"e2e_{ci}_prim_dest"] = prims["pct"], prims["nt"], prims["dest"]
        print(f"e2e case {ci} {ds}: {time.time()-t0:.1f}s")
        out[f"e2e_{ci}_cfg"] = np.array([ds, mm, str(S), str(tanh), str(seed), str(E2E_N)])
    out["n_cases"] = np.array(len(E2E_CASES))
    np.savez_compressed(os.path.join(HERE, "e2e.npz"), **out)


def gen_e2e_env():
    """Perturbation envelope of the REFERENCE loop on the e2e cases (random-init weights: ill-conditioned): the loop is
    re-run ENV_SEEDS times with uniform(-ENV_AMP, ENV_AMP) noise on every network output -- ENV_AMP = the size of the
    float32 kernel-vs-reference difference of the network output -- and the rotation difference to the unperturbed
    reference poses is stored per step.  The free-running GPU test asserts its own difference against this envelope."""
    params = load_params()
    ge = np.load(os.path.join(HERE, "e2e.npz"))
    out = {"amp": np.array(ENV_AMP), "n_seeds": np.array(ENV_SEEDS)}
    nets = {}
    for ci, (ds, mm, S, tanh, seed) in enumerate(E2E_CASES):
        key = (S, tanh)
        if key not in nets:
            nets[key] = ref_net(S, tanh, E2E_WEIGHT_SEED)
        d = synth.make_pairs(1, seed, ds)
        pts, ptw = synth.make_keypoints(1, E2E_N, seed, mm)
        env = np.zeros((ENV_SEEDS, 3))
        t0 = time.time()
        for k in range(ENV_SEEDS):
            tr = ref_loop(nets[key], d, pts, ptw, ds, mm, S, params[ds], noise_amp=ENV_AMP, noise_seed=7000 + 100 * ci + k)
            env[k] = [np.linalg.norm(tr[s][:3, :3] - ge[f"e2e_{ci}_R{s}"][:3, :3]) for s in range(3)]
        out[f"env_{ci}"] = env
        print(f"e2e_env case {ci} {ds}: max per step {env.max(0)}  median {np.median(env, 0)}  ({time.time()-t0:.1f}s)")
    np.savez_compressed(os.path.join(HERE, "e2e_env.npz"), **out)


def gen_e2e_env_tf():
    """gen_e2e_env for the TEACHER-FORCED loop (tests/test_gpu_pipeline.py::test_pipeline_teacher_forced_vs_oracle: level k starts from the
    reference's own pose after level k - 1): the reference's one-level response to uniform(-ENV_AMP, ENV_AMP) noise on the network output, per
    level, for the cases that test runs.  Free-running envelopes are O(1) after level 0 (the loop is chaotic with random weights); teacher-forced,
    every level is ONE step from a fixed pose and its envelope bounds what a float32 kernel difference can do to that level's pose."""
    from cases import TF_CASES
    params = load_params()
    ge = np.load(os.path.join(HERE, "e2e.npz"))
    out = {"amp": np.array(ENV_AMP), "n_seeds": np.array(ENV_SEEDS)}
    nets = {}
    for ci in TF_CASES:
        ds, mm, S, tanh, seed = E2E_CASES[ci]
        key = (S, tanh)
        if key not in nets:
            nets[key] = ref_net(S, tanh, E2E_WEIGHT_SEED)
        d = synth.make_pairs(1, seed, ds)
        pts, ptw = synth.make_keypoints(1, E2E_N, seed, mm)
        forced = [np.eye(4)] + [ge[f"e2e_{ci}_R{s}"] for s in range(2)]
        base = ref_loop(nets[key], d, pts, ptw, ds, mm, S, params[ds], R_forced=forced)
        drift = [float(np.linalg.norm(base[s][:3, :3] - ge[f"e2e_{ci}_R{s}"][:3, :3])) for s in range(3)]
        assert max(drift) < 1e-9, drift           # forcing the reference with its own poses reproduces its free-running trajectory
        env = np.zeros((ENV_SEEDS, 3))
        t0 = time.time()
        for k in range(ENV_SEEDS):
            tr = ref_loop(nets[key], d, pts, ptw, ds, mm, S, params[ds], noise_amp=ENV_AMP, noise_seed=9000 + 100 * ci + k, R_forced=forced)
            env[k] = [np.linalg.norm(tr[s][:3, :3] - ge[f"e2e_{ci}_R{s}"][:3, :3]) for s in range(3)]
        out[f"env_tf_{ci}"] = env
        print(f"e2e_env_tf case {ci} {ds}: max per level {env.max(0)}  median {np.median(env, 0)}  ({time.time()-t0:.1f}s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "e2e_env_tf.npz"), **out)


def gen_e2e_wc():
    """Well-conditioned end-to-end fixtures (cases.WC_CASES): synth.make_wc_pair scan pairs (keypoints = projections of
    common world points) + weights.make_descriptor_state_dict (descriptors follow the view-invariant texture).  Stores
    the reference poses after each step and the same perturbation envelope as gen_e2e_env (expected << 1e-5)."""
    out = {"amp": np.array(ENV_AMP), "n_seeds": np.array(ENV_SEEDS)}
    net = ref_net(WC_S, 1, None, sd=weights.make_descriptor_state_dict(WC_WEIGHT_SEED, WC_S))
    sig = np.tile(np.array([WC_SIGMAS]), (3, 1))
    for ci, seed in enumerate(WC_CASES):
        d, pts, ptw, T = synth.make_wc_pair(seed, **WC_KW)
        t0 = time.time()
        trace = ref_loop(net, d, pts, ptw, "suncg", "second", WC_S, sig)
        env = np.zeros((ENV_SEEDS, 3))
        for k in range(ENV_SEEDS):
            tr = ref_loop(net, d, pts, ptw, "suncg", "second", WC_S, sig, noise_amp=ENV_AMP, noise_seed=8000 + 100 * ci + k)
            env[k] = [np.linalg.norm(tr[s][:3, :3] - trace[s][:3, :3]) for s in range(3)]
        for s in range(3):
            out[f"wc_{ci}_R{s}"] = trace[s]
        out[f"wc_{ci}_T"] = T
        out[f"wc_env_{ci}"] = env
        print(f"e2e_wc case {ci} seed {seed}: rot err vs true motion {[float(np.linalg.norm(trace[s][:3,:3]-T[:3,:3])) for s in range(3)]}"
              f"  envelope max {env.max(0)}  ({time.time()-t0:.1f}s)")
    np.savez_compressed(os.path.join(HERE, "e2e_wc.npz"), **out)


def gen_e2e_wc2():
    """gen_e2e_wc under the Matterport and ScanNet conventions (cases.WC2_CASES): the reference's dataset branches of util.warping /
    depth2pc (util.py:119-158,468-523), the face rotation Rs[(i-1)%4] of getPixel (rputil.py:88-119), the 'kinect' mask, S=21 and,
    for ScanNet, useTanh=0 -- poses after each step + perturbation envelope, like e2e_wc.npz."""
    out = {"amp": np.array(ENV_AMP), "n_seeds": np.array(ENV_SEEDS)}
    sig = np.tile(np.array([WC_SIGMAS]), (3, 1))
    nets = {}
    for ci, (ds, mm, S, tanh, seed, kw) in enumerate(WC2_CASES):
        if (S, tanh) not in nets:
            nets[(S, tanh)] = ref_net(S, tanh, None, sd=weights.make_descriptor_state_dict(WC_WEIGHT_SEED, S))
        net = nets[(S, tanh)]
        d, pts, ptw, T = synth.make_wc_pair(seed, dataset=ds, mask_method=mm, **kw)
        t0 = time.time()
        trace = ref_loop(net, d, pts, ptw, ds, mm, S, sig)
        env = np.zeros((ENV_SEEDS, 3))
        for k in range(ENV_SEEDS):
            tr = ref_loop(net, d, pts, ptw, ds, mm, S, sig, noise_amp=ENV_AMP, noise_seed=8500 + 100 * ci + k)
            env[k] = [np.linalg.norm(tr[s][:3, :3] - trace[s][:3, :3]) for s in range(3)]
        for s in range(3):
            out[f"wc2_{ci}_R{s}"] = trace[s]
        out[f"wc2_{ci}_T"] = T
        out[f"wc2_env_{ci}"] = env
        print(f"e2e_wc2 case {ci} {ds} seed {seed}: rot err vs true motion {[float(np.linalg.norm(trace[s][:3,:3]-T[:3,:3])) for s in range(3)]}"
              f"  envelope max {env.max(0)}  ({time.time()-t0:.1f}s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "e2e_wc2.npz"), **out)


def gen_keypoints():
    """rputil.Sampling of the reference on distance maps built like getKeypoint :182-190."""
    import torch
    R = ref_loader.load()
    ru = R["rputil"]
    rs = np.random.RandomState(77)
    out = {}
    for tag, n in (("a", 12), ("b", 30)):
        featt = torch.from_numpy(np.tanh(rs.randn(32, 160, 640)).astype(np.float32))
        fs = torch.from_numpy(np.tanh(rs.randn(32, n)).astype(np.float32))
        dist = (fs.unsqueeze(2) - featt.view(32, 1, -1)).pow(2).sum(0).view(n, 160, 640)
        pts = ru.Sampling(dist.numpy().copy(), 2)
        out[f"kp_{tag}_pts"] = pts
        idx = sample_idx(dist.numel(), 4096, 5)
        out[f"kp_{tag}_dist_idx"], out[f"kp_{tag}_dist_val"] = idx, dist.reshape(-1)[idx].numpy()
    np.savez_compressed(os.path.join(HERE, "keypoints.npz"), **out)


def gen_getkeypoint():
    """rputil.getKeypoint / getKeypoint_kinect of the REFERENCE (rputil.py:141-353) with its cv2 dependency replaced by a stub that
    returns FIXED detections (cases.GK_CASES: seeded sub-pixel points in the observed region) and the fixed-point BGR->gray
    conversion; np.random seeded right before the call.  Inputs are regenerated from seeds by the test (synth.make_keypoint_case)."""
    import types
    import torch
    from cases import GK_CASES
    R = ref_loader.load()
    ru = R["rputil"]
    from relativepose_amd import rputil as mine
    out = {}
    for ci, (kind, seed) in enumerate(GK_CASES):
        rs, rt, feats, featt, det_s, det_t, rs_full, rt_full = synth.make_keypoint_case(seed, kind)
        queue = [det_s, det_t]

        class FakeSift:
            def detectAndCompute(self, gray, mask):
                pts = queue.pop(0)
                return [types.SimpleNamespace(pt=(float(x), float(y))) for x, y in pts], None
        ru.cv2.COLOR_BGR2GRAY = 6
        ru.cv2.cvtColor = lambda img, code: mine.bgr2gray(img)
        ru.cv2.xfeatures2d = types.SimpleNamespace(SIFT_create=lambda **kw: FakeSift())
        np.random.seed(seed)
        if kind == "kinect":
            res = ru.getKeypoint_kinect(rs, rt, torch.from_numpy(feats), torch.from_numpy(featt), rs_full, rt_full)
        else:
            res = ru.getKeypoint(rs, rt, torch.from_numpy(feats), torch.from_numpy(featt))
        for name, a in zip(("pts", "ptsNorm", "ptsW", "ptt", "pttNorm", "pttW"), res):
            out[f"gk_{ci}_{name}"] = np.asarray(a)
        print(f"getkeypoint case {ci} {kind}: {len(res[0])} source / {len(res[3])} target keypoints, weights==1: {int((res[2]==1).sum())}/{int((res[5]==1).sum())}")
    np.savez_compressed(os.path.join(HERE, "getkeypoint.npz"), **out)


def gmp_inputs(ci):
    from cases import GK_CASES
    kind, seed = GK_CASES[ci]
    return (kind, seed) + synth.make_matching_primitive_case(seed, kind)


def gen_gmp_nc():
    """gmp_nc.npz: rpmodule.getMatchingPrimitive of the REFERENCE (rpmodule.py:511-538) with doCompletion = 0 AND 1 -- the 'ours_nc' method of
    evaluation.py:74 keeps the keypoints of the observed region only (:534-537) -- on one 'second' and one 'kinect' getKeypoint fixture (cv2 stub
    with fixed detections as in gen_getkeypoint, np.random seeded): all eight outputs."""
    import types
    import torch
    from relativepose_amd import rputil as mine
    R = ref_loader.load()
    rp, ru = R["rpmodule"], R["rputil"]
    out = {}
    for ci in (0, 2):
        kind, seed, ds, dS, dT, det_s, det_t = gmp_inputs(ci)
        for comp in (0, 1):
            queue = [det_s, det_t]

            class FakeSift:
                def detectAndCompute(self, gray, mask):
                    pts = queue.pop(0)
                    return [types.SimpleNamespace(pt=(float(x), float(y))) for x, y in pts], None
            ru.cv2.COLOR_BGR2GRAY = 6
            ru.cv2.cvtColor = lambda img, code: mine.bgr2gray(img)
            ru.cv2.xfeatures2d = types.SimpleNamespace(SIFT_create=lambda **kw: FakeSift())
            rp.getKeypoint, rp.getKeypoint_kinect = ru.getKeypoint, ru.getKeypoint_kinect
            np.random.seed(seed)
            tS = dict(dS, feat=torch.from_numpy(dS["feat"])); tT = dict(dT, feat=torch.from_numpy(dT["feat"]))
            res = rp.getMatchingPrimitive(tS, tT, ds, "skybox", comp)
            for name, a in zip(("pts3d", "ptt3d", "ptsns", "ptsnt", "dess", "dest", "ptsW", "pttW"), res):
                out[f"gmp_{ci}_c{comp}_{name}"] = np.asarray(a)
            print(f"gmp_nc case {ci} {kind} doCompletion={comp}: {res[0].shape[1]} source / {res[1].shape[1]} target primitives")
    np.savez_compressed(os.path.join(HERE, "gmp_nc.npz"), **out)


def gen_stats():
    """util.parse_data + util.point_cloud_overlap of the reference on synthetic pairs."""
    R = ref_loader.load()
    util = R["util"]
    out = {}
    for ds, mm, seed in GEOM_CASES:
        d = synth.make_pairs(1, seed + 40, ds)
        rgb_u8 = (d["rgb"] * 255).clip(0, 255).astype("uint8")
        res = util.parse_data(d["depth"], rgb_u8, d["norm"], ds, "ours")
        pc_src, pc_tgt = res[6], res[7]
        R_gt = np.matmul(d["R"][0, 1], np.linalg.inv(d["R"][0, 0]))
        ov = util.point_cloud_overlap(pc_src, pc_tgt, R_gt)
        out[f"stats_{ds}_n"] = np.array([len(pc_src), len(pc_tgt)])
        out[f"stats_{ds}_pc_head"] = pc_src[:128]
        out[f"stats_{ds}_overlap"] = np.array(ov, dtype=np.float64)
        out[f"stats_{ds}_Rgt"] = R_gt
    np.savez_compressed(os.path.join(HERE, "stats.npz"), **out)


def gen_stats_full():
    """util.parse_data's full-resolution ScanNet branch (util.py:78-90: the baseline methods) and util.depth2pc :497-507 of the reference."""
    R = ref_loader.load()
    util = R["util"]
    depth, rgb = synth.make_full_res_pair(77)
    res = util.parse_data(depth, rgb, None, "scannet", "gs")
    out = {"seed": np.array(77)}
    assert res[2] is None and res[3] is None
    for tag, pc, col in (("src", res[6], res[4]), ("tgt", res[7], res[5])):
        out[f"{tag}_n"] = np.array(len(pc))
        out[f"{tag}_pc_sha"] = np.array(hashlib.sha256(np.ascontiguousarray(pc, dtype=np.float64).tobytes()).hexdigest())
        out[f"{tag}_pc_head"] = pc[:256]
        out[f"{tag}_col_sha"] = np.array(hashlib.sha256(np.ascontiguousarray(col, dtype=np.float64).tobytes()).hexdigest())
    np.savez_compressed(os.path.join(HERE, "stats_full.npz"), **out)


def gen_metrics():
    """util.angular_distance_np of the reference (util.py:176-187) on random rotation pairs."""
    R = ref_loader.load()
    util = R["util"]
    rs = np.random.RandomState(31)
    Rh = np.stack([synth.random_rigid(rs)[:3, :3] for _ in range(16)])
    Rg = np.stack([synth.random_rigid(rs)[:3, :3] for _ in range(16)])
    Rh[3] = Rg[3]                                    # zero-angle case (arccos clip)
    out = {"R_hat": Rh, "R_gt": Rg, "angular_distance": util.angular_distance_np(Rh, Rg)}
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)


if __name__ == "__main__":
    assert ref_loader.available(), "reference not present"
    groups = sys.argv[1:] or ["matcher", "matcher_stages", "matcher_stages_big512", "tune", "matcher_big", "geometry", "scnet", "scnet_variants", "e2e", "e2e_env", "e2e_env_tf", "e2e_wc", "e2e_wc2", "stats", "stats_full", "keypoints", "getkeypoint", "gmp_nc", "metrics"]
    for g in groups:
        t = time.time()
        globals()["gen_" + g]()
        print(f"[{g}] done in {time.time()-t:.1f}s")
