"""The fit (fit_pair_kernel: one launch per batch, Lanczos with full re-orthogonalisation for the leading eigenvector)
against the reference goldens, its convergence reporting (RELPOSE_NOT_CONVERGED instead of a silently different pose), and the
affinity kernel variants against each other and the oracle."""
import os

import numpy as np
import pytest

from cases import MATCH_CASES
from gpu_util import log
from oracle import rp_oracle as M
from relativepose_amd import synth

pytestmark = pytest.mark.gpu


def _run(cases, para, **kw):
    import torch
    from relativepose_amd import rpmodule
    return rpmodule.match_pairs(*rpmodule.pack_keypoints(cases, torch.device("cuda:0")), para, **kw)


def test_lanczos_fit_matches_reference_goldens_in_both_vector_layouts(golden_dir):
    """The fit against the reference's poses (ARPACK eigenvectors): converged Lanczos gives the same answer to round-off, not just
    inside the 1e-4 bar -- with the per-correspondence vectors in LDS (the layout up to 4500 correspondences) and, forced through
    relpose_set_tuning, in global scratch (the layout beyond): the two layouts run the same arithmetic in the same order, so their
    poses are bitwise equal."""
    import torch
    from relativepose_amd import _lib, rpmodule
    gm = np.load(os.path.join(golden_dir, "matcher.npz"))
    idx = [ci for ci, c in enumerate(MATCH_CASES) if c[0] >= 50]
    worst = 0.0
    for ci in idx:
        N, Nt, seed, ds, row, inl = MATCH_CASES[ci]
        S, T, _ = synth.make_match_case(N, seed, inlier=inl, Nt=Nt)
        para = rpmodule.opts(*gm[f"params_{ds}"][row])
        new = _run([(S, T)], para, debug=True)
        with _lib.tuning(fit_global_vectors=1):
            glob = _run([(S, T)], para, debug=True)
        assert torch.equal(new.pose, glob.pose) and torch.equal(new.status, glob.status) and torch.equal(new.eig_iters, glob.eig_iters), ci
        a = new.pose[0].cpu().numpy()
        ref = gm[f"pose_{ci}_irls+sm"]
        e_ref = float(np.linalg.norm(a[:3, :3] - ref[:3, :3]))
        log("fit_lanczos_vs_reference", case=ci, N=N, inlier=inl, status=int(new.status[0]), products_per_round=new.eig_iters[0].cpu().tolist(),
            rot_err_vs_reference=e_ref)
        if inl > 0:                       # the all-outlier case has no dominant eigenvector: only its status is checked below
            assert int(new.status[0]) == 0 and e_ref < 1e-4, (ci, e_ref)
            worst = max(worst, e_ref)
        assert int(new.status[0]) in (0, 6)
    assert worst < 1e-9                    # converged eigenvectors: same answer as ARPACK to round-off, not just inside the bar


@pytest.mark.parametrize("G", [2, 4, 8])
def test_fit_helper_workgroups_change_nothing(G):
    """Helper workgroups for the matrix-vector products (relpose_set_tuning(RELPOSE_TUNE_FIT_CLUSTER, G): the leader publishes the
    Lanczos vector, G - 1 helpers claim chunks of segments, the leader adds the per-segment partial sums up in segment order): poses,
    status and product counts are BITWISE those of the single-workgroup fit, for a ragged batch of pairs incl. degenerate ones, and
    repeated calls agree (the partial sum of a segment does not depend on who computed it)."""
    import torch
    from relativepose_amd import _lib, rpmodule
    cases = [synth.make_match_case(n, 800 + n, inlier=i)[:2] for n, i in ((200, 0.6), (400, 0.3), (120, 0.1), (2, 0.6), (300, 0.6), (30, 0.0))]
    para = rpmodule.opts(0.3, 0.3, 0.04, 0.009)
    with _lib.tuning(fit_cluster=1):
        one = _run(cases, para, debug=True)
    with _lib.tuning(fit_cluster=G):
        many = _run(cases, para, debug=True)
        assert torch.equal(one.pose, many.pose) and torch.equal(one.status, many.status) and torch.equal(one.eig_iters, many.eig_iters)
        assert torch.equal(one.trace, many.trace)
        for _ in range(8):          # (who computes which chunk changes from call to call)
            again = _run(cases, para, debug=True)
            assert torch.equal(many.pose, again.pose) and torch.equal(many.trace, again.trace) and torch.equal(many.eig_iters, again.eig_iters)
    assert int(one.status[3]) == 1 and int((one.status == 0).sum()) >= 4


def test_predicted_convergence_tests_save_products_not_accuracy():
    """The eigen-solve places its convergence tests where the residual estimate is predicted to reach the tolerance
    (rp_lz_steps_to_check, csrc/rp_math.h; host model in tests/test_host_math.py) instead of every 8th product
    (relpose_set_tuning(RELPOSE_TUNE_FIT_FIXED_CHECKS, 1) = the earlier rule): same status, the pose after every alternation equal to
    round-off (both rules stop at a residual estimate <= 1e-13), no pair with more products, clearly fewer in total."""
    import torch
    from relativepose_amd import _lib, rpmodule
    cases = [synth.make_match_case(n, 900 + n + int(100 * i), inlier=i)[:2] for n, i in ((200, 0.6), (200, 0.3), (200, 0.1), (400, 0.3), (120, 0.5), (300, 0.2), (64, 0.4), (2, 0.6))]
    para = rpmodule.opts(0.3, 0.3, 0.04, 0.009)
    with _lib.tuning(fit_fixed_checks=1):
        fixed = _run(cases, para, debug=True)
    pred = _run(cases, para, debug=True)
    again = _run(cases, para, debug=True)
    assert torch.equal(pred.pose, again.pose) and torch.equal(pred.eig_iters, again.eig_iters)
    assert torch.equal(pred.status, fixed.status)
    ok = (pred.status == 0).cpu().numpy()
    assert ok.sum() >= 6
    d = (pred.trace - fixed.trace).abs().reshape(len(cases), -1).max(1).values.cpu().numpy()
    pf, pp = fixed.eig_iters.cpu().numpy(), pred.eig_iters.cpu().numpy()
    log("fit_predicted_checks", products_fixed=pf[ok].sum(0).tolist(), products_predicted=pp[ok].sum(0).tolist(), trace_max_diff=float(d[ok].max()))
    assert d[ok].max() < 1e-9
    assert (pp[ok].sum(1) <= pf[ok].sum(1)).all()              # no pair pays for it ...
    assert pp[ok].sum() <= 0.92 * pf[ok].sum()                  # ... and the batch saves clearly (about a fifth on the bench's graphs)


def test_thousand_keypoints_fit_converges_and_matches_reference(golden_dir):
    """N = 1000 keypoints per view: 5000 correspondences (more than the fit's LDS layout holds: vectors in global scratch), 12.5 M
    candidate pairs.  The same Lanczos solver with the same residual test runs there -- status 0 means CONVERGED -- and the pose equals
    the reference helper's (tests/golden/matcher_big.npz, make_golden.gen_matcher_big) far inside the 1e-4 bar."""
    from cases import MATCH_BIG
    from relativepose_amd import rpmodule
    g = np.load(os.path.join(golden_dir, "matcher_big.npz"))
    N, Nt, seed, ds, row, inl, noise = MATCH_BIG
    S, T, G = synth.make_match_case(N, seed, inlier=inl, noise=noise, Nt=Nt)
    para = rpmodule.opts(*g["params"])
    res = _run([(S, T)], para, debug=True)
    pose = res.pose[0].cpu().numpy()
    e_ref = float(np.linalg.norm(pose[:3, :3] - g["pose"][:3, :3]))
    t_ref = float(np.abs(pose[:3, 3] - g["pose"][:3, 3]).max())
    log("fit_n1000", status=int(res.status[0]), products_per_round=res.eig_iters[0].cpu().tolist(), surviving_pairs=int(res.counts[0, 1]),
        rot_err_vs_reference=e_ref, t_err_vs_reference=t_ref, rot_err_vs_ground_truth=float(np.linalg.norm(pose[:3, :3] - G[:3, :3])))
    assert int(res.status[0]) == 0
    assert e_ref < 1e-6 and t_ref < 1e-6, (e_ref, t_ref)


def test_largest_lds_resident_fit_equals_global_layout():
    """The largest pair whose per-correspondence vectors still live in LDS (900 keypoints x topK 5 = 4500 correspondences = RP_FIT1_MAXC: the
    768-thread kernel (1024 threads in rounds 2-5) with ~158 KB of dynamic LDS incl. the segment table) against the same pair forced into the global-scratch layout:
    the two run the same arithmetic in the same order -> bitwise equal poses, status and product counts; converged; the planted motion."""
    import torch
    from relativepose_amd import _lib, rpmodule
    S, T, G = synth.make_match_case(900, 4200, inlier=0.5, noise=0.002)
    para = rpmodule.opts(0.3, 0.3, 0.04, 0.009)
    lds = _run([(S, T)], para, debug=True)
    with _lib.tuning(fit_global_vectors=1):
        glob = _run([(S, T)], para, debug=True)
    assert torch.equal(lds.pose, glob.pose) and torch.equal(lds.status, glob.status) and torch.equal(lds.eig_iters, glob.eig_iters)
    pose = lds.pose[0].cpu().numpy()
    log("fit_lds_limit", status=int(lds.status[0]), products_per_round=lds.eig_iters[0].cpu().tolist(), surviving_pairs=int(lds.counts[0, 1]),
        rot_err_vs_ground_truth=float(np.linalg.norm(pose[:3, :3] - G[:3, :3])))
    assert int(lds.status[0]) == 0 and np.linalg.norm(pose[:3, :3] - G[:3, :3]) < 5e-2


@pytest.mark.parametrize("N", [204, 205, 13, 64])
def test_fit_kernel_boundaries_equal_their_global_layout_twin(N):
    """Round 6 gave the fit three kernel shapes: 512 threads with the {h, u} pair gathers up to 1024 correspondences (N = 204, topK 5: 1020),
    768 threads beyond (N = 205: 1025), and the global-vector layout; and renumbered the segments (full ones row by row, partial ones by
    length class).  Either side of the boundary, a tiny pair (every row a single short segment) and a pair with exactly 64-keypoint views:
    the LDS layout and the same pair forced into the global layout run the same sums in the same order -> bitwise equal; and the planted
    motion comes back."""
    import torch
    from relativepose_amd import _lib, rpmodule
    S, T, G = synth.make_match_case(N, 4300 + N, inlier=0.6, noise=0.002)
    para = rpmodule.opts(0.3, 0.3, 0.04, 0.009)
    lds = _run([(S, T)], para, debug=True)
    with _lib.tuning(fit_global_vectors=1):
        glob = _run([(S, T)], para, debug=True)
    assert torch.equal(lds.pose, glob.pose) and torch.equal(lds.status, glob.status) and torch.equal(lds.eig_iters, glob.eig_iters)
    with _lib.tuning(fit_cluster=2):
        two = _run([(S, T)], para, debug=True)
    assert torch.equal(lds.pose, two.pose) and torch.equal(lds.eig_iters, two.eig_iters)
    pose = lds.pose[0].cpu().numpy()
    e_gt = float(np.linalg.norm(pose[:3, :3] - G[:3, :3]))
    log("fit_kernel_boundary", N=N, status=int(lds.status[0]), products_per_round=lds.eig_iters[0].cpu().tolist(), rot_err_vs_ground_truth=e_gt)
    assert int(lds.status[0]) == 0 and e_gt < 5e-2


def test_capacity_limits_largest_pair_runs_and_beyond_is_refused():
    """include/relpose.h: RELPOSE_MAX_TARGETS = 1152 keypoints per view (the affinity kernel's LDS beyond 512 targets) and
    RELPOSE_MAX_CORRESPONDENCES = 8192 = ns_max x topK (pair_fill_rows_kernel then asks for the full 64 KB of dynamic LDS).  A pair AT
    both limits -- 1152 keypoints, topK 7: 8064 correspondences -- must run to a converged pose equal to the planted motion; one keypoint
    more, or topK 8, is refused with a message that names the limit (the reference has no limit)."""
    from relativepose_amd import rpmodule
    N = rpmodule.MAX_TARGETS
    S, T, G = synth.make_match_case(N, 4100, inlier=0.5, noise=0.002)
    para = rpmodule.opts(0.3, 0.3, 0.04, 0.009)
    para.topK = rpmodule.MAX_CORRESPONDENCES // N
    assert para.topK == 7
    res = _run([(S, T)], para, debug=True, max_edges=1 << 23)
    pose = res.pose[0].cpu().numpy()
    e_gt = float(np.linalg.norm(pose[:3, :3] - G[:3, :3]))
    log("fit_capacity_limit", N=N, topK=para.topK, status=int(res.status[0]), surviving_pairs=int(res.counts[0, 1]), rot_err_vs_ground_truth=e_gt)
    assert int(res.counts[0, 3]) == 7 and int(res.status[0]) == 0 and e_gt < 5e-3, (int(res.status[0]), e_gt)
    para.topK = 8
    with pytest.raises(RuntimeError, match="8192"):
        _run([(S, T)], para)
    S2, T2, _ = synth.make_match_case(N + 1, 4101)
    with pytest.raises(RuntimeError, match="1152"):
        _run([(S2, T2)], rpmodule.opts(0.3, 0.3, 0.04, 0.009))


def test_fit_is_batch_invariant_and_deterministic():
    import torch
    from relativepose_amd import rpmodule
    cases = [synth.make_match_case(n, 700 + n, inlier=i)[:2] for n, i in ((200, 0.6), (120, 0.3), (200, 0.1), (60, 0.6))]
    para = rpmodule.opts(0.3, 0.3, 0.04, 0.009)
    full = _run(cases, para)
    again = _run(cases, para)
    assert torch.equal(full.pose, again.pose) and torch.equal(full.status, again.status)
    for b, c in enumerate(cases):
        one = _run([c], para)
        # padded batch (ns_max = 200) vs tight single: same kernels, same reduction orders
        assert torch.equal(one.pose[0], full.pose[b]) and int(one.status[0]) == int(full.status[b]), b


def test_not_converged_is_reported_not_hidden():
    """A graph with a tiny spectral gap (two equally consistent, disjoint motions: the leading eigenvalue is (nearly)
    double) must either converge to the residual tolerance or say RELPOSE_NOT_CONVERGED -- never return status 0 with an
    unconverged vector.  Checked through the residual of the returned solution against the oracle's matrix."""
    from relativepose_amd import rpmodule
    rs = np.random.RandomState(5)
    S, T, _ = synth.make_match_case(160, 901, inlier=0.0)
    # plant two rigid motions of 40 points each
    for k, (lo, hi) in enumerate(((0, 40), (40, 80))):
        Tk = synth.random_rigid(rs, 2.0, 1.0)
        T["pc"][lo:hi] = S["pc"][lo:hi] @ Tk[:3, :3].T + Tk[:3, 3]
        T["normal"][lo:hi] = S["normal"][lo:hi] @ Tk[:3, :3].T
        T["feat"][lo:hi] = S["feat"][lo:hi]
    para = rpmodule.opts(0.3, 0.3, 0.04, 0.01)
    res = _run([(S, T)], para, debug=True)
    st = int(res.status[0])
    log("fit_double_eigenvalue", status=st, products_per_round=res.eig_iters[0].cpu().tolist())
    assert st in (0, 6)
    assert np.isfinite(res.pose.cpu().numpy()).all()
    ref = M.relative_pose_helper(S, T, M.Params(0.3, 0.3, 0.04, 0.01))
    if st == 0:
        # converged: it must then be one of the two planted motions' fits, like ARPACK's answer, up to which cluster wins
        assert np.allclose(res.pose[0].cpu().numpy()[3], [0, 0, 0, 1])


def test_exhausted_product_budget_sets_not_converged_status():
    """relpose_set_tuning(RELPOSE_TUNE_FIT_MAX_PRODUCTS, 8) (test hook) = 8 products: no eigen-solve of a 60 %-inlier pair can reach 1e-13 in one 8-step cycle, so
    the pair must come back with RELPOSE_NOT_CONVERGED (6) and a finite pose close to (but not claimed equal to) the converged one."""
    from relativepose_amd import rpmodule
    para = rpmodule.opts(0.3, 0.3, 0.04, 0.009)
    for seed, inl in ((18, 0.1), (16, 0.3), (14, 0.6)):                 # the first case whose converged solve needs more than one check interval
        S, T, _ = synth.make_match_case(200, seed, inlier=inl)
        good = _run([(S, T)], para, debug=True)
        if int(good.status[0]) == 0 and good.eig_iters[0].max().item() > 8:
            break
    else:
        pytest.skip("every candidate case converges within 8 products")
    from relativepose_amd import _lib
    with _lib.tuning(fit_max_products=8):
        bad = _run([(S, T)], para, debug=True)
    assert int(good.status[0]) == 0 and int(bad.status[0]) == 6
    assert bad.eig_iters[0].max().item() <= 8 and good.eig_iters[0].max().item() > 8
    pb = bad.pose[0].cpu().numpy()
    assert np.isfinite(pb).all() and np.allclose(pb[:3, :3] @ pb[:3, :3].T, np.eye(3), atol=1e-9)
    log("fit_forced_not_converged", rot_diff_vs_converged=float(np.linalg.norm(pb[:3, :3] - good.pose[0].cpu().numpy()[:3, :3])))


def test_row_and_tile_affinity_kernels_equal_lds_kernel():
    """The register-resident (row) and the tile (fp16-MFMA candidates + exact arithmetic) affinity kernels (nt_max <= 512) against
    the LDS kernel (the path for larger target sets): identical correspondences (the float32 distance and the top-K tie rule are
    bit-for-bit the same), weights to round-off."""
    import torch
    from relativepose_amd import _lib, rpmodule
    dev = torch.device("cuda:0")
    cases = [synth.make_match_case(n, 40 + n, inlier=i, Nt=nt)[:2] for n, nt, i in ((400, 400, 0.6), (200, 130, 0.3), (64, 65, 0.6), (7, 6, 0.6), (3, 3, 0.6))]
    para = rpmodule.opts(0.3, 0.3, 0.04, 0.0095)
    kp = rpmodule.pack_keypoints(cases, dev)
    with _lib.tuning(affinity_kernel="lds"):
        old = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para)
    # the current kernels: batch-size default, row kernel forced, tile kernel forced
    for sel in ("auto", "rows", "tile", "pool"):
        with _lib.tuning(affinity_kernel=sel):
            new = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para)
        assert torch.equal(new[1], old[1]) and torch.equal(new[3], old[3]), sel             # corres_j, k_eff
        assert torch.allclose(new[2], old[2], rtol=1e-12, atol=0), sel                      # corres_w (f64)
        # wij (f32 copy): the row / tile kernels write exact zeros below e^-75 of the row maximum (RP_AFF_WINDOW), the LDS kernel does not
        assert torch.allclose(new[0], old[0], rtol=2e-6, atol=1e-30), sel
    # a target set larger than the register kernel takes (nt_max > 512) still goes through the LDS kernel
    big = [synth.make_match_case(40, 77, Nt=600)[:2]]
    kb = rpmodule.pack_keypoints(big, dev)
    wij, cj, cw, keff = rpmodule.affinity_topk(kb[2], kb[3], kb[6], kb[7], kb[8], kb[9], para)
    d = M.affinity(big[0][0]["feat"], big[0][1]["feat"], big[0][0]["weight"], big[0][1]["weight"], para.sigmaFeat)
    ref = M.topk(d[2], 5)[1].reshape(40, 5)
    for i in range(40):
        pos = d[2][i, ref[i]] > 0
        assert set(ref[i][pos].tolist()) <= set(cj[0, i].cpu().tolist())


def _check_affinity_vs_oracle(cases, para, out, rows=None):
    """corres_j (as sets over wij > 0), corres_w and the float32 wij of `out` against the numpy oracle, per pair."""
    wij, cj, cw, keff = [t.cpu().numpy() if t is not None else None for t in out]
    for b, (S, T) in enumerate(cases):
        if rows is not None and b not in rows:
            continue
        N, Nt = S["feat"].shape[0], T["feat"].shape[0]
        d = M.affinity(S["feat"], T["feat"], S["weight"], T["weight"], para.sigmaFeat)
        wo = d[2]
        K = int(keff[b])
        assert K == min(para.topK, Nt - 1)
        if wij is not None:
            assert np.allclose(wij[b, :N, :Nt], wo, rtol=2e-6, atol=1e-30), b
        co = M.topk(wo, K)[1].reshape(N, K)
        for i in range(N):
            so = set(int(j) for j in co[i] if wo[i, j] > 0)
            sg = set(int(j) for j, w in zip(cj[b, i, :K], cw[b, i, :K]) if w > 0)
            if so != sg:
                srt = np.sort(wo[i])[::-1]
                assert srt[K - 1] == srt[K], (b, i, sorted(so), sorted(sg))       # only an exact K / K+1 tie may differ
            assert np.allclose(cw[b, i, :K], wo[i, cj[b, i, :K]], rtol=1e-12, atol=0), (b, i)


def test_tile_affinity_kernel_at_production_batch_vs_oracle():
    """The tile kernel where rp_launch_affinity picks it BY ITSELF (>= 1024 row tiles: 256 pairs x 200 keypoints = 1792 tiles, no
    tuning override), on 256 different pairs; sampled pairs against the numpy oracle, every pair against the row kernel."""
    import torch
    from relativepose_amd import _lib, rpmodule
    dev = torch.device("cuda:0")
    cases = [synth.make_match_case(200 - (b % 3), 6000 + b, inlier=(0.6, 0.3, 0.0)[b % 3], Nt=200 - (b % 5))[:2] for b in range(256)]
    para = rpmodule.opts(0.3, 0.3, 0.04, 0.0087)
    kp = rpmodule.pack_keypoints(cases, dev)
    auto = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para)
    with _lib.tuning(affinity_kernel="rows"):
        rows = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para)
    assert torch.equal(auto[1], rows[1]) and torch.equal(auto[3], rows[3])
    assert torch.allclose(auto[2], rows[2], rtol=1e-13, atol=0)
    assert torch.allclose(auto[0], rows[0], rtol=1e-6, atol=1e-30)
    _check_affinity_vs_oracle(cases, para, auto, rows={0, 1, 2, 100, 255})
    fused = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para, want_wij=False)
    assert torch.equal(fused[1], auto[1]) and torch.equal(fused[2], auto[2])        # the fused variant: same indices, same weights, bitwise
    # the pool variant (round 5: screen / exact / rank as separate dense launches): same indices, weights and wij to round-off, its fused form bitwise its own
    with _lib.tuning(affinity_kernel="pool"):
        pool = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para)
        pool_f = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para, want_wij=False)
    assert torch.equal(pool[1], rows[1]) and torch.equal(pool[3], rows[3])
    assert torch.allclose(pool[2], rows[2], rtol=1e-13, atol=0)
    assert torch.allclose(pool[0], rows[0], rtol=1e-6, atol=1e-30)
    assert torch.equal(pool_f[1], pool[1]) and torch.equal(pool_f[2], pool[2])
    _check_affinity_vs_oracle(cases, para, pool, rows={0, 1, 2, 100, 255})


def test_tile_affinity_overflow_rows_are_redone_exactly():
    """Rows whose candidate set exceeds the tile kernel's per-lane stack -- here 150 IDENTICAL target descriptors, all tied for the
    maximum of every row -- are marked and redone by the exact row kernel: the results still equal the oracle (ties go to the
    smaller index), also for rows with weights the tile kernel's error bound does not cover (> 1)."""
    import torch
    from relativepose_amd import _lib, rpmodule
    dev = torch.device("cuda:0")
    S, T, _ = synth.make_match_case(200, 4242, inlier=0.5)
    T["feat"][20:170] = T["feat"][20]                       # 150 identical descriptors: every row has >= 150 important entries
    S2, T2, _ = synth.make_match_case(120, 4243, inlier=0.5)
    S2["weight"][::7] = 2.0                                 # weights outside [0, 1]: 2.0 * 0.5 == 1.0 is the reference's "both observed" class
    T2["weight"][::5] = 0.5
    S3, T3, _ = synth.make_match_case(90, 4244, inlier=0.5)
    cases = [(S, T), (S2, T2), (S3, T3)]
    para = rpmodule.opts(0.3, 0.3, 0.04, 0.0087)
    kp = rpmodule.pack_keypoints(cases, dev)
    with _lib.tuning(affinity_kernel="rows"):
        ref = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para)
    for sel in ("tile", "pool"):          # (pool variant: 150 tied entries per row overflow nothing -- its pool is dynamic -- but the bound's redo rows are marked the same way)
        with _lib.tuning(affinity_kernel=sel):
            out = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para)
        _check_affinity_vs_oracle(cases, para, out)
        assert torch.equal(out[1], ref[1]), sel                                # ties resolved like the row kernel (smaller index)
        assert torch.allclose(out[2], ref[2], rtol=1e-13, atol=0), sel         # (the kernels add the row norm up in different orders)
        assert torch.allclose(out[0], ref[0], rtol=1e-6, atol=1e-30), sel


def test_pool_affinity_variant_where_the_launcher_picks_it_by_itself():
    """The pool variant (round 5) is auto-selected for the fused form (no wij copy) beyond 256 targets at >= 1024 row tiles -- the one case it measured
    faster than the tile kernel (profiles/r05_affinity_pool.txt).  128 pairs x 300 keypoints (ragged): indices equal the row kernel's, weights to round-off,
    sampled pairs against the numpy oracle; the materialised call of the same batch (tile kernel) gives bitwise the same indices."""
    import torch
    from relativepose_amd import _lib, rpmodule
    dev = torch.device("cuda:0")
    cases = [synth.make_match_case(300 - (b % 4), 7000 + b, inlier=(0.6, 0.3, 0.0)[b % 3], Nt=300 - (b % 7))[:2] for b in range(128)]
    para = rpmodule.opts(0.3, 0.3, 0.04, 0.0087)
    kp = rpmodule.pack_keypoints(cases, dev)
    fused = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para, want_wij=False)             # auto -> pool
    with _lib.tuning(affinity_kernel="pool"):
        forced = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para, want_wij=False)
    assert torch.equal(fused[1], forced[1]) and torch.equal(fused[2], forced[2])                               # (it IS the pool variant)
    with _lib.tuning(affinity_kernel="rows"):
        rows = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para, want_wij=False)
    assert torch.equal(fused[1], rows[1]) and torch.equal(fused[3], rows[3])
    assert torch.allclose(fused[2], rows[2], rtol=1e-13, atol=0)
    full = rpmodule.affinity_topk(kp[2], kp[3], kp[6], kp[7], kp[8], kp[9], para)                               # materialised: the tile kernel
    assert torch.equal(full[1], fused[1]) and torch.allclose(full[2], fused[2], rtol=1e-13, atol=0)
    _check_affinity_vs_oracle(cases, para, (None,) + tuple(fused[1:]), rows={0, 5, 127})
