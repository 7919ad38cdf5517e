"""CPU: the host half of the batched per-level keypoint derivation (rputil.keypoint_plan / keypoint_tables) against the REFERENCE's
getKeypoint / getKeypoint_kinect outputs (tests/golden/getkeypoint.npz): everything in a keypoint list that does not depend on the
features -- the (sampled) SIFT detections at its head and the kept random points at its tail -- is produced by the reference's
np.random stream in the reference's call order (kinect: choice, choice; then choice, choice, rand, rand, choice; rputil.py:141-353), so
equality of both pins that order.  The feature-dependent middle (the NMS picks) is the GPU test's (tests/test_gpu_keypoints.py)."""
import os

import numpy as np
import pytest

from cases import GK_CASES
from relativepose_amd import rputil, synth


@pytest.mark.parametrize("ci", range(len(GK_CASES)))
def test_plan_reproduces_the_feature_independent_parts_of_the_reference_output(golden_dir, ci):
    g = np.load(os.path.join(golden_dir, "getkeypoint.npz"))
    kind, seed = GK_CASES[ci]
    H, W = 160, 640
    _, _, _, _, det_s, det_t, _, _ = synth.make_keypoint_case(seed, kind)
    P = rputil.keypoint_plan(rputil.map_detections(det_s, kind, H), rputil.map_detections(det_t, kind, H), kind, H, W, np.random.RandomState(seed))
    pts, ptt = g[f"gk_{ci}_pts"], g[f"gk_{ci}_ptt"]
    na, nb, nt = len(P["src_a"]), len(P["src_b"]), len(P["tgt_a"])
    assert na == (300 if kind == "kinect" else len(det_s)) and nt == (300 if kind == "kinect" else len(det_t))
    assert np.array_equal(pts[:na], P["src_a"]) and np.array_equal(ptt[:nt], P["tgt_a"])          # (sampled) detections, panorama coordinates
    assert nb > 0 and np.array_equal(pts[len(pts) - nb:], P["src_b"])                              # the kept random points close the source list
    # the middle of the lists are NMS picks: integer pixel coordinates off the last row / column, at most topk per query
    mid_s, mid_t = pts[na:len(pts) - nb], ptt[nt:]
    assert len(mid_s) <= 2 * len(P["q2"]) and len(mid_t) <= 2 * (len(P["q1"]) + len(P["q3"]))
    for m in (mid_s, mid_t):
        assert np.array_equal(m, np.round(m)) and (m[:, 0] < W - 1).all() and (m[:, 1] < H - 1).all()
    # weights of the host-known parts (1 inside the observed region, 0.99 outside: the random points are drawn outside)
    assert (g[f"gk_{ci}_ptsW"][:na] == 1).all() and (g[f"gk_{ci}_ptsW"][len(pts) - nb:] == 0.99).all()
    # queries are rows of the host-known lists
    assert all(any(np.array_equal(q, r) for r in P["src_a"]) for q in P["q1"]) and np.array_equal(P["q3"], P["src_b"])


def test_tables_are_consistent():
    H, W = 160, 640
    plans = []
    for b, (kind, seed) in enumerate(GK_CASES[:2]):
        _, _, _, _, det_s, det_t, _, _ = synth.make_keypoint_case(seed, kind)
        plans.append(rputil.keypoint_plan(rputil.map_detections(det_s, kind, H), rputil.map_detections(det_t, kind, H), kind, H, W,
                                          np.random.RandomState(seed)))
    T = rputil.keypoint_tables(plans, H, W)
    B, topk = 2, 2
    assert T["q_off"][0] == 0 and T["q_off"][-1] == T["nq"] and (np.diff(T["q_off"]) >= 0).all() and T["nq_view_max"] == np.diff(T["q_off"]).max()
    for v in range(2 * B):
        q = slice(T["q_off"][v], T["q_off"][v + 1])
        assert (T["q_map"][q] == v).all() and (T["q_src"][q] == (v ^ 1)).all()          # a query samples one view and searches the other
    assert T["q_pt"].dtype == np.float32 and (T["q_pt"] >= 0).all() and (T["q_pt"] < 1).all()
    # every pick (query, k) appears in exactly one slot, in the view whose map the query searched; slots are host / pick / empty only
    picks = T["slot_kind"][T["slot_kind"] >= 0]
    assert sorted(picks.tolist()) == list(range(T["nq"] * topk))
    for v in range(2 * B):
        pk = T["slot_kind"][v][T["slot_kind"][v] >= 0]
        assert (T["q_map"][pk // topk] == v).all()
        kinds = T["slot_kind"][v]
        n = int((kinds != -1).sum())
        assert (kinds[:n] != -1).all() and (kinds[n:] == -1).all()                      # no holes: the reference's concatenation order
    # source list = [detections, picks of the target's queries, random points]; target = [detections, picks(q1), picks(q3)]
    n1, n2, n3, na = len(plans[0]["q1"]), len(plans[0]["q2"]), len(plans[0]["q3"]), len(plans[0]["src_a"])
    k0 = T["slot_kind"][0]
    assert (k0[:na] == -2).all() and (k0[na:na + 2 * n2] >= 0).all() and (k0[na + 2 * n2:na + 2 * n2 + n3] == -2).all()
    assert np.array_equal(T["slot_xy"][0, :na], plans[0]["src_a"])


def test_a_view_without_detections_gives_an_empty_plan_not_an_error():
    """the reference does not fail when a view has no SIFT detections -- getKeypoint returns None before drawing any random number
    (rputil.py:156-166), that level's pose is the identity and the loop continues (rpmodule.py:522-523, evaluation.py:280-282).  keypoint_plan
    returns an EMPTY plan for such a pair (no queries, no slots: both views get 0 keypoints and the matcher its "return identity" status), the
    random stream untouched, and keypoint_tables builds a batch's tables around it without disturbing the other pairs."""
    from relativepose_amd import rputil
    H, W = 160, 640
    det = np.stack((np.linspace(170, 300, 9), np.linspace(10, 140, 9)), 1)
    for ps, pt in ((np.zeros((0, 2)), det), (det, np.zeros((0, 2))), (np.zeros((0, 2)), np.zeros((0, 2)))):
        rng = np.random.RandomState(5)
        before = rng.get_state()[1].copy()
        P = rputil.keypoint_plan(ps, pt, "second", H, W, rng)
        assert P.get("empty") and all(len(P[k]) == 0 for k in ("q1", "q2", "q3", "src_a", "src_b", "tgt_a"))
        assert np.array_equal(rng.get_state()[1], before)           # no np.random call was consumed, like the reference's early return
    full = rputil.keypoint_plan(det, det + 3, "second", H, W, np.random.RandomState(5))
    empty = rputil.keypoint_plan(np.zeros((0, 2)), det, "second", H, W, np.random.RandomState(6))
    alone = rputil.keypoint_tables([full], H, W)
    both = rputil.keypoint_tables([empty, full], H, W)
    assert both["nq"] == alone["nq"] and both["L"] == alone["L"] and both["nq_view_max"] == alone["nq_view_max"]
    assert list(both["q_off"][:3]) == [0, 0, 0] and np.array_equal(both["q_off"][2:] , alone["q_off"])
    assert (both["slot_kind"][:2] == -1).all()
    # the second pair's tables are the lone pair's, re-based to views 2, 3 (its pick indices are unchanged: the empty pair has no queries)
    assert np.array_equal(both["slot_kind"][2:], alone["slot_kind"]) and np.array_equal(both["slot_xy"][2:], alone["slot_xy"])
    assert np.array_equal(both["q_map"], alone["q_map"] + 2) and np.array_equal(both["q_src"], alone["q_src"] + 2)
    only_empty = rputil.keypoint_tables([empty], H, W)
    assert only_empty["nq"] == 0 and only_empty["L"] == 1 and (only_empty["slot_kind"] == -1).all()
