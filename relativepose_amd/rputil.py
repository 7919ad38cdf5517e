"""Host-side mirror of the reference's RPModule/rputil.py, backed by the HIP library:

  opts            rputil.py:11-22      (re-exported from rpmodule)
  interpolate     rputil.py:43-58      bilinear descriptor sampling
  getPixel        rputil.py:88-119     bilinear depth / normal sampling + unprojection
  Sampling        rputil.py:355-371    NMS keypoint sampling on distance maps
  feature_distance_map_dev             the descriptor-to-map distance of getKeypoint :182-190

  getKeypoint / getKeypoint_kinect  rputil.py:141-353  everything AROUND the SIFT detector: descriptor sampling at the detected points,
                                   feature-guided augmentation (distance maps + NMS on the GPU), random fill, weights

  keypoint_plan / keypoint_tables / keypoints_reference_dev   the same derivation batched over the views of a batch of scan pairs and kept
                                   on the device (csrc/keypoints.hip): the host pre-draws getKeypoint's np.random stream (it does not depend on
                                   the features), the device does the feature-dependent part -- what RelativePosePipeline(keypoints="reference")
                                   runs at every recurrent level

SIFT detection itself (cv2.xfeatures2d, third-party) is a hook: set_sift_detector(fn), fn(gray uint8 [h,w]) -> [[x, y], ...]; when
cv2 with xfeatures2d is importable it is the default."""
import numpy as np

from . import _lib


def feature_distance_map_dev(query, feat):
    """query [nsel,32] f32, feat [32,H,W] f32 (CUDA) -> dist [nsel,H,W] f32."""
    import torch
    _lib.require_gpu()
    nsel, (C, H, W) = query.shape[0], feat.shape
    assert C == 32 and query.shape[1] == 32
    dist = torch.empty(nsel, H, W, dtype=torch.float32, device=feat.device)
    rc = _lib.lib().relpose_feature_distance_map(_lib.ptr(query.contiguous()), _lib.ptr(feat.contiguous()), _lib.ptr(dist), nsel, H, W,
                                                 _lib.stream_ptr())
    _lib.check(rc, "relpose_feature_distance_map")
    return dist


def sampling_dev(dist, K, window=15):
    """dist [n,H,W] f32 CUDA -> pts [n,K,2] f64 (x,y): Sampling(np.exp(-dist/2)) of the reference."""
    import torch
    n, H, W = dist.shape
    pts = torch.empty(n, K, 2, dtype=torch.float64, device=dist.device)
    rc = _lib.lib().relpose_nms_sampling(_lib.ptr(dist.contiguous()), _lib.ptr(pts), n, H, W, K, window, _lib.stream_ptr())
    _lib.check(rc, "relpose_nms_sampling")
    return pts


def Sampling(heatmap, K):
    """rputil.py:355: numpy [n,h,w] distance maps (the reference passes dist and exponentiates inside) -> [n,K,2]."""
    import torch
    dev = _lib.require_gpu()
    return sampling_dev(torch.from_numpy(np.ascontiguousarray(heatmap, dtype=np.float32)).to(dev), K).cpu().numpy()


def interpolate(feat, pt):
    """rputil.py:43-58, same signature: feat torch [c,h,w] float32, pt torch [k,2] normalised (x/W, y/H) -> torch [c,k]
    (on the GPU; CPU tensors are uploaded, like torch_op.v does in the reference)."""
    import torch
    dev = _lib.require_gpu()
    f = feat.to(dev, torch.float32).contiguous()
    p = pt.to(dev, torch.float32).contiguous()
    c, h, w = f.shape
    k = p.shape[0]
    if k and not bool(((p >= 0) & (p < 1)).all()):
        raise IndexError("interpolate: pt must lie in [0, 1) (the reference indexes feat[:, y0 + 1, x0 + 1], rputil.py:52-55)")
    out = torch.empty(c, k, dtype=torch.float32, device=dev)
    if k:
        _lib.check(_lib.lib().relpose_interpolate(_lib.ptr(f), _lib.ptr(p), _lib.ptr(out), c, h, w, k, _lib.stream_ptr()),
                   "relpose_interpolate")
    return out


def getPixel(depth, normal, pts, dataset='suncg', representation='skybox'):
    """rputil.py:88-119, same signature: depth numpy [h,4h], normal numpy [h,4h,3], pts numpy [k,2] pixel coords
    (x <= 4h-2, y <= h-2) -> (pc [3,k], nn [k,3]) float64 numpy."""
    import torch
    from .util import dataset_id
    assert representation == 'skybox'                                    # rputil.py:62
    depth = np.ascontiguousarray(depth, dtype=np.float64)
    normal = np.ascontiguousarray(normal, dtype=np.float64)
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    h = depth.shape[0]
    assert depth.shape[1] == 4 * h and normal.shape == (h, 4 * h, 3)    # the reference asserts 160 x 640 (:63-64)
    k = pts.shape[0]
    if k == 0:
        return np.zeros((3, 0)), np.zeros((0, 3))
    if not ((pts[:, 0] >= 0) & (pts[:, 0] < 4 * h - 1) & (pts[:, 1] >= 0) & (pts[:, 1] < h - 1)).all():
        raise IndexError("getPixel: pts must satisfy 0 <= x < 4h-1, 0 <= y < h-1 (the reference reads depth[y + 1, x + 1], rputil.py:92-95)")
    dev = _lib.require_gpu()
    d, n, p = (torch.from_numpy(a).to(dev) for a in (depth, normal, pts))
    pc = torch.empty(k, 3, dtype=torch.float64, device=dev)
    nn = torch.empty(k, 3, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().relpose_get_pixel(_lib.ptr(d), _lib.ptr(n), _lib.ptr(p), k, h, dataset_id(dataset), _lib.ptr(pc), _lib.ptr(nn),
                                            _lib.stream_ptr()), "relpose_get_pixel")
    return pc.cpu().numpy().T, nn.cpu().numpy()


# ---- keypoint assembly (rputil.py:141-353) ------------------------------------------------------------------------------------
_sift_detector = None


def set_sift_detector(fn):
    """fn(gray: uint8 [h, w]) -> array-like [[x, y], ...] (sub-pixel, image coordinates of `gray`): the detector behind getKeypoint
    / getKeypoint_kinect (the reference: cv2.xfeatures2d.SIFT_create(contrastThreshold=0.02).detectAndCompute, rputil.py:152-156).
    Returns the previous detector."""
    global _sift_detector
    old, _sift_detector = _sift_detector, fn
    return old


def _detect(gray):
    if _sift_detector is not None:
        # a COPY: getKeypoint / getKeypoint_kinect shift and rescale the detections in place, and the hook may hand out an array it keeps
        return np.array(_sift_detector(gray), dtype=np.float64, copy=True).reshape(-1, 2)
    try:
        import cv2
        try:
            sift = cv2.xfeatures2d.SIFT_create(contrastThreshold=0.02)          # (the reference's call, rputil.py:152)
        except AttributeError:
            sift = cv2.SIFT_create(contrastThreshold=0.02)                      # OpenCV >= 4.4: SIFT moved into the main module
    except Exception as e:
        raise RuntimeError("relativepose_amd.rputil: no SIFT detector (neither cv2.xfeatures2d.SIFT_create nor cv2.SIFT_create is "
                           "available); install one with rputil.set_sift_detector(fn)") from e
    kps, _ = sift.detectAndCompute(gray, None)
    return np.array([k.pt for k in kps], dtype=np.float64).reshape(-1, 2)


def bgr2gray(img):
    """cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) for uint8 images: OpenCV's 14-bit fixed-point weights (B 1868, G 9617, R 4899)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def _v(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).float()


def _augment(query_desc, feat_other, topk, W, H):
    """rputil.py:186-196: for every query descriptor (columns of query_desc [C, n]) the `topk` NMS minima of its squared distance
    to every pixel of the other view's feature map (dense distance map + Sampling, both on the GPU), as [n*topk, 2] pixel coords,
    points on the last row / column dropped."""
    if query_desc.shape[1] == 0:
        return np.zeros((0, 2))
    dist = feature_distance_map_dev(query_desc.t().contiguous(), feat_other)
    aug = sampling_dev(dist, topk).cpu().numpy().reshape(-1, 2)
    return aug[(aug[:, 0] < W - 1) * (aug[:, 1] < H - 1)]


def _finish(pts, ptt, inside, W, H, marker):
    ptsNorm, pttNorm = pts.astype('float').copy(), ptt.astype('float').copy()
    ptsNorm[:, 0] /= W; ptsNorm[:, 1] /= H
    pttNorm[:, 0] /= W; pttNorm[:, 1] /= H
    ptsW, pttW = np.ones(len(pts)), np.ones(len(ptt))
    ptsW[~inside(pts)] *= marker
    pttW[~inside(ptt)] *= marker
    return pts, ptsNorm, ptsW, ptt, pttNorm, pttW


def _keypoints_common(pts, ptt, feats, featt, inside, n_match, n_random_draw, n_random_keep, marker=0.99, topk=2):
    """The part of getKeypoint / getKeypoint_kinect behind the detector (rputil.py:169-237 / :284-353), same order of np.random calls
    as the reference (choice, choice, rand, rand, choice): detected points -> descriptors -> cross-view augmentation -> random points of
    the unobserved region with their augmentation -> weights (1 inside the observed region, `marker` outside)."""
    import torch
    dev = _lib.require_gpu()
    feats, featt = feats.to(dev, torch.float32).contiguous(), featt.to(dev, torch.float32).contiguous()
    C, H, W = feats.shape
    nrm = lambda p: p.astype('float') / np.array([W, H], dtype=np.float64)
    fs0 = interpolate(feats, _v(nrm(pts)))
    ft0 = interpolate(featt, _v(nrm(ptt)))
    fsselect = np.random.choice(range(pts.shape[0]), min(n_match, pts.shape[0]))
    ftselect = np.random.choice(range(ptt.shape[0]), min(n_match, ptt.shape[0]))
    sel = lambda f, idx: f[:, torch.from_numpy(np.asarray(idx, dtype=np.int64)).to(dev)]
    pttAug = _augment(sel(fs0, fsselect), featt, topk, W, H)
    ptsAug = _augment(sel(ft0, ftselect), feats, topk, W, H)
    pts = np.concatenate((pts, ptsAug))
    ptt = np.concatenate((ptt, pttAug))
    xs = (np.random.rand(n_random_draw) * W).astype('int').clip(0, W - 2)
    ys = (np.random.rand(n_random_draw) * H).astype('int').clip(0, H - 2)
    ptsrnd = np.stack((xs, ys), 1)
    ptsrnd = ptsrnd[~inside(ptsrnd)]
    fs0 = interpolate(feats, _v(nrm(ptsrnd)))
    fsselect = np.random.choice(range(ptsrnd.shape[0]), min(n_random_keep, ptsrnd.shape[0]))
    pttAug = _augment(sel(fs0, fsselect), featt, topk, W, H)
    pts = np.concatenate((pts, ptsrnd[fsselect]))
    ptt = np.concatenate((ptt, pttAug))
    return _finish(pts, ptt, inside, W, H, marker)


def getKeypoint(rs, rt, feats, featt):
    """rputil.py:141-237, same signature and return value: rs / rt uint8 BGR panoramas [h, 4h, 3], feats / featt torch [32, h, 4h]
    -> (pts, ptsNorm, ptsW, ptt, pttNorm, pttW) or 6 x None when a view has no detections.  SIFT runs on the observed face only
    (columns [h, 2h)); np.random is drawn in the reference's order (seed it for reproducible keypoints)."""
    H = feats.shape[1]
    W = feats.shape[2]
    pts = _detect(np.ascontiguousarray(bgr2gray(rs)[:, H:2 * H]))
    if not len(pts):
        return None, None, None, None, None, None
    pts[:, 0] += H
    ptt = _detect(np.ascontiguousarray(bgr2gray(rt)[:, H:2 * H]))
    if not len(ptt):
        return None, None, None, None, None, None
    ptt[:, 0] += H
    inside = lambda p: (p[:, 0] >= H) * (p[:, 0] <= H * 2)
    return _keypoints_common(pts, ptt, feats, featt, inside, n_match=30, n_random_draw=30, n_random_keep=30)


def getKeypoint_kinect(rs, rt, feats, featt, rs_full, rt_full):
    """rputil.py:240-353, same signature: SIFT on the full 640x480 kinect frames rs_full / rt_full, detections mapped into the 88x66
    observed crop of the panorama, 300 of them drawn (with replacement, like the reference's np.random.choice), then the common
    assembly with 120 random draws / 100 kept."""
    H = feats.shape[1]
    W = feats.shape[2]
    KW, KH, FW, FH = 640, 480, 88, 66
    x0, y0 = H + H // 2 - FW // 2, H // 2 - FH // 2

    def detect(full):
        p = _detect(bgr2gray(full))
        if len(p):
            p[:, 0] = p[:, 0] / KW * FW + x0
            p[:, 1] = p[:, 1] / KH * FH + y0
        return p
    pts = detect(rs_full)
    if not len(pts):
        return None, None, None, None, None, None
    ptt = detect(rt_full)
    if not len(ptt):
        return None, None, None, None, None, None
    pts = pts[np.random.choice(range(len(pts)), 300), :]
    ptt = ptt[np.random.choice(range(len(ptt)), 300), :]
    inside = lambda p: ((p[:, 0] >= x0) * (p[:, 0] <= H + H // 2 + FW // 2) * (p[:, 1] >= y0) * (p[:, 1] <= H // 2 + FH // 2))
    return _keypoints_common(pts, ptt, feats, featt, inside, n_match=30, n_random_draw=120, n_random_keep=100)


# ---- the batched, device-resident form (RelativePosePipeline(keypoints="reference")) ------------------------------------------------------
KINECT = dict(KW=640, KH=480, FW=88, FH=66, N_SIFT=300)


def map_detections(det, kind, H):
    """Detector coordinates -> panorama coordinates: 'second' = the observed face [H, 2H) (rputil.py:163, :172), 'kinect' = the 640x480
    frame scaled into the 88x66 observed crop (:262-265)."""
    p = np.array(det, dtype=np.float64, copy=True).reshape(-1, 2)
    if kind == "second":
        p[:, 0] += H
    else:
        K = KINECT
        p[:, 0] = p[:, 0] / K["KW"] * K["FW"] + (H + H // 2 - K["FW"] // 2)
        p[:, 1] = p[:, 1] / K["KH"] * K["FH"] + (H // 2 - K["FH"] // 2)
    return p


def _inside(kind, H):
    if kind == "second":
        return lambda p: (p[:, 0] >= H) * (p[:, 0] <= H * 2)
    FW, FH = KINECT["FW"], KINECT["FH"]
    x0, y0 = H + H // 2 - FW // 2, H // 2 - FH // 2
    return lambda p: ((p[:, 0] >= x0) * (p[:, 0] <= H + H // 2 + FW // 2) * (p[:, 1] >= y0) * (p[:, 1] <= H // 2 + FH // 2))


MAX_QUERIES_PER_VIEW = 1000        # include/relpose.h RELPOSE_KP_MAX_QUERIES_PER_VIEW (tests/test_cabi_cpu.py checks that they agree)


def keypoint_plan(pts, ptt, kind, H, W, rng, n_match=30, topk=2):
    """The feature-INDEPENDENT half of getKeypoint ('second', rputil.py:141-237) / getKeypoint_kinect ('kinect', :240-353) for one scan
    pair and one recurrent level: every np.random call of the reference in its order (kinect: choice, choice for the 300 SIFT samples; then
    choice, choice, rand, rand, choice), drawn from `rng` (np.random.RandomState(seed) yields the stream np.random.seed(seed) would).
    pts / ptt: the views' SIFT detections in panorama coordinates (map_detections).  Returns the query points (pixel coordinates) whose
    descriptors are matched against the OTHER view's feature map and the host-known keypoints:
      q1 (source detections -> target map), q2 (target detections -> source map), q3 (source random points -> target map),
      src_a = source detections, src_b = the kept random points, tgt_a = target detections
    The reference's keypoint lists are then  source = [src_a, picks(q2), src_b],  target = [tgt_a, picks(q1), picks(q3)]  with `topk` picks
    per query in query order, picks on the last row / column dropped."""
    pts, ptt = np.asarray(pts, dtype=np.float64).reshape(-1, 2), np.asarray(ptt, dtype=np.float64).reshape(-1, 2)
    if not len(pts) or not len(ptt):
        # A view without SIFT detections: the reference's getKeypoint returns None BEFORE drawing any random number (rputil.py:156-166), the level's
        # pose becomes the identity and the loop goes on (rpmodule.py:522-523, evaluation.py:280-282).  The batched form of that: an EMPTY plan --
        # no queries, no slots -- so both views of the pair get 0 keypoints at this level and the matcher takes its "return identity" status
        # (RELPOSE_FEW_KEYPOINTS); the other pairs of the batch are untouched (raising here aborted the whole batch and, sharded,
        # left the other ranks waiting in the pose all_gather).
        e = np.zeros((0, 2))
        return {"q1": e, "q2": e, "q3": e, "src_a": e, "src_b": e, "tgt_a": e, "topk": topk, "empty": True}
    if kind == "kinect":
        pts = pts[rng.choice(range(len(pts)), KINECT["N_SIFT"]), :]
        ptt = ptt[rng.choice(range(len(ptt)), KINECT["N_SIFT"]), :]
        n_draw, n_keep = 120, 100
    else:
        n_draw, n_keep = 30, 30
    inside = _inside(kind, H)
    fsselect = rng.choice(range(pts.shape[0]), min(n_match, pts.shape[0]))
    ftselect = rng.choice(range(ptt.shape[0]), min(n_match, ptt.shape[0]))
    xs = (rng.rand(n_draw) * W).astype('int').clip(0, W - 2)
    ys = (rng.rand(n_draw) * H).astype('int').clip(0, H - 2)
    ptsrnd = np.stack((xs, ys), 1)
    ptsrnd = ptsrnd[~inside(ptsrnd)]
    sel3 = rng.choice(range(ptsrnd.shape[0]), min(n_keep, ptsrnd.shape[0])) if ptsrnd.shape[0] else np.zeros(0, dtype=np.int64)
    rnd = ptsrnd[sel3].astype('float').reshape(-1, 2)
    return {"q1": pts[fsselect], "q2": ptt[ftselect], "q3": rnd, "src_a": pts, "src_b": rnd, "tgt_a": ptt, "topk": topk}


def keypoint_tables(plans, H, W):
    """The plans of the B scan pairs of a batch (one recurrent level) -> the host arrays of relpose_keypoints_reference: queries grouped by
    the view whose feature map they search (view 2b = source of pair b, 2b + 1 = target), slot tables in the reference's concatenation
    order.  Returns a dict of numpy arrays + sizes."""
    B = len(plans)
    topk = plans[0]["topk"]
    nrm = lambda p: (np.asarray(p, dtype=np.float64).reshape(-1, 2) / np.array([W, H], dtype=np.float64)).astype(np.float32)   # torch_op.v(ptsNorm): float32
    q_src, q_pt, q_map, q_off, slots = [], [], [], [0], []
    nq = 0
    for b, P in enumerate(plans):
        n1, n2, n3 = len(P["q1"]), len(P["q2"]), len(P["q3"])
        # view 2b (source map): q2 (sampled on the target image 2b+1)
        first_q2 = nq
        q_src += [2 * b + 1] * n2; q_map += [2 * b] * n2; q_pt.append(nrm(P["q2"])); nq += n2
        q_off.append(nq)
        # view 2b+1 (target map): q1 then q3 (sampled on the source image 2b)
        first_q1 = nq
        q_src += [2 * b] * (n1 + n3); q_map += [2 * b + 1] * (n1 + n3); q_pt.append(nrm(P["q1"])); q_pt.append(nrm(P["q3"])); nq += n1 + n3
        first_q3 = first_q1 + n1
        q_off.append(nq)
        picks = lambda first, n: [(-3, first * topk + i) for i in range(n * topk)]
        host = lambda a: [(-2, tuple(p)) for p in np.asarray(a, dtype=np.float64).reshape(-1, 2)]
        slots.append(host(P["src_a"]) + picks(first_q2, n2) + host(P["src_b"]))
        slots.append(host(P["tgt_a"]) + picks(first_q1, n1) + picks(first_q3, n3))
    L = max(1, max(len(s_) for s_ in slots))          # (a batch of empty plans still needs one -- empty -- slot column)
    kind = np.full((2 * B, L), -1, dtype=np.int32)
    xy = np.zeros((2 * B, L, 2), dtype=np.float64)
    for v, sl in enumerate(slots):
        for i, (k, val) in enumerate(sl):
            if k == -2:
                kind[v, i] = -2; xy[v, i] = val
            else:
                kind[v, i] = val
    q_off = np.asarray(q_off, dtype=np.int32)
    if int(np.diff(q_off).max()) > MAX_QUERIES_PER_VIEW:
        raise RuntimeError(f"keypoint_tables: {int(np.diff(q_off).max())} queries search one view's feature map; the library takes {MAX_QUERIES_PER_VIEW} "
                           "(include/relpose.h: RELPOSE_KP_MAX_QUERIES_PER_VIEW)")
    return {"q_src": np.asarray(q_src, dtype=np.int32), "q_pt": np.concatenate(q_pt).astype(np.float32).reshape(-1, 2),
            "q_map": np.asarray(q_map, dtype=np.int32), "q_off": q_off, "nq": int(nq), "nq_view_max": int(np.diff(q_off).max()),
            "slot_kind": kind, "slot_xy": xy, "L": int(L), "topk": int(topk)}


def keypoints_reference_dev(f, feat_off, tab, mask_method, window=15, L=None, workspace=None, observed_only=False):
    """One recurrent level's keypoints of every view on the device: f [2B, C, H, W] the network output (CUDA), `tab` = keypoint_tables(...)
    uploaded (torch tensors on f.device; see upload_keypoint_tables).  Returns (pts [2B, L, 2] f64, weight [2B, L] f64, npts [2B] i32).
    observed_only: keep the weight-1 keypoints only (getMatchingPrimitive(..., doCompletion=0), rpmodule.py:534-537)."""
    import torch
    from .util import MASKS
    _lib.require_gpu()
    n, C, H, W = f.shape
    assert f.is_contiguous() and f.dtype == torch.float32
    Lt = tab["L"]
    L = Lt if L is None else L
    assert L >= Lt and tab["slot_kind"].shape == (n, L)
    dev = f.device
    pts = torch.empty(n, L, 2, dtype=torch.float64, device=dev)
    w = torch.empty(n, L, dtype=torch.float64, device=dev)
    npts = torch.empty(n, dtype=torch.int32, device=dev)
    nbytes = _lib.lib().relpose_keypoints_reference_workspace_bytes(max(tab["nq"], 1), H, W, tab["topk"])
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rc = _lib.lib().relpose_keypoints_reference(_lib.ptr(f), C * H * W, int(feat_off), n, H, W, _lib.ptr(tab["q_src"]), _lib.ptr(tab["q_pt"]),
                                                _lib.ptr(tab["q_map"]), _lib.ptr(tab["q_off"]), tab["nq"], tab["nq_view_max"], tab["topk"], int(window),
                                                _lib.ptr(tab["slot_kind"]), _lib.ptr(tab["slot_xy"]), L, MASKS[mask_method], 1 if observed_only else 0, _lib.ptr(pts), _lib.ptr(w),
                                                _lib.ptr(npts), _lib.ptr(workspace), workspace.numel(), _lib.stream_ptr())
    _lib.check(rc, "relpose_keypoints_reference")
    return pts, w, npts


def upload_keypoint_tables(tab, device, L=None):
    """keypoint_tables(...) -> the same dict with its arrays as torch tensors on `device` (slot tables padded to L columns)."""
    import torch
    out = dict(tab)
    Lt = tab["L"]
    L = Lt if L is None else L
    kind, xy = tab["slot_kind"], tab["slot_xy"]
    if L > Lt:
        kind = np.concatenate((kind, np.full((kind.shape[0], L - Lt), -1, dtype=np.int32)), 1)
        xy = np.concatenate((xy, np.zeros((xy.shape[0], L - Lt, 2))), 1)
    for k, a in (("q_src", tab["q_src"]), ("q_pt", tab["q_pt"]), ("q_map", tab["q_map"]), ("q_off", tab["q_off"]), ("slot_kind", kind), ("slot_xy", xy)):
        out[k] = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    out["L"] = L
    return out
