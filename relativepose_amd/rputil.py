"""Host-side mirror of the reference's RPModule/rputil.py, backed by the HIP library:

  opts            rputil.py:11-22      (re-exported from rpmodule)
  interpolate     rputil.py:43-58      bilinear descriptor sampling
  getPixel        rputil.py:88-119     bilinear depth / normal sampling + unprojection
  Sampling        rputil.py:355-371    NMS keypoint sampling on distance maps
  feature_distance_map_dev             the descriptor-to-map distance of getKeypoint :182-190

  getKeypoint / getKeypoint_kinect  rputil.py:141-353  everything AROUND the SIFT detector: descriptor sampling at the detected points,
                                   feature-guided augmentation (distance maps + NMS on the GPU), random fill, weights

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
