"""Host-side mirror of the reference's RPModule/rputil.py, backed by the HIP library:

  opts            rputil.py:11-22      (re-exported from rpmodule)
  interpolate     rputil.py:43-58      bilinear descriptor sampling
  getPixel        rputil.py:88-119     bilinear depth / normal sampling + unprojection
  Sampling        rputil.py:355-371    NMS keypoint sampling on distance maps
  feature_distance_map_dev             the descriptor-to-map distance of getKeypoint :182-190

SIFT detection (getKeypoint's cv2 part) is not built: see rpmodule.set_keypoint_provider."""
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
