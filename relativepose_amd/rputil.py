"""Host-side mirror of the keypoint-augmentation pieces of the reference's rputil.py that are
deterministic functions: the descriptor-to-map distance of getKeypoint (:182-190) and
``Sampling`` (:355-371).  SIFT detection and the random selection around them stay with the caller."""
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
