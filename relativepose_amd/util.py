"""Host-side mirror of the reference's panorama geometry (util.py), backed by
the HIP library.  The *_dev functions take/return CUDA tensors (batched, no host
round trip); the reference-named wrappers keep the reference's numpy signatures.

  apply_mask       util.py:209      warping          util.py:94
  Pano2PointCloud  util.py:751      build_view       evaluation.py:217-230
  sample_primitives  evaluation.py:246-253 + rpmodule.getMatchingPrimitive :526-532
"""
import numpy as np

from . import _lib

DATASETS = {"suncg": 0, "matterport": 1, "scannet": 2}
MASKS = {"second": 0, "kinect": 1}


def dataset_id(name):
    for k, v in DATASETS.items():
        if k in name:
            return v
    raise ValueError(f"unknown dataset {name}")


def build_view_dev(rgb, norm, depth, mask_method):
    """rgb/norm [n,3,h,4h], depth [n,h,4h] f32 CUDA -> view [n,8,h,4h]."""
    import torch
    _lib.require_gpu()
    n, _, h, w = rgb.shape
    assert w == 4 * h
    view = torch.empty(n, 8, h, w, dtype=torch.float32, device=rgb.device)
    rc = _lib.lib().relpose_build_view(_lib.ptr(rgb.contiguous()), _lib.ptr(norm.contiguous()), _lib.ptr(depth.contiguous()),
                                       _lib.ptr(view), n, h, MASKS[mask_method], _lib.stream_ptr())
    _lib.check(rc, "relpose_build_view")
    return view


def apply_mask_dev(x, mask_method):
    import torch
    _lib.require_gpu()
    n, c, h, w = x.shape
    assert w == 4 * h and x.is_contiguous() and x.dtype == torch.float32
    mask = torch.empty(n, 1, h, w, dtype=torch.float32, device=x.device)
    rc = _lib.lib().relpose_apply_mask(_lib.ptr(x), _lib.ptr(mask), n, c, h, MASKS[mask_method], _lib.stream_ptr())
    _lib.check(rc, "relpose_apply_mask")
    return x, mask


_warp_ws = {}


def warping_dev(view, pose, dataset, out=None):
    """view [n,8,h,4h] f32, pose [n,4,4] f64 (CUDA) -> warped view [n,8,h,4h] f32."""
    import torch
    _lib.require_gpu()
    n, c, h, w = view.shape
    assert c == 8 and w == 4 * h and view.is_contiguous() and pose.is_contiguous() and pose.dtype == torch.float64
    L = _lib.lib()
    nbytes = L.relpose_warp_workspace_bytes(n, h)
    key = (view.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _warp_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=view.device)       # (zeros: every warp call leaves its keys reset, see warp_pairs_dev)
        _warp_ws[key] = ws
    if out is None:
        out = torch.empty_like(view)
    rc = L.relpose_warp(_lib.ptr(view), _lib.ptr(pose), _lib.ptr(out), _lib.ptr(ws), n, h, dataset_id(dataset), _lib.stream_ptr())
    _lib.check(rc, "relpose_warp")
    return out


def warp_pairs_dev(x, pose, dataset):
    """In place on the network input x [n,16,h,4h]: x[i, 8:16] = warping(x[i^1, 0:8], pose[i]) (relpose_warp_pairs)."""
    import torch
    _lib.require_gpu()
    n, c, h, w = x.shape
    assert c == 16 and w == 4 * h and n % 2 == 0 and x.is_contiguous() and pose.is_contiguous() and pose.dtype == torch.float64
    L = _lib.lib()
    nbytes = L.relpose_warp_workspace_bytes(n, h)
    key = (x.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _warp_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=x.device)
        _warp_ws[key] = ws
    # the workspace was zero-initialised and every call resets the keys it consumed: no memset in front of the scatter pass
    rc = L.relpose_warp_pairs2(_lib.ptr(x), _lib.ptr(pose), _lib.ptr(ws), n, h, dataset_id(dataset), 1, _lib.stream_ptr())
    _lib.check(rc, "relpose_warp_pairs2")
    return x


def pose_inverse_dev(pose):
    import torch
    out = torch.empty_like(pose)
    rc = _lib.lib().relpose_pose_inverse(_lib.ptr(pose), _lib.ptr(out), pose.shape[0], _lib.stream_ptr())
    _lib.check(rc, "relpose_pose_inverse")
    return out


def pano2pc_dev(depth, dataset):
    """depth [n,h,4h] f32 -> (pc [n,3,4hh] f64, valid [n,4hh] uint8)."""
    import torch
    _lib.require_gpu()
    n, h, w = depth.shape
    pc = torch.empty(n, 3, h * w, dtype=torch.float64, device=depth.device)
    valid = torch.empty(n, h * w, dtype=torch.uint8, device=depth.device)
    rc = _lib.lib().relpose_pano2pc(_lib.ptr(depth.contiguous()), _lib.ptr(pc), _lib.ptr(valid), n, h, dataset_id(dataset),
                                    _lib.stream_ptr())
    _lib.check(rc, "relpose_pano2pc")
    return pc, valid


def sample_primitives_dev(f, feat_off, obs_norm, obs_depth, pts, npts, mask_method, dataset, compose=0):
    """f [n,cf,h,4h] net output, obs_norm [n,3,h,4h], obs_depth [n,h,4h] f32, pts [n,N,2] f64, npts [n] i32
    -> pc [n,N,3] f64, normal [n,N,3] f64, feat [n,N,32] f32.  compose: 0 = evaluation.py:250-251, 1 = rpmodule.py:633-634."""
    import torch
    _lib.require_gpu()
    n, cf, h, w = f.shape
    N = pts.shape[1]
    dev = f.device
    pc = torch.zeros(n, N, 3, dtype=torch.float64, device=dev)
    nn = torch.zeros(n, N, 3, dtype=torch.float64, device=dev)
    ft = torch.zeros(n, N, 32, dtype=torch.float32, device=dev)
    for t in (f, obs_norm, obs_depth, pts, npts):
        assert t.is_contiguous() and t.is_cuda
    rc = _lib.lib().relpose_sample_primitives(_lib.ptr(f), cf, feat_off, _lib.ptr(obs_norm), _lib.ptr(obs_depth), _lib.ptr(pts),
                                              _lib.ptr(npts), N, _lib.ptr(pc), _lib.ptr(nn), _lib.ptr(ft), n, h,
                                              MASKS[mask_method], int(compose), dataset_id(dataset), _lib.stream_ptr())
    _lib.check(rc, "relpose_sample_primitives")
    return pc, nn, ft


# ---- reference-named numpy wrappers -------------------------------------------------------------

def apply_mask(x, maskMethod, *arg):
    """util.py:209: x torch tensor [n,c,h,w] -> (x*mask, mask, geow).  geow (a training-only
    geometric weight) is not computed on the inference path and is returned as None."""
    import torch
    dev = _lib.require_gpu()
    xd = x.to(dev, torch.float32).contiguous().clone()
    xd, m = apply_mask_dev(xd, maskMethod)
    return xd, m, None


def warping(view, R, dataList):
    """util.py:94: view numpy [1,8,h,4h], R [4,4] -> numpy [1,8,h,4h] (float32 values; zeros for identity)."""
    import torch
    dev = _lib.require_gpu()
    v = torch.from_numpy(np.ascontiguousarray(view, dtype=np.float32)).to(dev)
    T = torch.from_numpy(np.ascontiguousarray(R, dtype=np.float64)[None]).to(dev)
    return warping_dev(v, T, dataList).cpu().numpy()


def Pano2PointCloud(depth, dataList):
    """util.py:751: depth numpy [h,4h] -> [3,n] float64 (scannet: zero-depth points dropped)."""
    import torch
    dev = _lib.require_gpu()
    d = torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float32)[None]).to(dev)
    pc, valid = pano2pc_dev(d, dataList)
    pc, valid = pc[0].cpu().numpy(), valid[0].cpu().numpy().astype(bool)
    return pc[:, valid]


# ---- evaluation-side statistics ---------------------------------------------------------------

def depth2pc_dev(depth, dataset):
    """depth [n,h,4h] f32 CUDA -> (pc [n,P,3] f64, valid [n,P] uint8) of the observed block (util.depth2pc)."""
    import torch
    _lib.require_gpu()
    n, h, w = depth.shape
    L = _lib.lib()
    P = L.relpose_observed_points(h, dataset_id(dataset))
    pc = torch.empty(n, P, 3, dtype=torch.float64, device=depth.device)
    valid = torch.empty(n, P, dtype=torch.uint8, device=depth.device)
    _lib.check(L.relpose_depth2pc(_lib.ptr(depth.contiguous()), _lib.ptr(pc), _lib.ptr(valid), n, h, dataset_id(dataset), _lib.stream_ptr()),
               "relpose_depth2pc")
    return pc, valid


def nn_dist_dev(query, ref, pose=None, query_valid=None, ref_valid=None):
    """min distance of every (pose-moved) query point [nq,3] to the reference set [nr,3] (f64 CUDA tensors)."""
    import torch
    out = torch.empty(query.shape[0], dtype=torch.float64, device=query.device)
    rc = _lib.lib().relpose_nn_dist(_lib.ptr(query.contiguous()), _lib.ptr(query_valid), query.shape[0], _lib.ptr(ref.contiguous()),
                                    _lib.ptr(ref_valid), ref.shape[0], _lib.ptr(pose), _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "relpose_nn_dist")
    return out


def point_cloud_overlap(pc_src, pc_tgt, R_gt_44):
    """util.py:21-40 with the two KDTree queries replaced by brute-force GPU nearest neighbours.
    numpy in, (overlap_val, cam_dist, pc_dist, pc_nn) out."""
    import torch
    dev = _lib.require_gpu()
    ps = torch.from_numpy(np.ascontiguousarray(pc_src, dtype=np.float64)).to(dev)
    pt = torch.from_numpy(np.ascontiguousarray(pc_tgt, dtype=np.float64)).to(dev)
    R = np.ascontiguousarray(R_gt_44, dtype=np.float64)
    d_s2t = nn_dist_dev(ps, pt, torch.from_numpy(R).to(dev)).cpu().numpy()
    d_t2s = nn_dist_dev(pt, ps, torch.from_numpy(np.ascontiguousarray(np.linalg.inv(R))).to(dev)).cpu().numpy()
    ov = max((d_s2t < 0.08).sum() / pc_src.shape[0], (d_t2s < 0.08).sum() / pc_tgt.shape[0])
    src_trans = np.matmul(R[:3, :3], pc_src.T) + R[:3, 3:4]
    return ov, np.linalg.norm(R[:3, 3]), np.linalg.norm(src_trans.mean(1) - pc_tgt.T.mean(1)), (d_s2t.min() + d_t2s.min()) / 2


def depth2pc(depth, dataList):
    """util.py:468: depth numpy = one h x h face (suncg / matterport) or the 66x88 kinect crop (scannet)
    -> (pc [k,3] of the non-zero depths, mask)."""
    import torch
    dev = _lib.require_gpu()
    ds = dataset_id(dataList)
    hh, ww = depth.shape
    if ds == 2 and (hh, ww) == (480, 640):                # the full-resolution kinect image (util.py:497-507: the baselines' clouds)
        dd = torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float32)[None]).to(dev)
        pc = torch.empty(1, hh * ww, 3, dtype=torch.float64, device=dev)
        valid = torch.empty(1, hh * ww, dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().relpose_depth2pc_full(_lib.ptr(dd), _lib.ptr(pc), _lib.ptr(valid), 1, hh, ww, _lib.stream_ptr()), "relpose_depth2pc_full")
        m = valid[0].cpu().numpy().astype(bool)
        return pc[0].cpu().numpy()[m], m
    h = 160 if ds == 2 else hh
    pano = np.zeros((1, h, 4 * h), np.float32)
    if ds == 2:
        assert (hh, ww) == (66, 88), "scannet: the 66x88 crop or the 480x640 image (the reference defines no other shape, util.py:498,508)"
        pano[0, 47:113, 196:284] = depth
    else:
        pano[0, :, h:2 * h] = depth
    pc, valid = depth2pc_dev(torch.from_numpy(pano).to(dev), dataList)
    m = valid[0].cpu().numpy().astype(bool)
    return pc[0].cpu().numpy()[m], m


def parse_data(depth, rgb, norm, dataList, method):
    """util.py:42-92, same signature and 8-tuple: both scans as point clouds (+ colours, normals).  suncg / matterport: the observed face
    [160,320) of depth [1,2,160,640], rgb uint8 [1,2,3,160,640], norm [1,2,3,160,640].  scannet with an 'ours' method: the 66x88 kinect crop
    of the same panoramas, normals renormalised (:56-76); scannet with a baseline method (:78-90): depth [1,2,480,640], rgb [1,2,3,480,640] are
    the full-resolution kinect IMAGES, back-projected whole, and there are no normals (None, None)."""
    full = False
    if 'suncg' in dataList or 'matterport' in dataList:
        ys, xs = slice(None), slice(160, 320)
    elif 'scannet' in dataList:
        if 'ours' in method:
            ys, xs = slice(80 - 33, 80 + 33), slice(160 + 80 - 44, 160 + 80 + 44)
        else:
            ys, xs, full = slice(None), slice(None), True
    else:
        raise ValueError(f"unknown dataset {dataList}")
    out = {}
    for v, tag in ((0, "src"), (1, "tgt")):
        d = depth[0, v, ys, xs]
        col = rgb[0, v, :, ys, xs].transpose(1, 2, 0)
        pc, mask = depth2pc(d, dataList)
        col = col.reshape(-1, 3)[mask] / 255.
        nrm = None
        if not full:
            nrm = norm[0, v, :, ys, xs].copy().transpose(1, 2, 0).reshape(-1, 3)[mask]
            if 'scannet' in dataList:
                with np.errstate(divide="ignore", invalid="ignore"):
                    nrm = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
                nrm[np.isnan(nrm.sum(1))] = 0
        out[tag] = (d, nrm, col, pc)
    s, t = out["src"], out["tgt"]
    return s[0], t[0], s[1], t[1], s[2], t[2], s[3], t[3]


def angular_distance_np(R_hat, R):
    """util.py:176-187 (lives in relativepose_amd.evaluation; re-exported under the reference's module name)."""
    from .evaluation import angular_distance_np as f
    return f(R_hat, R)
