"""Host-side mirror of the reference pose module, backed by the HIP library.

Same names and argument meaning as /root/reference/RPModule/rpmodule.py and
rputil.py for the hot-path entry points:

  opts                               rputil.py:11-22
  RelativePoseEstimation_helper      rpmodule.py:317   (numpy dicts in, 4x4 numpy out)
  getMatchingPrimitive               rpmodule.py:511   (completed-scan dicts -> keypoint primitives)
  RelativePoseEstimation             rpmodule.py:540   (getMatchingPrimitive + helper)
  RelativePoseEstimationViaCompletion rpmodule.py:569  (the recurrent completion / matching loop for one scan pair)
  match_pairs                        batched device API the helper is built on

Keypoints come from rputil.getKeypoint / getKeypoint_kinect (rputil.py:141-353; everything but the cv2 SIFT detector is built, the
detector is a hook: rputil.set_sift_detector) or from a *keypoint provider* with the same signature installed with
``set_keypoint_provider`` -- e.g. ``fixed_keypoints(...)`` for injected points (parity tests, bench).

Degenerate inputs return identity like the reference; status codes say why.
"""
import ctypes as C

import numpy as np

from . import _lib

METHODS = {"irls+sm": 0, "horn87": 1, "irls": 2, "spectral": 3}


class opts:
    """rputil.py:11-22."""

    def __init__(self, sigmaAngle1=0.523 / 2, sigmaAngle2=0.523 / 2, sigmaDist=0.08 / 2, sigmaFeat=0.01):
        self.distThre = 0.08
        self.distSepThre = 1.5 * 0.08
        self.angleThre = 45 / 180. * np.pi
        self.sigmaAngle1 = sigmaAngle1
        self.sigmaAngle2 = sigmaAngle2
        self.sigmaDist = sigmaDist
        self.sigmaFeat = sigmaFeat
        self.mu = 0.3
        self.topK = 5
        self.method = 'irls+sm'


def _c_params(para):
    if para.method not in METHODS:
        raise Exception("unknown method!")          # rpmodule.py:508
    p = _lib.Params()
    for k in ("distThre", "distSepThre", "angleThre", "sigmaAngle1", "sigmaAngle2", "sigmaDist", "sigmaFeat", "mu"):
        setattr(p, k, float(getattr(para, k)))
    p.topK = int(para.topK)
    p.method = METHODS[para.method]
    return p


class MatchResult:
    __slots__ = ("pose", "status", "corres_j", "corres_w", "counts", "trace", "eig_iters", "wij")


_ws_cache = {}


MAX_CORRESPONDENCES = 8192       # include/relpose.h RELPOSE_MAX_CORRESPONDENCES, RELPOSE_MAX_TARGETS (tests/test_cabi_cpu.py checks that they agree)
MAX_TARGETS = 1152


def _workspace(nbytes, dev):
    import torch
    key = (dev.index, torch.cuda.current_stream().cuda_stream)      # streams must not share scratch
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf


def match_pairs(pc_s, n_s, f_s, w_s, pc_t, n_t, f_t, w_t, ns, nt, para, debug=False, want_wij=False, max_edges=0, fit_cluster=0, affinity_kernel=0):
    """Batched matcher on device tensors.
    pc_*/n_* [B,N,3] f64, f_* [B,N,32] f32, w_* [B,N] f64, ns/nt [B] int32 (all CUDA, contiguous).
    fit_cluster / affinity_kernel: the call's own kernel choices (RelposeMatchArgs; 0 = by problem size; same results whatever the value).
    Returns MatchResult with pose [B,4,4] f64 and status [B] int32 (device tensors)."""
    import torch
    dev = _lib.require_gpu()
    L = _lib.lib()
    B, ns_max, nt_max = pc_s.shape[0], pc_s.shape[1], pc_t.shape[1]
    for t, dt in ((pc_s, torch.float64), (n_s, torch.float64), (f_s, torch.float32), (w_s, torch.float64),
                  (pc_t, torch.float64), (n_t, torch.float64), (f_t, torch.float32), (w_t, torch.float64),
                  (ns, torch.int32), (nt, torch.int32)):
        assert t.is_cuda and t.is_contiguous() and t.dtype == dt, (t.dtype, dt, t.is_contiguous())
    assert f_s.shape[2] == 32 and f_t.shape[2] == 32
    p = _c_params(para)
    kp = _lib.Keypoints(B, ns_max, nt_max, _lib.ptr(ns), _lib.ptr(nt), _lib.ptr(pc_s), _lib.ptr(n_s), _lib.ptr(f_s), _lib.ptr(w_s),
                        _lib.ptr(pc_t), _lib.ptr(n_t), _lib.ptr(f_t), _lib.ptr(w_t))
    nbytes = L.relpose_match_workspace_bytes(B, ns_max, nt_max, p.topK, int(max_edges))
    if nbytes == 0:
        if ns_max * p.topK > MAX_CORRESPONDENCES:
            raise RuntimeError(f"relpose_match_pairs: {ns_max} keypoints x topK {p.topK} = {ns_max * p.topK} correspondences per pair exceed the "
                               f"library's limit of {MAX_CORRESPONDENCES} (include/relpose.h: RELPOSE_MAX_CORRESPONDENCES)")
        if nt_max > MAX_TARGETS:
            raise RuntimeError(f"relpose_match_pairs: {nt_max} target keypoints exceed the library's limit of {MAX_TARGETS} "
                               "(include/relpose.h: RELPOSE_MAX_TARGETS)")
        raise RuntimeError("relpose_match_workspace_bytes: invalid shape")
    ws = _workspace(nbytes, dev)
    res = MatchResult()
    res.pose = torch.empty(B, 4, 4, dtype=torch.float64, device=dev)
    res.status = torch.empty(B, dtype=torch.int32, device=dev)
    res.corres_j = res.corres_w = res.counts = res.trace = res.eig_iters = res.wij = None
    dbg = None
    if debug or want_wij:
        dbg = _lib.MatchDebug()
        if want_wij:
            res.wij = torch.zeros(B, ns_max, nt_max, dtype=torch.float32, device=dev)
        if debug:
            res.corres_j = torch.zeros(B, ns_max, p.topK, dtype=torch.int32, device=dev)
            res.corres_w = torch.zeros(B, ns_max, p.topK, dtype=torch.float64, device=dev)
            res.counts = torch.zeros(B, 4, dtype=torch.int32, device=dev)
            res.trace = torch.zeros(B, 6, 4, 4, dtype=torch.float64, device=dev)
            res.eig_iters = torch.zeros(B, 5, dtype=torch.int32, device=dev)
        dbg.wij, dbg.corres_j, dbg.corres_w = _lib.ptr(res.wij), _lib.ptr(res.corres_j), _lib.ptr(res.corres_w)
        dbg.counts, dbg.trace, dbg.eig_iters = _lib.ptr(res.counts), _lib.ptr(res.trace), _lib.ptr(res.eig_iters)
    a = _lib.MatchArgs()
    a.struct_size = C.sizeof(_lib.MatchArgs)
    a.fit_cluster = int(fit_cluster)
    a.affinity_kernel = _lib.AFFINITY_KERNELS.get(affinity_kernel, affinity_kernel) if isinstance(affinity_kernel, str) else int(affinity_kernel)
    a.params_host, a.kp_host = C.pointer(p), C.pointer(kp)
    a.workspace, a.workspace_bytes, a.max_edges = _lib.ptr(ws), ws.numel(), int(max_edges)
    a.pose, a.status = _lib.ptr(res.pose), _lib.ptr(res.status)
    a.debug_host = C.pointer(dbg) if dbg is not None else None
    a.stream = _lib.stream_ptr()
    _lib.check(L.relpose_match_pairs_ex(C.byref(a)), "relpose_match_pairs_ex")
    return res


def affinity_topk_buffers(B, ns_max, nt_max, topK, dev, want_wij=True):
    """Output buffers of affinity_topk (zero-filled: padded rows / columns are never written)."""
    import torch
    wij = torch.zeros(B, ns_max, nt_max, dtype=torch.float32, device=dev) if want_wij else None
    cj = torch.zeros(B, ns_max, topK, dtype=torch.int32, device=dev)
    cw = torch.zeros(B, ns_max, topK, dtype=torch.float64, device=dev)
    keff = torch.zeros(B, dtype=torch.int32, device=dev)
    return wij, cj, cw, keff, torch.zeros(1, dtype=torch.float64, device=dev)


def affinity_topk(f_s, w_s, f_t, w_t, ns, nt, para, want_wij=True, out=None):
    """Stage A+B only (rpmodule.py:342-379): returns (wij or None, corres_j, corres_w, k_eff).
    out = affinity_topk_buffers(...) reuses output buffers (back-to-back launches without allocator calls)."""
    dev = _lib.require_gpu()
    L = _lib.lib()
    B, ns_max, nt_max = f_s.shape[0], f_s.shape[1], f_t.shape[1]
    p = _c_params(para)
    wij, cj, cw, keff, dummy = out if out is not None else affinity_topk_buffers(B, ns_max, nt_max, p.topK, dev, want_wij)
    kp = _lib.Keypoints(B, ns_max, nt_max, _lib.ptr(ns), _lib.ptr(nt), _lib.ptr(dummy), _lib.ptr(dummy), _lib.ptr(f_s), _lib.ptr(w_s),
                        _lib.ptr(dummy), _lib.ptr(dummy), _lib.ptr(f_t), _lib.ptr(w_t))
    rc = L.relpose_affinity_topk(C.byref(p), C.byref(kp), _lib.ptr(wij), _lib.ptr(cj), _lib.ptr(cw), _lib.ptr(keff), _lib.stream_ptr())
    _lib.check(rc, "relpose_affinity_topk")
    return wij, cj, cw, keff


def pack_keypoints(cases, dev):
    """List of (dataS, dataT) helper dicts -> padded device tensors."""
    import torch
    B = len(cases)
    ns = [c[0]['pc'].shape[0] for c in cases]
    nt = [c[1]['pc'].shape[0] for c in cases]
    nsm, ntm = max(max(ns), 1), max(max(nt), 1)

    def pad(key, side, n, w, dt):
        out = np.zeros((B, n) + w, dtype=dt)
        for b, c in enumerate(cases):
            a = np.asarray(c[side][key], dtype=dt)
            if a.shape[0]:
                out[b, :a.shape[0]] = a.reshape((a.shape[0],) + w)
        return torch.from_numpy(out).to(dev)

    args = [pad('pc', 0, nsm, (3,), np.float64), pad('normal', 0, nsm, (3,), np.float64), pad('feat', 0, nsm, (32,), np.float32),
            pad('weight', 0, nsm, (), np.float64),
            pad('pc', 1, ntm, (3,), np.float64), pad('normal', 1, ntm, (3,), np.float64), pad('feat', 1, ntm, (32,), np.float32),
            pad('weight', 1, ntm, (), np.float64),
            torch.tensor(ns, dtype=torch.int32, device=dev), torch.tensor(nt, dtype=torch.int32, device=dev)]
    return args


def RelativePoseEstimation_helper(dataS, dataT, para):
    """Drop-in for rpmodule.py:317: dict keys 'pc'[k,3] 'normal'[k,3] 'feat'[k,32] 'weight'[k]."""
    dev = _lib.require_gpu()
    res = match_pairs(*pack_keypoints([(dataS, dataT)], dev), para)
    return res.pose[0].cpu().numpy()


# ---- keypoint provider hook -------------------------------------------------------------------------------------
_keypoint_provider = None


def set_keypoint_provider(fn):
    """fn(rgbS, rgbT, featS, featT[, rgb_fullS, rgb_fullT]) -> (pts, ptsNorm, ptsW, ptt, pttNorm, pttW): the signature
    of the reference's rputil.getKeypoint (:141) / getKeypoint_kinect (:240).  pts [k,2] pixel coords (x,y),
    ptsNorm = pts / (W,H), ptsW in {1, .99}.  Returns the previous provider."""
    global _keypoint_provider
    old, _keypoint_provider = _keypoint_provider, fn
    return old


def fixed_keypoints(pts, ptsW, ptt, pttW):
    """Provider that injects given keypoints (what the parity tests and the bench use)."""
    pts, ptt = np.asarray(pts, dtype=np.float64), np.asarray(ptt, dtype=np.float64)

    def provider(rgbS, rgbT, featS, featT, *rest):
        H, W = featS.shape[1], featS.shape[2]
        return pts, pts / np.array([W, H], dtype=np.float64), np.asarray(ptsW), ptt, ptt / np.array([W, H], dtype=np.float64), np.asarray(pttW)
    return provider


def getKeypoint(rgbS, rgbT, featS, featT, *rest):
    """rputil.getKeypoint (rputil.py:141-237) -- or the installed provider (set_keypoint_provider; the parity tests and the bench inject
    fixed keypoints).  Without a provider the reference-named assembly in rputil runs: SIFT through rputil.set_sift_detector / cv2,
    everything else on the GPU."""
    if _keypoint_provider is not None:
        return _keypoint_provider(rgbS, rgbT, featS, featT, *rest)
    from . import rputil
    return rputil.getKeypoint(rgbS, rgbT, featS, featT)


def getKeypoint_kinect(rgbS, rgbT, featS, featT, rgbS_full=None, rgbT_full=None):
    """rputil.getKeypoint_kinect (rputil.py:240-353), or the installed provider."""
    if _keypoint_provider is not None:
        return _keypoint_provider(rgbS, rgbT, featS, featT, rgbS_full, rgbT_full)
    from . import rputil
    return rputil.getKeypoint_kinect(rgbS, rgbT, featS, featT, rgbS_full, rgbT_full)


def getMatchingPrimitive(dataS, dataT, dataset, representation, doCompletion):
    """rpmodule.py:511-538, same signature and return value: dataS/dataT hold 'rgb' [h,4h,3] u8, 'normal' [h,4h,3],
    'depth' [h,4h] (numpy) and 'feat' [32,h,4h] (torch) of the completed scans; returns
    (pts3d [3,k], ptt3d [3,k'], ptsns [k,3], ptsnt [k',3], dess [k,32], dest [k',32], ptsW, pttW) or 8 x None."""
    import torch
    from . import rputil
    if 'suncg' in dataset or 'matterport' in dataset:
        kp = getKeypoint(dataS['rgb'], dataT['rgb'], dataS['feat'], dataT['feat'])
    elif 'scannet' in dataset:
        kp = getKeypoint_kinect(dataS['rgb'], dataT['rgb'], dataS['feat'], dataT['feat'], dataS.get('rgb_full'), dataT.get('rgb_full'))
    else:
        raise ValueError(f"unknown dataset {dataset}")
    pts, ptsNorm, ptsW, ptt, pttNorm, pttW = kp
    if pts is None or ptt is None or pts.shape[1] < 2 or ptt.shape[1] < 2:          # rpmodule.py:522-523
        return None, None, None, None, None, None, None, None
    pts3d, ptsns = rputil.getPixel(dataS['depth'], dataS['normal'], pts, dataset=dataset, representation=representation)
    ptt3d, ptsnt = rputil.getPixel(dataT['depth'], dataT['normal'], ptt, dataset=dataset, representation=representation)
    v = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()                  # torch_op.v: float32
    dess = rputil.interpolate(dataS['feat'], v(ptsNorm)).cpu().numpy().T
    dest = rputil.interpolate(dataT['feat'], v(pttNorm)).cpu().numpy().T
    ptsW, pttW = np.asarray(ptsW), np.asarray(pttW)
    if not doCompletion:                                                             # rpmodule.py:534-537
        pts3d, ptsns, dess, ptsW = pts3d[:, ptsW == 1], ptsns[ptsW == 1], dess[ptsW == 1], ptsW[ptsW == 1]
        ptt3d, ptsnt, dest, pttW = ptt3d[:, pttW == 1], ptsnt[pttW == 1], dest[pttW == 1], pttW[pttW == 1]
    return pts3d, ptt3d, ptsns, ptsnt, dess, dest, ptsW, pttW


def RelativePoseEstimation(dataS, dataT, para, dataset, representation, maskMethod, doCompletion=True, index=None):
    """rpmodule.py:540-567, same signature: keypoint primitives of two completed scans -> 4x4 pose (identity when
    too few keypoints are found)."""
    R_hat = np.eye(4)
    pts3d, ptt3d, ptsns, ptsnt, dess, dest, ptsW, pttW = getMatchingPrimitive(dataS, dataT, dataset, representation, doCompletion)
    if pts3d is None or ptt3d is None or pts3d.shape[0] < 2:                         # rpmodule.py:557 (sic: tests shape[0] == 3)
        return R_hat
    return RelativePoseEstimation_helper({'pc': pts3d.T, 'normal': ptsns, 'feat': dess, 'weight': ptsW},
                                         {'pc': ptt3d.T, 'normal': ptsnt, 'feat': dest, 'weight': pttW}, para)


def RelativePoseEstimationViaCompletion(net, data_s, data_t, args):
    """rpmodule.py:569-662, same signature: alternate scan completion and pairwise matching for ONE scan pair.
    data_s/data_t: 'rgb' [h,4h,3], 'norm' [h,4h,3], 'depth' [h,4h] numpy (HWC like the reference).  args needs
    snumclass, featureDim, outputType, maskMethod, alterStep, dataset, para (an opts whose four sigmas are per-step
    sequences), representation, completion.  The loop stays on the device: warp -> SCNet -> compose + sample -> match;
    only the keypoint provider sees host data (rgb u8 + the feature maps as CUDA tensors), once per alternation like
    the reference's getKeypoint call.  Output composition follows THIS function (normal / (|normal| + 1e-12),
    :633-634), not evaluation.py's variant."""
    import copy

    import torch
    from . import util
    dev = _lib.require_gpu()
    # rpmodule.py:583-593: the feature block starts behind the heads that exist.  The loop itself reads channels 3:6 as the completed
    # normal and 6 as the depth (:625-630) and needs idx_f_end (set only with 'f', :592), so the reference runs with rgb, n, d and f present,
    # with or without the semantic head: 'rgbdnsf' (evaluation.py) or 'rgbdnf'
    for head in ('rgb', 'n', 'd', 'f'):
        if head not in args.outputType:
            raise ValueError(f"outputType {args.outputType!r}: RelativePoseEstimationViaCompletion reads the {head!r} head (rpmodule.py:592,625-630)")
    args.idx_f_start = 3 + 3 + 1 + (args.snumclass if 's' in args.outputType else 0)
    args.idx_f_end = args.idx_f_start + args.featureDim
    assert args.featureDim == 32
    h = data_s['depth'].shape[0]
    chw = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32).transpose(2, 0, 1)))
    rgb = torch.stack((chw(data_s['rgb']), chw(data_t['rgb']))).to(dev)
    nrm = torch.stack((chw(data_s['norm']), chw(data_t['norm']))).to(dev)
    dep = torch.from_numpy(np.stack((np.asarray(data_s['depth'], dtype=np.float32), np.asarray(data_t['depth'], dtype=np.float32)))).to(dev)
    view = util.build_view_dev(rgb, nrm, dep, args.maskMethod)                       # apply_mask + valid channel (:603-613)
    x = torch.empty(2, 16, h, 4 * h, dtype=torch.float32, device=dev)
    x[:, :8].copy_(view)
    m_dev = util.apply_mask_dev(torch.ones(2, 1, h, 4 * h, dtype=torch.float32, device=dev), args.maskMethod)[1]
    m_np = m_dev[:, 0].cpu().numpy()[..., None]
    rgb_u8 = [(m_np[v] * np.asarray(d['rgb']) * 255).astype('uint8') for v, d in enumerate((data_s, data_t))]     # :640-641
    R_hat = np.eye(4)
    for alter_ in range(args.alterStep):
        R_dev = torch.from_numpy(np.ascontiguousarray(R_hat, dtype=np.float64)[None]).to(dev)
        poses = torch.cat((util.pose_inverse_dev(R_dev), R_dev)).contiguous()       # image 0 gets t2s (inv R), image 1 gets s2t (R)
        util.warp_pairs_dev(x, poses, args.dataset)
        f = net(x)
        feats = [f[v, args.idx_f_start:args.idx_f_end] for v in range(2)]
        if 'scannet' in args.dataset:
            kp = getKeypoint_kinect(rgb_u8[0], rgb_u8[1], feats[0], feats[1],
                                    (np.asarray(data_s['rgb_full']) * 255).astype('uint8') if 'rgb_full' in data_s else None,
                                    (np.asarray(data_t['rgb_full']) * 255).astype('uint8') if 'rgb_full' in data_t else None)
        else:
            kp = getKeypoint(rgb_u8[0], rgb_u8[1], feats[0], feats[1])
        pts, _, ptsW, ptt, _, pttW = kp
        if pts is None or ptt is None or pts.shape[1] < 2 or ptt.shape[1] < 2:
            R_hat = np.eye(4)                                                        # RelativePoseEstimation's early return (:557-559)
            continue
        ptsW, pttW = np.asarray(ptsW, dtype=np.float64), np.asarray(pttW, dtype=np.float64)
        if not args.completion:
            pts, ptsW = pts[ptsW == 1], ptsW[ptsW == 1]
            ptt, pttW = ptt[pttW == 1], pttW[pttW == 1]
        N = max(len(pts), len(ptt), 1)
        P = np.zeros((2, N, 2)); P[0, :len(pts)] = pts; P[1, :len(ptt)] = ptt
        Wt = np.zeros((2, N)); Wt[0, :len(pts)] = ptsW; Wt[1, :len(ptt)] = pttW
        npts = torch.tensor([len(pts), len(ptt)], dtype=torch.int32, device=dev)
        pc, nn, ft = util.sample_primitives_dev(f, args.idx_f_start, nrm, dep, torch.from_numpy(P).to(dev), npts, args.maskMethod,
                                                args.dataset, compose=1)
        para_this = copy.copy(args.para)
        para_this.sigmaAngle1 = args.para.sigmaAngle1[alter_]
        para_this.sigmaAngle2 = args.para.sigmaAngle2[alter_]
        para_this.sigmaDist = args.para.sigmaDist[alter_]
        para_this.sigmaFeat = args.para.sigmaFeat[alter_]
        Wd = torch.from_numpy(Wt).to(dev)
        res = match_pairs(pc[0:1].contiguous(), nn[0:1].contiguous(), ft[0:1].contiguous(), Wd[0:1].contiguous(),
                          pc[1:2].contiguous(), nn[1:2].contiguous(), ft[1:2].contiguous(), Wd[1:2].contiguous(),
                          npts[0:1].contiguous(), npts[1:2].contiguous(), para_this)
        R_hat = res.pose[0].cpu().numpy()
    return R_hat
