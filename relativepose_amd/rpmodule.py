"""Host-side mirror of the reference pose module, backed by the HIP library.

Same names and argument meaning as /root/reference/RPModule/rpmodule.py and
rputil.py for the hot-path entry points:

  opts                               rputil.py:11-22
  RelativePoseEstimation_helper      rpmodule.py:317   (numpy dicts in, 4x4 numpy out)
  match_pairs                        batched device API the helper is built on

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


def _workspace(nbytes, dev):
    import torch
    key = (dev.index, torch.cuda.current_stream().cuda_stream)      # streams must not share scratch
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf


def match_pairs(pc_s, n_s, f_s, w_s, pc_t, n_t, f_t, w_t, ns, nt, para, debug=False, want_wij=False, max_edges=0):
    """Batched matcher on device tensors.
    pc_*/n_* [B,N,3] f64, f_* [B,N,32] f32, w_* [B,N] f64, ns/nt [B] int32 (all CUDA, contiguous).
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
    rc = L.relpose_match_pairs(C.byref(p), C.byref(kp), _lib.ptr(ws), ws.numel(), int(max_edges), _lib.ptr(res.pose),
                               _lib.ptr(res.status), C.byref(dbg) if dbg is not None else None, _lib.stream_ptr())
    _lib.check(rc, "relpose_match_pairs")
    return res


def affinity_topk(f_s, w_s, f_t, w_t, ns, nt, para, want_wij=True):
    """Stage A+B only (rpmodule.py:342-379): returns (wij or None, corres_j, corres_w, k_eff)."""
    import torch
    dev = _lib.require_gpu()
    L = _lib.lib()
    B, ns_max, nt_max = f_s.shape[0], f_s.shape[1], f_t.shape[1]
    p = _c_params(para)
    dummy = torch.zeros(1, dtype=torch.float64, device=dev)
    kp = _lib.Keypoints(B, ns_max, nt_max, _lib.ptr(ns), _lib.ptr(nt), _lib.ptr(dummy), _lib.ptr(dummy), _lib.ptr(f_s), _lib.ptr(w_s),
                        _lib.ptr(dummy), _lib.ptr(dummy), _lib.ptr(f_t), _lib.ptr(w_t))
    wij = torch.zeros(B, ns_max, nt_max, dtype=torch.float32, device=dev) if want_wij else None
    cj = torch.zeros(B, ns_max, p.topK, dtype=torch.int32, device=dev)
    cw = torch.zeros(B, ns_max, p.topK, dtype=torch.float64, device=dev)
    keff = torch.zeros(B, dtype=torch.int32, device=dev)
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
