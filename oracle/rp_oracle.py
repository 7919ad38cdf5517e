"""ORACLE (test infrastructure, not product code): CPU restatement in numpy of the
reference's spectral-matching pose module.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this.  The shipped path is the HIP library; it never calls into here.

Follows /root/reference/RPModule/rpmodule.py (lines cited per function) and
RPModule/rputil.py:11-22 (``opts``).  Pinned against the reference itself: tests/golden/make_golden.py imports the
reference (in the build container) and stores its outputs for seeded inputs in
tests/golden/*.npz; tests/test_oracle_golden.py asserts this oracle equals them.

Each stage returns its intermediates so the HIP stages can be checked one by
one.  Numeric types follow the reference exactly: descriptors/dij float32,
everything else float64.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

FEAT_SCALING = 100  # rpmodule.py:327
OBS_W = 1.2         # rpmodule.py:328
UNOBS_PENALTY = 0.6  # rpmodule.py:467

STATUS_OK, STATUS_FEW_KEYPOINTS, STATUS_DIST_FILTER, STATUS_ANGLE_FILTER, STATUS_ZERO_WEIGHT = 0, 1, 2, 3, 4


class Params:
    """Hyper-parameters, rputil.py:11-22."""

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


def sum32_lanes8(sq):
    """float32 sum over a trailing axis of length 32 in the order numpy's
    pairwise reduction uses for a contiguous axis (8 strided partial sums, then
    a fixed tree).  Written out so the order does not depend on the numpy build
    (SURVEY.md §7 'bit-exact correspondence indices')."""
    assert sq.dtype == np.float32 and sq.shape[-1] == 32
    r = sq[..., 0:8].copy()
    for i in (8, 16, 24):
        r = r + sq[..., i:i + 8]
    return ((r[..., 0] + r[..., 1]) + (r[..., 2] + r[..., 3])) + ((r[..., 4] + r[..., 5]) + (r[..., 6] + r[..., 7]))


def affinity(featS, featT, wS, wT, sigmaFeat):
    """rpmodule.py:342-363.  Returns dij (f32), eij (f64 exponent), wij (f64,
    row-normalised)."""
    fs = (featS / FEAT_SCALING).astype(np.float32)
    ft = (featT / FEAT_SCALING).astype(np.float32)
    diff = fs[:, None, :] - ft[None, :, :]
    dij = sum32_lanes8(diff * diff)
    both = (wS[:, None] * wT[None, :]) == 1
    sig = np.where(both, sigmaFeat / OBS_W, np.ones(both.shape) * sigmaFeat)
    eij = np.divide(-dij, 2 * np.power(sig / 5, 2))
    wij = np.exp(eij)
    nm = np.linalg.norm(wij, axis=1, keepdims=True)
    zero = (nm == 0)
    nm[zero] = 1
    wij = wij / nm
    wij[zero.squeeze(1), :] = 0
    return dij, eij, wij


def topk(wij, K):
    """rpmodule.py:367-375.  corres[2,C] int64."""
    ns = wij.shape[0]
    k = min(K, wij.shape[1] - 1)
    idx = np.argpartition(-wij, k, axis=1)[:, :k]
    corres = np.zeros((2, ns * k), dtype=np.int64)
    corres[0] = np.arange(ns).repeat(k)
    corres[1] = idx.reshape(-1)
    return corres


def _acos_dot(a, b):
    return np.arccos((a * b).sum(1).clip(-1, 1))


def pair_consistency(S, T, corres, wij, p):
    """rpmodule.py:381-467.  Returns dict with status, surviving pair index
    arrays (c1 < c2 index into corres columns), weights and filter counts."""
    C = corres.shape[1]
    c2, c1 = np.meshgrid(np.arange(C), np.arange(C))   # c1 = row (idy), c2 = col (idx)
    keep = c2 > c1
    c1, c2 = c1[keep], c2[keep]                        # row-major: c1 ascending, then c2
    ps, pt, ns_, nt_ = S['pc'], T['pc'], S['normal'], T['normal']
    i1, j1, i2, j2 = corres[0, c1], corres[1, c1], corres[0, c2], corres[1, c2]
    dis_s = np.linalg.norm(ps[i1] - ps[i2], axis=1)
    dis_t = np.linalg.norm(pt[j1] - pt[j2], axis=1)
    d = np.power(dis_s - dis_t, 2)
    ok = np.logical_and(d < np.power(p.distThre, 2), np.minimum(dis_s, dis_t) > 1.5 * np.power(p.distSepThre, 2))
    out = {'n_pairs': len(c1), 'n_dist': int(ok.sum())}
    if ok.sum() < 3:
        out['status'] = STATUS_DIST_FILTER
        return out
    c1, c2, d = c1[ok], c2[ok], d[ok]
    i1, j1, i2, j2 = corres[0, c1], corres[1, c1], corres[0, c2], corres[1, c2]
    e1 = ps[i1] - ps[i2]
    e2 = pt[j1] - pt[j2]
    e1 = e1 / np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = e2 / np.linalg.norm(e2, axis=1, keepdims=True)
    al = np.power(_acos_dot(ns_[i1], ns_[i2]) - _acos_dot(nt_[j1], nt_[j2]), 2)
    be = np.power(_acos_dot(ns_[i1], e1) - _acos_dot(nt_[j1], e2), 2)
    ga = np.power(_acos_dot(ns_[i2], e1) - _acos_dot(nt_[j2], e2), 2)
    a2 = np.power(p.angleThre, 2)
    ok = (al < a2) & (be < a2) & (ga < a2)
    out['n_angle'] = int(ok.sum())
    if ok.sum() < 3:
        out['status'] = STATUS_ANGLE_FILTER
        return out
    c1, c2, d, al, be, ga = c1[ok], c2[ok], d[ok], al[ok], be[ok], ga[ok]
    i1, j1, i2, j2 = corres[0, c1], corres[1, c1], corres[0, c2], corres[1, c2]
    w = wij[i1, j1] * wij[i2, j2] * np.exp(-d / (2 * p.sigmaDist ** 2) - al / (2 * p.sigmaAngle1 ** 2)
                                           - be / (2 * p.sigmaAngle2 ** 2) - ga / (2 * p.sigmaAngle2 ** 2))
    ww = S['weight'][i1] * S['weight'][i2] * T['weight'][j1] * T['weight'][j2]
    w[ww != 1] *= UNOBS_PENALTY
    out.update(c1=c1, c2=c2, w=w)
    out['status'] = STATUS_ZERO_WEIGHT if (w != 0).sum() < 1 else STATUS_OK
    return out


def horn87(src, tgt, weight):
    """rpmodule.py:17-58 for one problem.  src,tgt [3,n], weight [n] -> R[3,3]."""
    M = src @ (tgt * weight[None, :]).T
    N = np.array([
        [M[0, 0] + M[1, 1] + M[2, 2], M[1, 2] - M[2, 1], M[2, 0] - M[0, 2], M[0, 1] - M[1, 0]],
        [M[1, 2] - M[2, 1], M[0, 0] - M[1, 1] - M[2, 2], M[0, 1] + M[1, 0], M[0, 2] + M[2, 0]],
        [M[2, 0] - M[0, 2], M[0, 1] + M[1, 0], M[1, 1] - M[0, 0] - M[2, 2], M[1, 2] + M[2, 1]],
        [M[0, 1] - M[1, 0], M[2, 0] + M[0, 2], M[1, 2] + M[2, 1], M[2, 2] - M[0, 0] - M[1, 1]]])
    v, u = np.linalg.eig(N)
    q = np.real(u[:, np.argmax(v)])
    return quat_to_rot(q)


def quat_to_rot(q):
    a, b, c, d = q
    return np.array([
        [a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
        [2 * (c * b + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
        [2 * (d * b - a * c), 2 * (d * c + a * b), a * a - b * b - c * c + d * d]])


class _Fit:
    """State shared by the four fit methods: the stacked [2M] point/normal
    arrays the reference builds at rpmodule.py:474-489."""
    EPS = 1e-12
    OFFSET = 50

    def __init__(self, S, T, corres, pc, mu):
        c1, c2 = pc['c1'], pc['c2']
        i1, j1, i2, j2 = corres[0, c1], corres[1, c1], corres[0, c2], corres[1, c2]
        self.SP = np.concatenate((S['pc'][i1], S['pc'][i2]))
        self.TP = np.concatenate((T['pc'][j1], T['pc'][j2]))
        self.SN = np.concatenate((S['normal'][i1], S['normal'][i2]))
        self.TN = np.concatenate((T['normal'][j1], T['normal'][j2]))
        self.w = pc['w']
        self.mu = mu
        self.nt = T['pc'].shape[0]
        self.ns = S['pc'].shape[0]
        self.row = i1 * self.nt + j1
        self.col = i2 * self.nt + j2

    def solve(self, W):
        """centre with position weights, Horn; W is the [4M] weight vector."""
        WP = W[:len(W) // 2]
        den = WP.sum() + self.EPS
        self.ms = (self.SP * WP[:, None]).sum(0) / den
        self.mt = (self.TP * WP[:, None]).sum(0) / den
        self.SPc, self.TPc = self.SP - self.ms, self.TP - self.mt
        R = horn87(np.concatenate((self.SPc, self.SN)).T, np.concatenate((self.TPc, self.TN)).T, W)
        t = -R @ self.ms + self.mt
        return R, t

    def residuals(self, R):
        rp = self.mu * np.power(R @ self.SPc.T - self.TPc.T, 2).sum(0)
        rn = np.power(R @ self.SN.T - self.TN.T, 2).sum(0)
        return rp, rn

    def irls(self, W, n_iter=5):
        for _ in range(n_iter):
            R, t = self.solve(W)
            rp, rn = self.residuals(R)
            W = W * 1.0 / (1.0 + np.concatenate((rp, rn)))
        return R, t, W

    def spectral_weights(self, R, base_w):
        """rpmodule.py:262-285: leading eigenvector of the pair-compatibility
        graph -> per-pair weights x."""
        rp, rn = self.residuals(R)
        a = base_w * (self.OFFSET - (rp + rn))
        a[a < 0] = 0
        a = a.reshape(2, -1).sum(0)
        n = self.ns * self.nt
        A = sp.csc_matrix((a, (self.row, self.col)), shape=(n, n))
        A = A + A.T
        _, u = spla.eigs(A, k=1)
        u = u.real
        u /= np.linalg.norm(u)
        x = (u[self.row] * u[self.col]).squeeze()
        x[x < 0] = 0
        return x * self.w, a


def _pose(R, t):
    P = np.eye(4)
    P[:3, :3] = R
    P[:3, 3] = t
    return P


def fit(S, T, corres, pc, p, trace=None):
    """Dispatch on p.method: rpmodule.py:491-508 (fit_horn87 :60, fit_spectral
    :86, fit_irls :169, fit_irls_sm :212)."""
    f = _Fit(S, T, corres, pc, p.mu)
    w2 = np.concatenate((f.w, f.w))
    W0 = np.concatenate((w2 * p.mu, w2))
    if p.method == 'horn87':
        return _pose(*f.solve(W0))
    if p.method == 'irls':
        R, t, _ = f.irls(W0)
        return _pose(R, t)
    if p.method == 'spectral':
        R, t = f.solve(W0)
        base = w2                        # fit_spectral multiplies by allWP (:126)
        for _ in range(5):
            x, _ = f.spectral_weights(R, base)
            W = np.tile(x, 4)
            W[:len(W) // 2] *= p.mu
            base = W[:len(W) // 2]       # allWP is rebound to mu*x (:148)
            R, t = f.solve(W)
        return _pose(R, t)
    if p.method == 'irls+sm':
        R, t, _ = f.irls(W0)
        if trace is not None:
            trace.append(_pose(R, t))
        for _ in range(5):
            x, _ = f.spectral_weights(R, w2)
            W = np.tile(x, 4)
            W[:len(W) // 2] *= p.mu
            R, t, _ = f.irls(W)
            if trace is not None:
                trace.append(_pose(R, t))
        return _pose(R, t)
    raise Exception("unknown method!")


def relative_pose_helper(S, T, p, detail=None):
    """RelativePoseEstimation_helper, rpmodule.py:317-508.  Degenerate inputs
    return identity (5 early exits)."""
    d = {} if detail is None else detail
    d['status'] = STATUS_OK
    if S['pc'].shape[0] < 3 or T['pc'].shape[0] < 3:
        d['status'] = STATUS_FEW_KEYPOINTS
        return np.eye(4)
    dij, eij, wij = affinity(S['feat'], T['feat'], S['weight'], T['weight'], p.sigmaFeat)
    corres = topk(wij, p.topK)
    d.update(dij=dij, eij=eij, wij=wij, corres=corres)
    if corres.shape[1] < 3:
        d['status'] = STATUS_FEW_KEYPOINTS
        return np.eye(4)
    pc = pair_consistency(S, T, corres, wij, p)
    d['pairs'] = pc
    d['status'] = pc['status']
    if pc['status'] != STATUS_OK:
        return np.eye(4)
    trace = []
    d['trace'] = trace
    return fit(S, T, corres, pc, p, trace)
