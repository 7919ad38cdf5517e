"""Seeded synthetic inputs for the relative-pose hot path.

No dataset or pretrained checkpoint ships with the reference, so every parity
test and the bench run on data produced here from ``np.random.RandomState``.
Shapes and value ranges mirror what the reference dataset loaders hand to
``evaluation.py`` (datasets/SUNCG.py:492-499): ``rgb`` in [0,1], unit
``norm``, metric ``depth``, all 160x640 four-face skybox panoramas.

Nothing in this module touches the GPU or the oracle.
"""
import numpy as np

H, W = 160, 640
DATASETS = ("suncg", "matterport", "scannet")

# face rotations of the skybox, index = face slot after dataset shift
_RS = np.zeros((4, 3, 3))
_RS[0] = np.eye(3)
_RS[1] = [[0, 0, -1], [0, 1, 0], [1, 0, 0]]
_RS[2] = [[-1, 0, 0], [0, 1, 0], [0, 0, -1]]
_RS[3] = [[0, 0, 1], [0, 1, 0], [-1, 0, 0]]


def face_rotation(dataset, slot):
    """Rotation (face frame -> panorama frame) of column block ``slot``.

    suncg uses Rs[slot]; matterport/scannet use Rs[(slot-1)%4]
    (reference util.py:766-810)."""
    return _RS[slot if "suncg" in dataset else (slot - 1) % 4]


def random_rigid(rs, max_angle=np.pi, max_t=1.0):
    ax = rs.randn(3)
    ax /= np.linalg.norm(ax)
    th = rs.uniform(0, max_angle)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    t = rs.randn(3)
    t *= rs.uniform(0, max_t) / np.linalg.norm(t)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def _texture(p):
    """Smooth procedural colour of a world point, in [0,1]^3."""
    f = np.stack([np.sin(p[..., 0] * 2.1 + p[..., 1] * 1.3),
                  np.sin(p[..., 1] * 1.7 - p[..., 2] * 2.3 + 1.0),
                  np.sin(p[..., 2] * 1.9 + p[..., 0] * 0.7 + 2.0)], -1)
    return (0.5 + 0.5 * f).astype(np.float32)


def render_room(rs, dataset="suncg", hole_frac=0.0, h=H):
    """One scan pair of a box room: returns rgb[2,3,h,4h], norm[2,3,h,4h],
    depth[2,h,4h] (float32) and poses R[2,4,4] (camera-to-world)."""
    half = rs.uniform(1.5, 4.0, 3)
    poses = []
    for _ in range(2):
        T = random_rigid(rs, np.pi, 1.0)
        T[:3, 3] = np.clip(T[:3, 3], -0.6 * half, 0.6 * half)
        poses.append(T)
    rgb = np.zeros((2, 3, h, 4 * h), np.float32)
    nrm = np.zeros((2, 3, h, 4 * h), np.float32)
    dep = np.zeros((2, h, 4 * h), np.float32)
    ys, xs = np.meshgrid(np.arange(h), np.arange(h), indexing="ij")
    yn, xn = (0.5 - ys / h) * 2, (xs / h - 0.5) * 2
    for v in range(2):
        Rw, tw = poses[v][:3, :3], poses[v][:3, 3]
        for slot in range(4):
            Rf = face_rotation(dataset, slot)
            # direction of unit z-depth in the panorama frame
            d_face = np.stack([xn, yn, -np.ones_like(xn)], -1)  # (x,y,-z) at z=1
            d_cam = d_face @ Rf.T
            d_w = d_cam @ Rw.T
            # ray-box (inside) intersection: smallest positive s with |tw+s*d|=half
            with np.errstate(divide="ignore", invalid="ignore"):
                s1 = (half - tw) / d_w
                s2 = (-half - tw) / d_w
            s = np.where(d_w > 0, s1, s2)
            s[~np.isfinite(s)] = np.inf
            axis = np.argmin(s, -1)
            z = np.min(s, -1)
            hit = tw + z[..., None] * d_w
            n_w = np.zeros_like(hit)
            sign = -np.sign(np.take_along_axis(d_w, axis[..., None], -1))[..., 0]
            for a in range(3):
                n_w[..., a] = np.where(axis == a, sign, 0.0)
            n_cam = n_w @ Rw  # world -> camera
            sl = slice(slot * h, (slot + 1) * h)
            dep[v, :, sl] = z
            nrm[v, :, :, sl] = np.moveaxis(n_cam, -1, 0)
            rgb[v, :, :, sl] = np.moveaxis(_texture(hit), -1, 0)
        if hole_frac > 0:
            holes = rs.rand(h, 4 * h) < hole_frac
            dep[v][holes] = 0
    return rgb, nrm, dep, np.stack(poses)


def make_pairs(B, seed, dataset="suncg", h=H):
    """A batch shaped like the reference DataLoader output (batch dim = pairs)."""
    rgb = np.zeros((B, 2, 3, h, 4 * h), np.float32)
    nrm = np.zeros((B, 2, 3, h, 4 * h), np.float32)
    dep = np.zeros((B, 2, h, 4 * h), np.float32)
    Rp = np.zeros((B, 2, 4, 4))
    hole = 0.0 if "suncg" in dataset else 0.01
    for b in range(B):
        rs = np.random.RandomState(seed + b)
        rgb[b], nrm[b], dep[b], Rp[b] = render_room(rs, dataset, hole, h)
    return {"rgb": rgb, "norm": nrm, "depth": dep, "R": Rp}


def observed_box(mask_method, h=H):
    """(y0,y1,x0,x1) of the observed region (reference util.py:215-228)."""
    if mask_method == "second":
        return 0, h, h, 2 * h
    s = h / 160.0
    dw, dh = int(int(89.67 // 2) * s), int(int(67.25 // 2) * s)
    return h // 2 - dh, h // 2 + dh, h + h // 2 - dw, h + h // 2 + dw


def make_keypoints(B, N, seed, mask_method="second", h=H):
    """Injected keypoints (replaces cv2 SIFT): per pair and
    view N sub-pixel (x,y) coords, half inside the observed region (weight 1),
    half outside (weight .99), x<=W-2, y<=H-2.  Returns pts[B,2,N,2] f64,
    w[B,2,N] f64."""
    w_ = 4 * h
    y0, y1, x0, x1 = observed_box(mask_method, h)
    pts = np.zeros((B, 2, N, 2))
    wts = np.zeros((B, 2, N))
    for b in range(B):
        rs = np.random.RandomState(seed + 7919 * (b + 1))
        for v in range(2):
            n_in = N // 2
            xin = rs.uniform(x0, min(x1, w_ - 2), n_in)
            yin = rs.uniform(y0, min(y1, h - 2), n_in)
            xo, yo = [], []
            while len(xo) < N - n_in:
                x, y = rs.uniform(0, w_ - 2), rs.uniform(0, h - 2)
                if not (x0 <= x <= x1 and y0 <= y <= y1):
                    xo.append(x)
                    yo.append(y)
            pts[b, v, :n_in, 0], pts[b, v, :n_in, 1] = xin, yin
            pts[b, v, n_in:, 0], pts[b, v, n_in:, 1] = xo, yo
            wts[b, v, :n_in] = 1.0
            wts[b, v, n_in:] = 0.99
    return pts, wts


def make_match_case(N, seed, inlier=0.6, noise=0.005, Nt=None):
    """Matcher-only input (helper dict format, reference rpmodule.py:317-326):
    random cloud, rigidly moved + permuted target, 60 % inliers."""
    rs = np.random.RandomState(seed)
    Nt = N if Nt is None else Nt
    P = rs.uniform(-2, 2, (N, 3))
    n = rs.randn(N, 3)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    f = np.tanh(rs.randn(N, 32)).astype(np.float32)
    T = random_rigid(rs, np.pi, 1.0)
    R, t = T[:3, :3], T[:3, 3]
    m = min(N, Nt)
    Pt = rs.uniform(-2, 2, (Nt, 3))
    nt = rs.randn(Nt, 3)
    nt /= np.linalg.norm(nt, axis=1, keepdims=True)
    ft = np.tanh(rs.randn(Nt, 32)).astype(np.float32)
    inl = np.where(rs.rand(m) < inlier)[0]
    Pt[inl] = P[inl] @ R.T + t + rs.randn(len(inl), 3) * noise
    nn = n[inl] @ R.T + rs.randn(len(inl), 3) * 0.01
    nt[inl] = nn / np.linalg.norm(nn, axis=1, keepdims=True)
    ft[inl] = np.tanh(np.arctanh(np.clip(f[inl], -0.999, 0.999)) + rs.randn(len(inl), 32) * 0.02).astype(np.float32)
    perm = rs.permutation(Nt)
    Pt, nt, ft = Pt[perm], nt[perm], ft[perm]
    ws = np.where(rs.rand(N) < 0.5, 1.0, 0.99)
    wt = np.where(rs.rand(Nt) < 0.5, 1.0, 0.99)
    S = {"pc": P, "normal": n, "feat": f, "weight": ws}
    Tt = {"pc": Pt, "normal": nt, "feat": ft, "weight": wt}
    return S, Tt, T
