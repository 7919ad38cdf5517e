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


def render_room(rs, dataset="suncg", hole_frac=0.0, h=H, poses=None, half=None):
    """One scan pair of a box room: returns rgb[2,3,h,4h], norm[2,3,h,4h],
    depth[2,h,4h] (float32) and poses R[2,4,4] (camera-to-world).  ``poses`` / ``half``
    (optional) fix the two camera poses / the room half-extents instead of drawing them."""
    if half is None:
        half = rs.uniform(1.5, 4.0, 3)
    if poses is None:
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


def make_sift_detections(B, n, seed, mask_method="second", h=H):
    """Synthetic SIFT detections for RelativePosePipeline(keypoints="reference") (the detector itself -- cv2 -- is injected / out of scope,
    SURVEY.md 8a a6.3): per pair (source [n,2], target [n,2]) sub-pixel detector-frame coordinates -- the observed face [h,h] ('second',
    rputil.py:155-163) or the 640x480 kinect frame ('kinect', :255-265) -- a few pixels off the borders like SIFT's."""
    out = []
    for b in range(B):
        rs = np.random.RandomState(seed + 104729 * (b + 1))
        if mask_method == "kinect":
            det = lambda: np.stack((rs.uniform(8, 630, n), rs.uniform(8, 470, n)), 1)
        else:
            det = lambda: np.stack((rs.uniform(5, h - 7, n), rs.uniform(5, h - 7, n)), 1)
        out.append((det(), det()))
    return out


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


def make_tune_primitives(n, N, seed0):
    """A list of cached matching primitives in the reference's format (trainRelativePoseModuleRecFD.py:207-208: pc/normal/feat/weight of
    source and target + R_gt) from seeded synthetic matcher cases of growing size and inlier ratio."""
    out = []
    for i in range(n):
        S, T, G = make_match_case(N + 7 * i, seed0 + i, inlier=0.5 + 0.05 * i, noise=0.01)
        out.append({'pc_src': S['pc'], 'normal_src': S['normal'], 'feat_src': S['feat'], 'weight_src': S['weight'],
                    'pc_tgt': T['pc'], 'normal_tgt': T['normal'], 'feat_tgt': T['feat'], 'weight_tgt': T['weight'], 'R_gt': G})
    return out


# ---- well-conditioned scan pair (tests/golden "wc" fixtures) ---------------------------------------------------
def _ray_box(half, T, dataset, slot, px, py, h):
    """World points hit by the rays of sub-pixel panorama coords (px, py) of face ``slot`` (camera pose T)."""
    xn, yn = ((px - slot * h) / h - 0.5) * 2, (0.5 - py / h) * 2
    d_w = (np.stack([xn, yn, -np.ones_like(xn)], -1) @ face_rotation(dataset, slot).T) @ T[:3, :3].T
    tw = T[:3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(d_w > 0, (half - tw) / d_w, (-half - tw) / d_w)
    s[~np.isfinite(s)] = np.inf
    return tw + s.min(-1)[:, None] * d_w


def _project(P_w, T, dataset, slot, h):
    """Sub-pixel panorama coords of world points in face ``slot`` of the camera with pose T (+ in-frustum flag)."""
    q = ((P_w - T[:3, 3]) @ T[:3, :3]) @ face_rotation(dataset, slot)
    z = -q[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        xs, ys = q[:, 0] / z, q[:, 1] / z
    return (xs / 2 + 0.5) * h + slot * h, (0.5 - ys / 2) * h, (z > 0) & (np.abs(xs) < 1) & (np.abs(ys) < 1)


def make_wc_pair(seed, n_match=150, n_free=50, angle=0.2, shift=0.25, h=H, margin=14.0, dataset="suncg", mask_method="second"):
    """A scan pair on which the recurrent loop is WELL-CONDITIONED (any dataset convention / mask method: the face rotation
    table of ``dataset``, keypoints inside ``observed_box(mask_method)`` shrunk by ``margin`` pixels): the two
    cameras differ by a small rigid motion so that their observed faces see the same walls; ``n_match`` keypoints
    are the projections of common world points into BOTH observed faces (sub-pixel coordinates), ``n_free`` more
    per view are unrelated; all lie in the observed region (weight 1).  With a network whose descriptors follow
    the view-invariant wall texture (weights.make_descriptor_state_dict) the true correspondences dominate the
    matching graph.  Returns (data dict like make_pairs, pts [1,2,N,2], ptw [1,2,N], T_rel 4x4 source->target)."""
    rs = np.random.RandomState(seed)
    half = rs.uniform(2.0, 3.5, 3)
    T0 = random_rigid(rs, np.pi, 0.3)
    dT = random_rigid(rs, angle, shift)
    T1 = T0 @ dT
    rgb, nrm, dep, poses = render_room(rs, dataset, 0.0, h, poses=[T0, T1], half=half)
    N = n_match + n_free
    pts = np.zeros((1, 2, N, 2))
    y0, y1, x0, x1 = observed_box(mask_method, h)              # the observed region lies inside face slot 1 for both mask methods
    lo_x, hi_x, lo_y, hi_y = x0 + margin, x1 - margin, y0 + margin, y1 - margin
    got = 0
    while got < n_match:
        px = rs.uniform(lo_x, hi_x, 4 * n_match)
        py = rs.uniform(lo_y, hi_y, 4 * n_match)
        Pw = _ray_box(half, T0, dataset, 1, px, py, h)
        tx, ty, ok = _project(Pw, T1, dataset, 1, h)
        ok &= (tx > lo_x) & (tx < hi_x) & (ty > lo_y) & (ty < hi_y)
        k = min(int(ok.sum()), n_match - got)
        pts[0, 0, got:got + k, 0], pts[0, 0, got:got + k, 1] = px[ok][:k], py[ok][:k]
        pts[0, 1, got:got + k, 0], pts[0, 1, got:got + k, 1] = tx[ok][:k], ty[ok][:k]
        got += k
    for v in range(2):
        pts[0, v, n_match:, 0] = rs.uniform(lo_x, hi_x, n_free)
        pts[0, v, n_match:, 1] = rs.uniform(lo_y, hi_y, n_free)
    perm = rs.permutation(N)                       # the target list is not in source order
    pts[0, 1] = pts[0, 1][perm]
    ptw = np.ones((1, 2, N))
    data = {"rgb": rgb[None], "norm": nrm[None], "depth": dep[None], "R": poses[None]}
    return data, pts, ptw, np.linalg.inv(T1) @ T0


def make_keypoint_case(seed, kind="second", h=H):
    """Inputs of rputil.getKeypoint ('second') / getKeypoint_kinect ('kinect') with the SIFT detections given: two uint8 BGR panoramas,
    two smooth [32,h,4h] float32 feature maps (low-resolution noise, bilinearly upsampled: every distance map has well-separated
    minima), seeded sub-pixel detections in the detector's image coordinates (the observed face [h,h] / the 640x480 kinect frame),
    and the two full kinect frames (None for 'second').  Returns (rs, rt, feats, featt, det_s, det_t, rs_full, rt_full)."""
    rs_ = np.random.RandomState(seed)
    w = 4 * h

    def smooth(c):
        lo = rs_.randn(c, h // 8 + 1, w // 8 + 1)
        ys, xs = np.linspace(0, h // 8, h), np.linspace(0, w // 8, w)
        y0, x0 = np.minimum(ys.astype(int), h // 8 - 1), np.minimum(xs.astype(int), w // 8 - 1)
        fy, fx = (ys - y0)[None, :, None], (xs - x0)[None, None, :]
        a, b = lo[:, y0][:, :, x0], lo[:, y0][:, :, x0 + 1]
        c_, d = lo[:, y0 + 1][:, :, x0], lo[:, y0 + 1][:, :, x0 + 1]
        return np.ascontiguousarray(((a * (1 - fx) + b * fx) * (1 - fy) + (c_ * (1 - fx) + d * fx) * fy).astype(np.float32))
    feats, featt = smooth(32), smooth(32)
    rs = rs_.randint(0, 256, (h, w, 3)).astype(np.uint8)
    rt = rs_.randint(0, 256, (h, w, 3)).astype(np.uint8)
    if kind == "kinect":
        rs_full = rs_.randint(0, 256, (480, 640, 3)).astype(np.uint8)
        rt_full = rs_.randint(0, 256, (480, 640, 3)).astype(np.uint8)
        det = lambda n: np.stack((rs_.uniform(2, 636, n), rs_.uniform(2, 476, n)), 1)
        return rs, rt, feats, featt, det(57), det(43), rs_full, rt_full
    det = lambda n: np.stack((rs_.uniform(1, h - 3, n), rs_.uniform(1, h - 3, n)), 1)
    return rs, rt, feats, featt, det(41), det(25), None, None


def make_matching_primitive_case(seed, kind="second"):
    """Inputs of rpmodule.getMatchingPrimitive for the getKeypoint fixture (seed, kind) (tests/golden/gmp_nc.npz): make_keypoint_case + the depth /
    normal maps of a seeded synthetic scan pair as the 'completed' geometry.  Returns (dataset, dataS, dataT, det_s, det_t) with 'feat' as numpy
    (the caller wraps it in a tensor)."""
    ds = "scannet" if kind == "kinect" else "suncg"
    rs, rt, feats, featt, det_s, det_t, rs_full, rt_full = make_keypoint_case(seed, kind)
    d = make_pairs(1, 4000 + seed, ds)
    mk = lambda rgb, full, feat, v: {"rgb": rgb, "rgb_full": full, "feat": feat, "depth": d["depth"][0, v].astype(np.float64),
                                     "normal": np.ascontiguousarray(d["norm"][0, v].transpose(1, 2, 0)).astype(np.float64)}
    return ds, mk(rs, rs_full, feats, 0), mk(rt, rt_full, featt, 1), det_s, det_t


def make_full_res_pair(seed):
    """A synthetic full-resolution kinect pair (what util.parse_data's baseline branch takes on ScanNet, util.py:78-90):
    depth [1,2,480,640] f32 with holes (zeros), rgb uint8 [1,2,3,480,640]."""
    rs = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, 480), np.linspace(-1, 1, 640), indexing="ij")
    depth = np.stack([(2.0 + 0.6 * np.sin(3 * xx + k) * np.cos(2 * yy - k) + 0.05 * rs.rand(480, 640)) for k in range(2)])[None].astype(np.float32)
    depth[rs.rand(*depth.shape) < 0.15] = 0
    rgb = rs.randint(0, 256, (1, 2, 3, 480, 640)).astype(np.uint8)
    return depth, rgb
