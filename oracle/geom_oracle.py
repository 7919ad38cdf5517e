"""ORACLE (test infrastructure, not product code): numpy restatement of the
reference's panorama geometry on the hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this.  Follows /root/reference/util.py (apply_mask :209-232, warping :94-172,
depth2pc :468-523, reproj_helper :537-749, Pano2PointCloud :751-811),
RPModule/rputil.py (interpolate :43-58, getPixel/getPixel_helper :61-119) and
evaluation.py:217-270 (mask, valid channel, output composition).  Pinned against
the reference by tests/golden/make_golden.py.

The reference hard-codes 160x640; this restatement is parameterised by the
face size ``h`` (panorama = h x 4h) and validated against the reference at
h=160 only -- at other sizes parity is unpinned.
"""
import numpy as np

_RS = np.zeros((4, 3, 3))
_RS[0] = np.eye(3)
_RS[1] = [[0, 0, -1], [0, 1, 0], [1, 0, 0]]
_RS[2] = [[-1, 0, 0], [0, 1, 0], [0, 0, -1]]
_RS[3] = [[0, 0, 1], [0, 1, 0], [-1, 0, 0]]

KINECT_FX, KINECT_FY = 0.8921875 * 2, 1.1895 * 2


def _shift(dataset):
    return 0 if 'suncg' in dataset else -1


def face_R(dataset, slot):
    return _RS[(slot + _shift(dataset)) % 4]


def kinect_box(h):
    s = h / 160.0
    dw, dh = int(int(89.67 // 2) * s), int(int(67.25 // 2) * s)
    return h // 2 - dh, h // 2 + dh, h + h // 2 - dw, h + h // 2 + dw


def apply_mask(x, method):
    """util.py:209-232 (geow is unused at inference and omitted).
    x [n,c,h,4h] float32 -> (x*mask, mask[n,1,h,4h]) float32."""
    n, _, h, w = x.shape
    m = np.zeros((n, 1, h, w), np.float32)
    if method == 'second':
        m[:, :, :h, h:2 * h] = 1
    elif method == 'kinect':
        y0, y1, x0, x1 = kinect_box(h)
        m[:, :, y0:y1, x0:x1] = 1
    else:
        raise ValueError(method)
    return x * m, m


def build_view(rgb, norm, depth, method):
    """evaluation.py:217-230: masked 7 channels + valid (depth!=0) channel.
    rgb/norm [3,h,4h], depth [h,4h] -> view [1,8,h,4h] f32, mask [h,4h,1] f32."""
    comp = np.concatenate((rgb, norm, depth[None]), 0)[None].astype(np.float32)
    v, m = apply_mask(comp.copy(), method)
    valid = (v[:, 6:7] != 0).astype(np.float32)
    return np.concatenate((v, valid), 1), m[0].transpose(1, 2, 0)


def pano2pc(depth, dataset):
    """util.py:751-811.  depth [h,4h] -> [3, n] float64, faces concatenated
    (scannet drops zero-depth pixels and divides x,y by the kinect factors)."""
    h = depth.shape[0]
    assert depth.shape[1] == 4 * h
    ys, xs = np.meshgrid(range(h), range(h), indexing='ij')
    ys, xs = ((0.5 - ys / h) * 2).flatten(), ((xs / h - 0.5) * 2).flatten()
    out = []
    for i in range(4):
        zs = depth[:, i * h:(i + 1) * h].flatten()
        if 'scannet' in dataset:
            k = zs != 0
            zs = zs[k]
            y_, x_ = ys[k] * zs / KINECT_FY, xs[k] * zs / KINECT_FX
        else:
            y_, x_ = ys * zs, xs * zs
        p = np.concatenate((x_, y_, -zs)).reshape(3, -1)
        out.append(np.matmul(face_R(dataset, i), p))
    return np.concatenate(out, 1)


def depth2pc(depth, dataset):
    """util.py:468-523 for the shapes the hot path uses: one h x h face
    (suncg: rotated by Rs[1]; matterport: not rotated) or the kinect crop."""
    hh, ww = depth.shape
    ys, xs = np.meshgrid(range(hh), range(ww), indexing='ij')
    ys, xs = (0.5 - ys / hh) * 2, (xs / ww - 0.5) * 2
    zs = depth.flatten()
    mask = zs != 0
    zs = zs[mask]
    xs = xs.flatten()[mask] * zs
    ys = ys.flatten()[mask] * zs
    if 'scannet' in dataset:
        if (hh, ww) == (480, 640):
            pc = np.stack((xs / KINECT_FX, ys / KINECT_FY, -zs), 1)
        else:
            # the crop is (66,88) at h=160; the reference rescales by /160
            pc = np.stack((xs * ww / 160, ys * hh / 160, -zs), 1)
    else:
        assert hh == ww
        pc = np.stack((xs, ys, -zs), 1)
        if 'suncg' in dataset:
            pc = np.matmul(_RS[1], pc.T).T
    return pc, mask


def _reproject(pts, vals, h, mode, dataset):
    """reproj_helper util.py:537-749: scatter the points into the four face
    slots; within a slot the last point (in point order) wins."""
    shape = (h, 4 * h) if mode == 'depth' else (h, 4 * h, 3)
    proj = np.zeros(shape)
    for slot in range(4):
        tp = np.matmul(face_R(dataset, slot).T, pts) if not (slot + _shift(dataset)) % 4 == 0 else pts.copy()
        tp[:2, :] /= (np.abs(tp[2, :]) + 1e-32)
        hit = (tp[2, :] < 0) * (np.abs(tp[0, :]) < 1) * (np.abs(tp[1, :]) < 1)
        v = -tp[2, hit] if mode == 'depth' else vals[hit, :]
        c = tp[:2, hit]
        c[0, :] = (c[0, :] + 1) * 0.5 * h
        c[1, :] = (1 - c[1, :]) * 0.5 * h
        c = c.round().clip(0, h - 1).astype('int')
        c[0, :] += slot * h
        proj[c[1, :], c[0, :]] = v
    return proj


def warping(view, R, dataset):
    """util.py:94-172.  view [1,8,h,4h] (numpy f32), R [4,4] f64 ->
    [1,8,h,4h] f64; identity R gives zeros (util.py:95-96)."""
    if np.linalg.norm(R - np.eye(4)) == 0:
        return np.zeros(view.shape)
    h = view.shape[2]
    rgb = view[0, 0:3].transpose(1, 2, 0)
    nrm = view[0, 3:6].transpose(1, 2, 0)
    dep = view[0, 6]
    if 'suncg' in dataset:
        # observed face = slot 1; zero depths are NOT dropped (util.py:114-123)
        pts = pano2pc(dep, 'suncg')[:, h * h:2 * h * h]
        col = rgb[:, h:2 * h, :].reshape(-1, 3)
        nn = nrm[:, h:2 * h, :].reshape(-1, 3)
    elif 'matterport' in dataset:
        pc, k = depth2pc(dep[:, h:2 * h], 'matterport')
        pts = pc.T
        col = rgb[:, h:2 * h, :].reshape(-1, 3)[k, :]
        nn = nrm[:, h:2 * h, :].reshape(-1, 3)[k, :]
    else:
        y0, y1, x0, x1 = kinect_box(h)
        pc, k = depth2pc(dep[y0:y1, x0:x1], 'scannet')
        pts = pc.T
        col = rgb[y0:y1, x0:x1, :].reshape(-1, 3)[k]
        nn = nrm[y0:y1, x0:x1, :].reshape(-1, 3)[k]
    pts = np.matmul(R, np.concatenate((pts, np.ones([1, pts.shape[1]]))))[:3, :]
    nn = np.matmul(R[:3, :3], nn.T).T
    c = _reproject(pts, col, h, 'color', dataset)
    n = _reproject(pts, nn, h, 'normal', dataset)
    d = _reproject(pts, None, h, 'depth', dataset)
    m = (d != 0).astype('int')
    return np.concatenate((c, n, d[:, :, None], m[:, :, None]), 2)[None].transpose(0, 3, 1, 2)


def compose(f, mask, obs_normal, obs_depth):
    """evaluation.py:246-253.  f [C,h,4h] f32 net output, mask [h,4h,1] f32,
    obs_normal [h,4h,3] (the complete input normal), obs_depth [h,4h].
    Note the normal is divided by the norm of the *input* normal + 1e-6."""
    n = (1 - mask) * f[3:6].transpose(1, 2, 0) + mask * obs_normal
    n = n / (np.linalg.norm(obs_normal, axis=2, keepdims=True) + 1e-6)
    d = (1 - mask[:, :, 0]) * f[6] + mask[:, :, 0] * obs_depth
    return n, d


def interpolate(feat, pt):
    """rputil.py:43-58 in float32 (the reference runs it in torch float32).
    feat [c,h,w] f32, pt [k,2] f32 normalised -> [c,k] f32."""
    feat = feat.astype(np.float32)
    pt = pt.astype(np.float32)
    h, w = feat.shape[1], feat.shape[2]
    x = pt[:, 0] * np.float32(w - 1)
    y = pt[:, 1] * np.float32(h - 1)
    x0, y0 = np.floor(x), np.floor(y)
    xi, yi = x0.astype(np.int64), y0.astype(np.int64)
    one = np.float32(1)
    return (feat[:, yi, xi] * (x0 + one - x) * (y0 + one - y)
            + feat[:, yi + 1, xi] * (x0 + one - x) * (y - y0)
            + feat[:, yi, xi + 1] * (x - x0) * (y0 + one - y)
            + feat[:, yi + 1, xi + 1] * (x - x0) * (y - y0))


def get_pixel(depth, normal, pts, dataset):
    """rputil.py:61-119.  Bilinear depth/normal at sub-pixel pts [k,2] (pixel
    units, x<=W-2, y<=H-2), unproject with the per-face rotation.  Normals are
    renormalised but NOT rotated (as in the reference).  -> pc [3,k], nn [k,3]."""
    h = depth.shape[0]
    tp = np.floor(pts).astype('int')
    fx1, fx0 = pts[:, 0] - tp[:, 0], tp[:, 0] + 1 - pts[:, 0]
    fy1, fy0 = pts[:, 1] - tp[:, 1], tp[:, 1] + 1 - pts[:, 1]
    val = (depth[tp[:, 1], tp[:, 0]] * fy0 * fx0 + depth[tp[:, 1], tp[:, 0] + 1] * fx1 * fy0
           + depth[tp[:, 1] + 1, tp[:, 0]] * fy1 * fx0 + depth[tp[:, 1] + 1, tp[:, 0] + 1] * fx1 * fy1)
    nn = (normal[tp[:, 1], tp[:, 0], :] * fy0[:, None] * fx0[:, None]
          + normal[tp[:, 1], tp[:, 0] + 1, :] * fx1[:, None] * fy0[:, None]
          + normal[tp[:, 1] + 1, tp[:, 0], :] * fy1[:, None] * fx0[:, None]
          + normal[tp[:, 1] + 1, tp[:, 0] + 1, :] * fx1[:, None] * fy1[:, None])
    nn = nn / np.linalg.norm(nn, axis=1, keepdims=True)
    pc = np.zeros((len(pts), 3))
    for i in range(len(pts)):
        slot = int(pts[i, 0] // h)
        ystp, xstp = (0.5 - pts[i, 1] / h) * 2, ((pts[i, 0] - slot * h) / h - 0.5) * 2
        z = val[i]
        pc[i] = np.matmul(face_R(dataset, slot), np.array([xstp * z, ystp * z, -z]))
    return pc.T, nn


def feature_distance_map(fs_sel, featt):
    """rputil.py:186 (torch float32): fs_sel [32,n] descriptors, featt [32,H,W] -> dist [n,H,W]."""
    import torch
    a = torch.from_numpy(np.ascontiguousarray(fs_sel, dtype=np.float32))
    b = torch.from_numpy(np.ascontiguousarray(featt, dtype=np.float32))
    C, H, W = b.shape
    return (a.unsqueeze(2) - b.view(C, 1, -1)).pow(2).sum(0).view(a.shape[1], H, W).numpy()


def sampling(dist, K, window=15):
    """rputil.Sampling :355-371: heat = exp(-dist/2); K x {argmax, suppress [y-15,y+15) x [x-15,x+15) with the min}."""
    heat = np.exp(-dist / 2)
    n, h, w = heat.shape
    pt = np.zeros([n, K, 2])
    for i in range(n):
        for j in range(K):
            idx = np.argmax(heat[i])
            y, x = np.unravel_index(idx, heat[i].shape)
            pt[i, j] = (x, y)
            heat[i][max(0, y - window):min(h - 1, y + window), max(0, x - window):min(w - 1, x + window)] = heat[i].min()
    return pt
