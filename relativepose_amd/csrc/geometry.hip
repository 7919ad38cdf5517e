// Panorama geometry on gfx950: mask, unprojection, rigid warp with deterministic
// scatter re-projection, output composition + keypoint sampling.  One thread per
// pixel / point / keypoint; these kernels are HBM-bound streaming passes.
//
// Reference sites: util.py apply_mask :209-232, warping :94-172, depth2pc :468-523,
// reproj_helper :537-749, Pano2PointCloud :751-811; rputil.py interpolate :43-58,
// getPixel :61-119; evaluation.py :217-270.
//
// Compiled with -ffp-contract=off: pixel rounding and the float32 bilinear
// descriptor gather must round like numpy / torch (no FMA fusion).
#include "common.h"
#include "rp_math.h"
#include <limits.h>

namespace {

// Face rotations Rs[k] of the skybox (util.py:757-761).  Entries are 0/+-1, so applying
// them is an exact permutation/sign flip.  ROT(k, v) = Rs[k] @ v ;  ROTT = Rs[k]^T @ v.
__device__ __forceinline__ void face_rot(int k, double x, double y, double z, double& ox, double& oy, double& oz) {
    switch (k & 3) {
        case 0: ox = x; oy = y; oz = z; break;
        case 1: ox = -z; oy = y; oz = x; break;
        case 2: ox = -x; oy = y; oz = -z; break;
        default: ox = z; oy = y; oz = -x; break;
    }
}
__device__ __forceinline__ void face_rot_t(int k, double x, double y, double z, double& ox, double& oy, double& oz) {
    switch (k & 3) {
        case 0: ox = x; oy = y; oz = z; break;
        case 1: ox = z; oy = y; oz = -x; break;
        case 2: ox = -x; oy = y; oz = -z; break;
        default: ox = -z; oy = y; oz = x; break;
    }
}
__device__ __forceinline__ int face_index(int dataset, int slot) { return dataset == RELPOSE_SUNCG ? slot : (slot + 3) & 3; }

struct Box { int y0, y1, x0, x1; };
__host__ __device__ inline Box observed_box(int method, int h) {
    Box b;
    if (method == RELPOSE_MASK_SECOND) { b.y0 = 0; b.y1 = h; b.x0 = h; b.x1 = 2 * h; }
    else {
        // util.py:226-228: dw = int(89.67//2) = 44, dh = int(67.25//2) = 33 at h = 160
        int dw = (int)(44 * (h / 160.0)), dh = (int)(33 * (h / 160.0));
        b.y0 = h / 2 - dh; b.y1 = h / 2 + dh; b.x0 = h + h / 2 - dw; b.x1 = h + h / 2 + dw;
    }
    return b;
}

__global__ void apply_mask_kernel(float* x, float* mask, int n, int c, int h, Box bx) {
    const size_t hw = (size_t)h * 4 * h;
    const size_t total = (size_t)n * hw;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t img = idx / hw, p = idx - img * hw;
        const int y = (int)(p / (4 * h)), xx = (int)(p - (size_t)y * 4 * h);
        const float m = (y >= bx.y0 && y < bx.y1 && xx >= bx.x0 && xx < bx.x1) ? 1.0f : 0.0f;
        if (mask) mask[idx] = m;
        for (int ch = 0; ch < c; ++ch) { float* q = x + (img * c + ch) * hw + p; *q = *q * m; }
    }
}

__global__ void build_view_kernel(const float* __restrict__ rgb, const float* __restrict__ nrm, const float* __restrict__ dep,
                                  float* __restrict__ view, int n, int h, Box bx) {
    const size_t hw = (size_t)h * 4 * h;
    const size_t total = (size_t)n * hw;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t img = idx / hw, p = idx - img * hw;
        const int y = (int)(p / (4 * h)), xx = (int)(p - (size_t)y * 4 * h);
        const float m = (y >= bx.y0 && y < bx.y1 && xx >= bx.x0 && xx < bx.x1) ? 1.0f : 0.0f;
        float* v = view + img * 8 * hw + p;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            v[ch * hw] = rgb[(img * 3 + ch) * hw + p] * m;
            v[(3 + ch) * hw] = nrm[(img * 3 + ch) * hw + p] * m;
        }
        const float d = dep[img * hw + p] * m;
        v[6 * hw] = d;
        v[7 * hw] = (d != 0.0f) ? 1.0f : 0.0f;
    }
}

// util.py:763-771: (x,y,-z) = ((u/h-.5)*2*z, (.5-v/h)*2*z, -z), rotated by the face rotation.
__global__ void pano2pc_kernel(const float* __restrict__ depth, double* __restrict__ pc, uint8_t* __restrict__ valid,
                               int n, int h, int dataset) {
    const size_t hw = (size_t)h * 4 * h;
    const size_t total = (size_t)n * hw;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t img = idx / hw, q = idx - img * hw;          // q = face-major point index
        const int face = (int)(q / ((size_t)h * h));
        const int r = (int)(q - (size_t)face * h * h);
        const int v = r / h, u = r - v * h;
        const double z = (double)depth[img * hw + (size_t)v * 4 * h + face * h + u];
        double xs = ((double)u / h - 0.5) * 2, ys = (0.5 - (double)v / h) * 2;
        double x, y;
        if (dataset == RELPOSE_SCANNET) { y = ys * z / (1.1895 * 2); x = xs * z / (0.8921875 * 2); }
        else { y = ys * z; x = xs * z; }
        double ox, oy, oz;
        face_rot(face_index(dataset, face), x, y, -z, ox, oy, oz);
        double* o = pc + img * 3 * hw;
        o[q] = ox; o[hw + q] = oy; o[2 * hw + q] = oz;
        if (valid) valid[idx] = (dataset == RELPOSE_SCANNET) ? (z != 0.0) : 1;
    }
}

// ---- warp ------------------------------------------------------------------------------------
struct WarpSrc { int npts, pw; int y0, x0; };   // observed block: pw columns starting at (y0,x0)

__host__ __device__ inline WarpSrc warp_src(int dataset, int h) {
    WarpSrc s;
    if (dataset == RELPOSE_SCANNET) {
        Box b = observed_box(RELPOSE_MASK_KINECT, h);
        s.y0 = b.y0; s.x0 = b.x0; s.pw = b.x1 - b.x0; s.npts = (b.y1 - b.y0) * s.pw;
    } else { s.y0 = 0; s.x0 = h; s.pw = h; s.npts = h * h; }
    return s;
}

// Source point p of image `img` in the panorama frame, then moved by the pose.  Returns false if
// the reference drops the point (depth == 0 for matterport / scannet; suncg keeps everything).
__device__ __forceinline__ bool warp_point(const float* view, const double* T, int h, int dataset, const WarpSrc& s, int p,
                                           double* q, int& py, int& px) {
    const size_t hw = (size_t)h * 4 * h;
    const int v = p / s.pw, u = p - v * s.pw;
    py = s.y0 + v; px = s.x0 + u;
    const double z = (double)view[6 * hw + (size_t)py * 4 * h + px];
    if (dataset != RELPOSE_SUNCG && z == 0.0) return false;
    double x, y, X, Y, Z;
    if (dataset == RELPOSE_SCANNET) {
        const int ph = s.npts / s.pw;
        const double xs = ((double)u / s.pw - 0.5) * 2, ys = (0.5 - (double)v / ph) * 2;
        x = (xs * z) * s.pw / 160; y = (ys * z) * ph / 160;      // util.py:519-521
        X = x; Y = y; Z = -z;
    } else {
        const double xs = ((double)u / h - 0.5) * 2, ys = (0.5 - (double)v / h) * 2;
        x = xs * z; y = ys * z;
        if (dataset == RELPOSE_SUNCG) face_rot(1, x, y, -z, X, Y, Z);   // observed face = slot 1, Rs[1]
        else { X = x; Y = y; Z = -z; }                                  // matterport: Rs[(1-1)%4] = I
    }
    // np.matmul(R44, [p;1])[:3]
#pragma unroll
    for (int a = 0; a < 3; ++a) q[a] = ((T[a * 4 + 0] * X + T[a * 4 + 1] * Y) + T[a * 4 + 2] * Z) + T[a * 4 + 3] * 1.0;
    return true;
}

// Projection of a moved point into face slot `slot` (reproj_helper).  Returns pixel or -1.
__device__ __forceinline__ int warp_project(const double* q, int h, int dataset, int slot, double& depth_out) {
    double x, y, z;
    face_rot_t(face_index(dataset, slot), q[0], q[1], q[2], x, y, z);
    const double az = fabs(z) + 1e-32;
    const double xn = x / az, yn = y / az;
    if (!((z < 0) && (fabs(xn) < 1) && (fabs(yn) < 1))) return -1;
    double cx = (xn + 1) * 0.5 * h, cy = (1 - yn) * 0.5 * h;
    cx = rint(cx); cy = rint(cy);                       // np.round = round-half-even
    cx = cx < 0 ? 0 : (cx > h - 1 ? h - 1 : cx);
    cy = cy < 0 ? 0 : (cy > h - 1 ? h - 1 : cy);
    depth_out = -z;
    return (int)cy * 4 * h + slot * h + (int)cx;
}

__device__ __forceinline__ bool pose_is_identity(const double* T) {
    bool id = true;
#pragma unroll
    for (int a = 0; a < 16; ++a) id = id && (T[a] == ((a % 5 == 0) ? 1.0 : 0.0));
    return id;
}

// pass 1: every source point claims its target pixel in each face with atomicMax(point index + 1):
// numpy fancy assignment = the last point in point order wins (util.py:603-608).
// The source view of output image `img` is image (img ^ swap) of `view`, whose images are `vstride` floats apart
// (swap = 1, vstride = 16*hw: the partner's own-view channels of the in-place network input, relpose_warp_pairs).
__global__ void warp_scatter_kernel(const float* __restrict__ view, const double* __restrict__ pose, int* __restrict__ keys,
                                    int n, int h, int dataset, WarpSrc s, size_t vstride, int swap) {
    const int img = blockIdx.y;
    const double* T = pose + (size_t)img * 16;
    if (pose_is_identity(T)) return;
    const size_t hw = (size_t)h * 4 * h;
    const float* vw = view + (size_t)(img ^ swap) * vstride;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < s.npts; p += gridDim.x * blockDim.x) {
        double q[3]; int py, px;
        if (!warp_point(vw, T, h, dataset, s, p, q, py, px)) continue;
#pragma unroll
        for (int slot = 0; slot < 4; ++slot) {
            double dz;
            const int pix = warp_project(q, h, dataset, slot, dz);
            if (pix >= 0) atomicMax(&keys[(size_t)img * hw + pix], p + 1);
        }
    }
}

// pass 2: every output pixel gathers from the winning point -- and resets the key it consumed, so that the key image is all zeros
// again when the call returns (a caller that keeps its workspace never needs the 4 n h 4h-byte memset in front of the scatter pass:
// relpose_warp_pairs2 with RELPOSE_WARP_KEYS_CLEAN).
__global__ void warp_gather_kernel(const float* view, const double* __restrict__ pose, int* __restrict__ keys,
                                   float* out, int n, int h, int dataset, WarpSrc s, size_t vstride, int swap, size_t ostride) {
    const int img = blockIdx.y;
    const double* T = pose + (size_t)img * 16;
    const bool ident = pose_is_identity(T);
    const size_t hw = (size_t)h * 4 * h;
    const float* vw = view + (size_t)(img ^ swap) * vstride;
    float* o = out + (size_t)img * ostride;
    for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < (int)hw; pix += gridDim.x * blockDim.x) {
        float r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int key = ident ? 0 : keys[(size_t)img * hw + pix];
        if (key > 0) {
            keys[(size_t)img * hw + pix] = 0;
            const int p = key - 1;
            double q[3]; int py, px;
            warp_point(vw, T, h, dataset, s, p, q, py, px);
            const int xx = pix % (4 * h);
            double dz;
            warp_project(q, h, dataset, xx / h, dz);
            const size_t sp = (size_t)py * 4 * h + px;
            const double n0 = (double)vw[3 * hw + sp], n1 = (double)vw[4 * hw + sp], n2 = (double)vw[5 * hw + sp];
            r[0] = vw[sp]; r[1] = vw[hw + sp]; r[2] = vw[2 * hw + sp];
#pragma unroll
            for (int a = 0; a < 3; ++a) r[3 + a] = (float)((T[a * 4 + 0] * n0 + T[a * 4 + 1] * n1) + T[a * 4 + 2] * n2);
            r[6] = (float)dz;
            r[7] = (dz != 0.0) ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) o[ch * hw + pix] = r[ch];
    }
}

__global__ void pose_inverse_kernel(const double* __restrict__ pose, double* __restrict__ inv, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a[16], o[16];
    for (int k = 0; k < 16; ++k) a[k] = pose[(size_t)i * 16 + k];
    if (!rp_inv4(a, o)) for (int k = 0; k < 16; ++k) o[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int k = 0; k < 16; ++k) inv[(size_t)i * 16 + k] = o[k];
}

// ---- compose + sample ------------------------------------------------------------------------
// evaluation.py:248-253 in float32 exactly as numpy evaluates it:
//   normal = ((1-m)*f_n + m*obs_n) / (||obs_n|| + 1e-6) ;  depth = (1-m)*f_d + m*obs_d
// compose = 1 is the library form of the same loop (rpmodule.py:629-636): the blended normal is divided by ITS OWN norm + 1e-12.
__device__ __forceinline__ void composed_pixel(const float* f, int cf, const float* on, const float* od, size_t hw, int h, Box bx,
                                               int compose, int y, int x, float* nout, float& dout) {
    const size_t p = (size_t)y * 4 * h + x;
    const float m = (y >= bx.y0 && y < bx.y1 && x >= bx.x0 && x < bx.x1) ? 1.0f : 0.0f;
    const float om = 1.0f - m;
    const float o0 = on[p], o1 = on[hw + p], o2 = on[2 * hw + p];
    const float b0 = om * f[3 * hw + p] + m * o0, b1 = om * f[4 * hw + p] + m * o1, b2 = om * f[5 * hw + p] + m * o2;
    const float nn = compose ? sqrtf((b0 * b0 + b1 * b1) + b2 * b2) + 1e-12f : sqrtf((o0 * o0 + o1 * o1) + o2 * o2) + 1e-6f;
    nout[0] = b0 / nn;
    nout[1] = b1 / nn;
    nout[2] = b2 / nn;
    dout = om * f[6 * hw + p] + m * od[p];
}

__global__ void sample_primitives_kernel(const float* __restrict__ f, int cf, int feat_off, const float* __restrict__ obs_norm,
                                         const float* __restrict__ obs_depth, const double* __restrict__ pts,
                                         const int* __restrict__ npts, int npts_max, double* __restrict__ pc,
                                         double* __restrict__ normal, float* __restrict__ feat, int n, int h, Box bx, int dataset,
                                         int compose) {
    const int img = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= npts[img]) return;
    const size_t hw = (size_t)h * 4 * h;
    const int W = 4 * h;
    const float* fi = f + (size_t)img * cf * hw;
    const float* on = obs_norm + (size_t)img * 3 * hw;
    const float* od = obs_depth + (size_t)img * hw;
    const double px = pts[((size_t)img * npts_max + k) * 2 + 0], py = pts[((size_t)img * npts_max + k) * 2 + 1];
    // rputil.getPixel :88-119 (float64 bilinear of the float32 composed maps)
    const int tx = (int)floor(px), ty = (int)floor(py);
    const double fx1 = px - tx, fx0 = tx + 1 - px, fy1 = py - ty, fy0 = ty + 1 - py;
    float n00[3], n01[3], n10[3], n11[3], d00, d01, d10, d11;
    composed_pixel(fi, cf, on, od, hw, h, bx, compose, ty, tx, n00, d00);
    composed_pixel(fi, cf, on, od, hw, h, bx, compose, ty, tx + 1, n01, d01);
    composed_pixel(fi, cf, on, od, hw, h, bx, compose, ty + 1, tx, n10, d10);
    composed_pixel(fi, cf, on, od, hw, h, bx, compose, ty + 1, tx + 1, n11, d11);
    const double val = (((double)d00 * fy0 * fx0 + (double)d01 * fx1 * fy0) + (double)d10 * fy1 * fx0) + (double)d11 * fx1 * fy1;
    double nn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
        nn[a] = (((double)n00[a] * fy0 * fx0 + (double)n01[a] * fx1 * fy0) + (double)n10[a] * fy1 * fx0) + (double)n11[a] * fx1 * fy1;
    const double nl = rp_norm3(nn[0], nn[1], nn[2]);
    double* no = normal + ((size_t)img * npts_max + k) * 3;
    no[0] = nn[0] / nl; no[1] = nn[1] / nl; no[2] = nn[2] / nl;
    // getPixel_helper :61-86
    const int slot = (int)floor(px / h);          // xs // H
    const double ystp = (0.5 - py / h) * 2, xstp = ((px - slot * h) / h - 0.5) * 2;
    double ox, oy, oz;
    face_rot(face_index(dataset, slot), xstp * val, ystp * val, -val, ox, oy, oz);
    double* po = pc + ((size_t)img * npts_max + k) * 3;
    po[0] = ox; po[1] = oy; po[2] = oz;
    // rputil.interpolate :43-58 in float32: pt = (x/W, y/H) cast to float32, x = pt*(W-1)
    const float ptx = (float)(px / W), pty = (float)(py / h);
    const float x = ptx * (float)(W - 1), y = pty * (float)(h - 1);
    const float x0 = floorf(x), y0 = floorf(y);
    const int xi = (int)x0, yi = (int)y0;
    const float wx0 = x0 + 1.0f - x, wy0 = y0 + 1.0f - y, wx1 = x - x0, wy1 = y - y0;
    const float* ff = fi + (size_t)feat_off * hw;
    float* fo = feat + ((size_t)img * npts_max + k) * 32;
    for (int c = 0; c < 32; ++c) {
        const float* fc = ff + (size_t)c * hw;
        const float v00 = fc[(size_t)yi * W + xi], v10 = fc[(size_t)(yi + 1) * W + xi];
        const float v01 = fc[(size_t)yi * W + xi + 1], v11 = fc[(size_t)(yi + 1) * W + xi + 1];
        fo[c] = ((v00 * wx0 * wy0 + v10 * wx0 * wy1) + v01 * wx1 * wy0) + v11 * wx1 * wy1;
    }
}

// ---- stand-alone rputil.getPixel / rputil.interpolate (the reference-named shims) ---------------------------------
// rputil.getPixel :88-119 + getPixel_helper :61-86 on caller-composed maps: depth [h,4h] f64, normal [h,4h,3] f64 (HWC like
// the reference's numpy arrays), pts [k,2] f64 pixel coords -> pc [k,3] (the reference returns its transpose), nn [k,3].
__global__ void get_pixel_kernel(const double* __restrict__ depth, const double* __restrict__ normal, const double* __restrict__ pts,
                                 int k_total, int h, int dataset, double* __restrict__ pc, double* __restrict__ nn_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= k_total) return;
    const int W = 4 * h;
    const double px = pts[(size_t)k * 2 + 0], py = pts[(size_t)k * 2 + 1];
    const int tx = (int)floor(px), ty = (int)floor(py);
    if (!(tx >= 0 && tx + 1 < W && ty >= 0 && ty + 1 < h)) {
        // outside the reference's precondition x <= W-2, y <= H-2 (rputil.py:194-203; numpy raises IndexError there): no read past the
        // maps, NaN outputs; the reference-named shim raises IndexError before launching
        for (int a = 0; a < 3; ++a) { nn_out[(size_t)k * 3 + a] = NAN; pc[(size_t)k * 3 + a] = NAN; }
        return;
    }
    const double fx1 = px - tx, fx0 = tx + 1 - px, fy1 = py - ty, fy0 = ty + 1 - py;
    const size_t p00 = (size_t)ty * W + tx, p01 = p00 + 1, p10 = p00 + W, p11 = p10 + 1;
    const double val = ((depth[p00] * fy0 * fx0 + depth[p01] * fx1 * fy0) + depth[p10] * fy1 * fx0) + depth[p11] * fx1 * fy1;
    double nn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
        nn[a] = ((normal[p00 * 3 + a] * fy0 * fx0 + normal[p01 * 3 + a] * fx1 * fy0) + normal[p10 * 3 + a] * fy1 * fx0) +
                normal[p11 * 3 + a] * fx1 * fy1;
    const double nl = rp_norm3(nn[0], nn[1], nn[2]);
    nn_out[(size_t)k * 3 + 0] = nn[0] / nl; nn_out[(size_t)k * 3 + 1] = nn[1] / nl; nn_out[(size_t)k * 3 + 2] = nn[2] / nl;
    const int slot = (int)floor(px / h);
    const double ystp = (0.5 - py / h) * 2, xstp = ((px - slot * h) / h - 0.5) * 2;
    double ox, oy, oz;
    face_rot(face_index(dataset, slot), xstp * val, ystp * val, -val, ox, oy, oz);
    pc[(size_t)k * 3 + 0] = ox; pc[(size_t)k * 3 + 1] = oy; pc[(size_t)k * 3 + 2] = oz;
}

// rputil.interpolate :43-58 (float32 like torch): feat [c,h,w], pt [k,2] normalised -> out [c,k]
__global__ void interpolate_kernel(const float* __restrict__ feat, const float* __restrict__ pt, float* __restrict__ out, int c_total,
                                   int h, int w, int k_total) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= k_total) return;
    const float x = pt[(size_t)k * 2 + 0] * (float)(w - 1), y = pt[(size_t)k * 2 + 1] * (float)(h - 1);
    const float x0 = floorf(x), y0 = floorf(y);
    const int xi = (int)x0, yi = (int)y0;
    const float wx0 = x0 + 1.0f - x, wy0 = y0 + 1.0f - y, wx1 = x - x0, wy1 = y - y0;
    if (!(xi >= 0 && xi + 1 < w && yi >= 0 && yi + 1 < h)) {      // pt outside [0, 1): torch raises IndexError in the reference (rputil.py:52-55)
        for (int c = 0; c < c_total; ++c) out[(size_t)c * k_total + k] = NAN;
        return;
    }
    for (int c = 0; c < c_total; ++c) {
        const float* fc = feat + (size_t)c * h * w;
        const float v00 = fc[(size_t)yi * w + xi], v10 = fc[(size_t)(yi + 1) * w + xi];
        const float v01 = fc[(size_t)yi * w + xi + 1], v11 = fc[(size_t)(yi + 1) * w + xi + 1];
        out[(size_t)c * k_total + k] = ((v00 * wx0 * wy0 + v10 * wx0 * wy1) + v01 * wx1 * wy0) + v11 * wx1 * wy1;
    }
}

// ---- evaluation-side statistics: util.parse_data / point_cloud_overlap -----------------------
// depth2pc (util.py:468-523) of the observed block of each panorama: points in pixel order + validity
// (depth != 0); the same unprojection the warp uses (warp_point with the identity pose).
__global__ void depth2pc_kernel(const float* __restrict__ depth, double* __restrict__ pc, uint8_t* __restrict__ valid,
                                int n, int h, int dataset, WarpSrc s) {
    const int img = blockIdx.y;
    const size_t hw = (size_t)h * 4 * h;
    const float* d = depth + (size_t)img * hw;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < s.npts; p += gridDim.x * blockDim.x) {
        const int v = p / s.pw, u = p - v * s.pw;
        const double z = (double)d[(size_t)(s.y0 + v) * 4 * h + s.x0 + u];
        double X, Y, Z;
        if (dataset == RELPOSE_SCANNET) {
            const int ph = s.npts / s.pw;
            const double xs = ((double)u / s.pw - 0.5) * 2, ys = (0.5 - (double)v / ph) * 2;
            X = (xs * z) * s.pw / 160; Y = (ys * z) * ph / 160; Z = -z;
        } else {
            const double xs = ((double)u / h - 0.5) * 2, ys = (0.5 - (double)v / h) * 2;
            if (dataset == RELPOSE_SUNCG) face_rot(1, xs * z, ys * z, -z, X, Y, Z);
            else { X = xs * z; Y = ys * z; Z = -z; }
        }
        double* o = pc + ((size_t)img * s.npts + p) * 3;
        o[0] = X; o[1] = Y; o[2] = Z;
        valid[(size_t)img * s.npts + p] = (z != 0.0);
    }
}

// util.depth2pc's full-resolution kinect branch (util.py:497-507): a 480 x 640 depth IMAGE (not a block of a panorama) back-projected with
// the ScanNet intrinsics folded into the two divisors; feeds the baselines' point clouds through util.parse_data :79-90.
__global__ void depth2pc_full_kernel(const float* __restrict__ depth, double* __restrict__ pc, uint8_t* __restrict__ valid, int hh, int ww) {
    const int img = blockIdx.y, np_ = hh * ww;
    const float* d = depth + (size_t)img * np_;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < np_; p += gridDim.x * blockDim.x) {
        const int v = p / ww, u = p - v * ww;
        const double z = (double)d[p];
        const double xs = ((double)u / ww - 0.5) * 2, ys = (0.5 - (double)v / hh) * 2;
        double* o = pc + ((size_t)img * np_ + p) * 3;
        o[0] = (xs * z) / (0.8921875 * 2); o[1] = (ys * z) / (1.1895 * 2); o[2] = -z;
        valid[(size_t)img * np_ + p] = (z != 0.0);
    }
}

// Nearest-neighbour distance of every (optionally rigidly moved) query point to a reference set: brute force,
// reference points staged through LDS in tiles of 1024.  Replaces the sklearn KDTree queries of
// util.point_cloud_overlap (util.py:21-40); distances are sqrt((dx^2+dy^2)+dz^2) like KDTree's metric.
__global__ __launch_bounds__(256) void nn_dist_kernel(const double* __restrict__ q, const uint8_t* __restrict__ qv, int nq,
                                                       const double* __restrict__ r, const uint8_t* __restrict__ rv, int nr,
                                                       const double* __restrict__ T, double* __restrict__ out) {
    __shared__ double rs[1024 * 3];
    __shared__ uint8_t rok[1024];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double x = 0, y = 0, z = 0;
    if (i < nq) {
        const double a = q[(size_t)i * 3], b = q[(size_t)i * 3 + 1], c = q[(size_t)i * 3 + 2];
        if (T) {      // np.matmul(R[:3,:3], pc.T) + R[:3,3:4]
            x = ((T[0] * a + T[1] * b) + T[2] * c) + T[3];
            y = ((T[4] * a + T[5] * b) + T[6] * c) + T[7];
            z = ((T[8] * a + T[9] * b) + T[10] * c) + T[11];
        } else { x = a; y = b; z = c; }
    }
    double best = INFINITY;
    for (int t0 = 0; t0 < nr; t0 += 1024) {
        __syncthreads();
        for (int k = threadIdx.x; k < 1024 * 3; k += 256) rs[k] = (t0 * 3 + k < nr * 3) ? r[(size_t)t0 * 3 + k] : 0.0;
        for (int k = threadIdx.x; k < 1024; k += 256) rok[k] = (t0 + k < nr) && (!rv || rv[t0 + k]);
        __syncthreads();
        const int lim = min(1024, nr - t0);
        for (int k = 0; k < lim; ++k) {
            if (!rok[k]) continue;
            const double dx = x - rs[k * 3], dy = y - rs[k * 3 + 1], dz = z - rs[k * 3 + 2];
            const double d2 = (dx * dx + dy * dy) + dz * dz;
            best = d2 < best ? d2 : best;
        }
    }
    if (i < nq) out[i] = (!qv || qv[i]) ? sqrt(best) : -1.0;
}

// ---- feature-guided keypoint augmentation: rputil.getKeypoint :182-190, Sampling :355-371 -----
// dist[s, p] = sum_c (q[s][c] - feat[c][p])^2, float32, for n_sel query descriptors over the H*W map.
// HBM-bound: every thread owns one pixel, reads its 32 channels once (coalesced per channel plane) and
// produces all n_sel distances (queries in LDS).
__global__ __launch_bounds__(256) void feature_distance_kernel(const float* __restrict__ q, const float* __restrict__ feat,
                                                                float* __restrict__ dist, int nsel, int hw) {
    extern __shared__ float qs[];                 // [nsel][32]
    for (int i = threadIdx.x; i < nsel * 32; i += 256) qs[i] = q[i];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    float f[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) f[c] = feat[(size_t)c * hw + p];
    for (int s = 0; s < nsel; ++s) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) { const float d = qs[s * 32 + c] - f[c]; acc = acc + d * d; }
        dist[(size_t)s * hw + p] = acc;
    }
}

// Sampling: per map, K times {argmax of exp(-dist/2) (first index on ties), suppress the window around it with the
// map's minimum}.  One workgroup per map; the heat map is evaluated on the fly, suppression is a list of windows.
__global__ __launch_bounds__(1024) void nms_sampling_kernel(const float* __restrict__ dist, double* __restrict__ pts, int H, int W, int K,
                                                             int win) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    __shared__ int wy0[8], wy1[8], wx0[8], wx1[8];
    const int map = blockIdx.x, hw = H * W;
    const float* dm = dist + (size_t)map * hw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < K; ++k) {
        float best = -1.f;                         // heat values are in (0, 1]
        int bidx = INT_MAX;
        for (int p = threadIdx.x; p < hw; p += blockDim.x) {
            const int y = p / W, x = p - y * W;
            bool sup = false;
            for (int j = 0; j < k; ++j) sup = sup || (y >= wy0[j] && y < wy1[j] && x >= wx0[j] && x < wx1[j]);
            // suppressed pixels hold the map minimum, which can only win if the whole map is suppressed
            const float v = sup ? -0.5f : expf(-dm[p] / 2);
            if (v > best) { best = v; bidx = p; }  // p increases per thread: first index kept on ties
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bidx, m, 64);
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        if (lane == 0) { bv[wave] = best; bi[wave] = bidx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float b = bv[0]; int id = bi[0];
            for (int w = 1; w < (int)(blockDim.x >> 6); ++w) if (bv[w] > b || (bv[w] == b && bi[w] < id)) { b = bv[w]; id = bi[w]; }
            if (!(b > 0.f)) id = 0;                // every remaining heat underflowed to 0 = the value of the suppressed pixels: np.argmax returns pixel 0
            const int y = id / W, x = id - y * W;
            pts[((size_t)map * K + k) * 2 + 0] = x;
            pts[((size_t)map * K + k) * 2 + 1] = y;
            // heatmap[i][topl[1]:botr[1], topl[0]:botr[0]] = min   (rputil.py:368-370; end exclusive)
            wy0[k] = max(0, y - win); wy1[k] = min(H - 1, y + win);
            wx0[k] = max(0, x - win); wx1[k] = min(W - 1, x + win);
        }
        __syncthreads();
    }
}

inline int grid_for(size_t total, int block = 256, int cap = 4096) {
    size_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

}  // namespace

extern "C" {

int relpose_apply_mask(float* x, float* mask, int32_t n, int32_t c, int32_t h, int32_t method, void* stream) {
    if (!x || n <= 0 || c <= 0 || h <= 0 || method < 0 || method > 1) return RELPOSE_EINVAL;
    hipLaunchKernelGGL(apply_mask_kernel, dim3(grid_for((size_t)n * h * 4 * h)), dim3(256), 0, (hipStream_t)stream, x, mask, n, c, h,
                       observed_box(method, h));
    RP_CHECK_LAUNCH();
    return 0;
}

int relpose_build_view(const float* rgb, const float* norm, const float* depth, float* view, int32_t n, int32_t h, int32_t method,
                       void* stream) {
    if (!rgb || !norm || !depth || !view || n <= 0 || h <= 0 || method < 0 || method > 1) return RELPOSE_EINVAL;
    hipLaunchKernelGGL(build_view_kernel, dim3(grid_for((size_t)n * h * 4 * h)), dim3(256), 0, (hipStream_t)stream, rgb, norm, depth,
                       view, n, h, observed_box(method, h));
    RP_CHECK_LAUNCH();
    return 0;
}

int relpose_pano2pc(const float* depth, double* pc, uint8_t* valid, int32_t n, int32_t h, int32_t dataset, void* stream) {
    if (!depth || !pc || n <= 0 || h <= 0 || dataset < 0 || dataset > 2) return RELPOSE_EINVAL;
    hipLaunchKernelGGL(pano2pc_kernel, dim3(grid_for((size_t)n * h * 4 * h, 256, 8192)), dim3(256), 0, (hipStream_t)stream, depth, pc,
                       valid, n, h, dataset);
    RP_CHECK_LAUNCH();
    return 0;
}

size_t relpose_warp_workspace_bytes(int32_t n, int32_t h) { return (n <= 0 || h <= 0) ? 0 : rp_align((size_t)n * h * 4 * h * 4); }

static int launch_warp(const float* view, size_t vstride, int swap, const double* pose, float* out, size_t ostride, void* workspace,
                       int32_t n, int32_t h, int32_t dataset, hipStream_t s, bool keys_clean = false) {
    const size_t hw = (size_t)h * 4 * h;
    int* keys = (int*)workspace;
    if (!keys_clean) RP_HIP(hipMemsetAsync(keys, 0, (size_t)n * hw * 4, s));
    const WarpSrc src = warp_src(dataset, h);
    hipLaunchKernelGGL(warp_scatter_kernel, dim3((src.npts + 255) / 256, n), dim3(256), 0, s, view, pose, keys, n, h, dataset, src, vstride,
                       swap);
    RP_CHECK_LAUNCH();
    hipLaunchKernelGGL(warp_gather_kernel, dim3((int)((hw + 255) / 256), n), dim3(256), 0, s, view, pose, keys, out, n, h, dataset, src,
                       vstride, swap, ostride);
    RP_CHECK_LAUNCH();
    return 0;
}

int relpose_warp(const float* view, const double* pose, float* out, void* workspace, int32_t n, int32_t h, int32_t dataset, void* stream) {
    if (!view || !pose || !out || !workspace || n <= 0 || h <= 0 || dataset < 0 || dataset > 2) return RELPOSE_EINVAL;
    const size_t hw = (size_t)h * 4 * h;
    return launch_warp(view, 8 * hw, 0, pose, out, 8 * hw, workspace, n, h, dataset, (hipStream_t)stream);
}

int relpose_warp_pairs(float* x, const double* pose, void* workspace, int32_t n, int32_t h, int32_t dataset, void* stream) {
    if (!x || !pose || !workspace || n <= 0 || (n & 1) || h <= 0 || dataset < 0 || dataset > 2) return RELPOSE_EINVAL;
    const size_t hw = (size_t)h * 4 * h;
    return launch_warp(x, 16 * hw, 1, pose, x + 8 * hw, 16 * hw, workspace, n, h, dataset, (hipStream_t)stream);
}

int relpose_warp_pairs2(float* x, const double* pose, void* workspace, int32_t n, int32_t h, int32_t dataset, int32_t flags, void* stream) {
    if (!x || !pose || !workspace || n <= 0 || (n & 1) || h <= 0 || dataset < 0 || dataset > 2 || (flags & ~RELPOSE_WARP_KEYS_CLEAN)) return RELPOSE_EINVAL;
    const size_t hw = (size_t)h * 4 * h;
    return launch_warp(x, 16 * hw, 1, pose, x + 8 * hw, 16 * hw, workspace, n, h, dataset, (hipStream_t)stream, (flags & RELPOSE_WARP_KEYS_CLEAN) != 0);
}

int relpose_pose_inverse(const double* pose, double* inv, int32_t n, void* stream) {
    if (!pose || !inv || n <= 0) return RELPOSE_EINVAL;
    hipLaunchKernelGGL(pose_inverse_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose, inv, n);
    RP_CHECK_LAUNCH();
    return 0;
}

int relpose_sample_primitives(const float* f, int32_t cf, int32_t feat_off, const float* obs_norm, const float* obs_depth,
                              const double* pts, const int32_t* npts, int32_t npts_max, double* pc, double* normal, float* feat,
                              int32_t n, int32_t h, int32_t mask_method, int32_t compose, int32_t dataset, void* stream) {
    if (!f || !obs_norm || !obs_depth || !pts || !npts || !pc || !normal || !feat || n <= 0 || h <= 0 || npts_max <= 0 ||
        cf < 7 || feat_off < 7 || feat_off + 32 > cf || mask_method < 0 || mask_method > 1 || compose < 0 || compose > 1 ||
        dataset < 0 || dataset > 2)
        return RELPOSE_EINVAL;
    hipLaunchKernelGGL(sample_primitives_kernel, dim3((npts_max + 63) / 64, n), dim3(64), 0, (hipStream_t)stream, f, cf, feat_off,
                       obs_norm, obs_depth, pts, npts, npts_max, pc, normal, feat, n, h, observed_box(mask_method, h), dataset, compose);
    RP_CHECK_LAUNCH();
    return 0;
}


int relpose_depth2pc(const float* depth, double* pc, uint8_t* valid, int32_t n, int32_t h, int32_t dataset, void* stream) {
    if (!depth || !pc || !valid || n <= 0 || h <= 0 || dataset < 0 || dataset > 2) return RELPOSE_EINVAL;
    const WarpSrc src = warp_src(dataset, h);
    hipLaunchKernelGGL(depth2pc_kernel, dim3((src.npts + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, depth, pc, valid, n, h, dataset, src);
    RP_CHECK_LAUNCH();
    return 0;
}

int relpose_depth2pc_full(const float* depth, double* pc, uint8_t* valid, int32_t n, int32_t hh, int32_t ww, void* stream) {
    // (the reference defines this branch for exactly one shape, util.py:498; any other leaves its `pc` unbound)
    if (!depth || !pc || !valid || n <= 0 || hh != 480 || ww != 640) return RELPOSE_EINVAL;
    hipLaunchKernelGGL(depth2pc_full_kernel, dim3((hh * ww + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, depth, pc, valid, hh, ww);
    RP_CHECK_LAUNCH();
    return 0;
}

int relpose_get_pixel(const double* depth, const double* normal, const double* pts, int32_t k, int32_t h, int32_t dataset, double* pc,
                      double* nn, void* stream) {
    if (!depth || !normal || !pts || !pc || !nn || k <= 0 || h <= 0 || dataset < 0 || dataset > 2) return RELPOSE_EINVAL;
    hipLaunchKernelGGL(get_pixel_kernel, dim3((k + 63) / 64), dim3(64), 0, (hipStream_t)stream, depth, normal, pts, k, h, dataset, pc, nn);
    RP_CHECK_LAUNCH();
    return 0;
}

int relpose_interpolate(const float* feat, const float* pt, float* out, int32_t c, int32_t h, int32_t w, int32_t k, void* stream) {
    if (!feat || !pt || !out || c <= 0 || h < 2 || w < 2 || k <= 0) return RELPOSE_EINVAL;
    hipLaunchKernelGGL(interpolate_kernel, dim3((k + 63) / 64), dim3(64), 0, (hipStream_t)stream, feat, pt, out, c, h, w, k);
    RP_CHECK_LAUNCH();
    return 0;
}

int32_t relpose_observed_points(int32_t h, int32_t dataset) { return (h > 0 && dataset >= 0 && dataset <= 2) ? warp_src(dataset, h).npts : 0; }

int relpose_nn_dist(const double* query, const uint8_t* query_valid, int32_t nq, const double* ref, const uint8_t* ref_valid, int32_t nr,
                    const double* pose, double* dist, void* stream) {
    if (!query || !ref || !dist || nq <= 0 || nr <= 0) return RELPOSE_EINVAL;
    hipLaunchKernelGGL(nn_dist_kernel, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, query, query_valid, nq, ref, ref_valid, nr,
                       pose, dist);
    RP_CHECK_LAUNCH();
    return 0;
}


int relpose_feature_distance_map(const float* query, const float* feat, float* dist, int32_t nsel, int32_t H, int32_t W, void* stream) {
    if (!query || !feat || !dist || nsel <= 0 || nsel > 256 || H <= 0 || W <= 0) return RELPOSE_EINVAL;
    const int hw = H * W;
    hipLaunchKernelGGL(feature_distance_kernel, dim3((hw + 255) / 256), dim3(256), (size_t)nsel * 32 * sizeof(float), (hipStream_t)stream,
                       query, feat, dist, nsel, hw);
    RP_CHECK_LAUNCH();
    return 0;
}

int relpose_nms_sampling(const float* dist, double* pts, int32_t nmaps, int32_t H, int32_t W, int32_t K, int32_t window, void* stream) {
    if (!dist || !pts || nmaps <= 0 || H <= 0 || W <= 0 || K < 1 || K > 8 || window < 0) return RELPOSE_EINVAL;
    hipLaunchKernelGGL(nms_sampling_kernel, dim3(nmaps), dim3(1024), 0, (hipStream_t)stream, dist, pts, H, W, K, window);
    RP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
