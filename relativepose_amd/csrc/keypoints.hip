// Per-level keypoint derivation of the reference, batched over the views of a batch of scan pairs and kept on the device
// (rputil.getKeypoint :141-237 / getKeypoint_kinect :240-353, called once per recurrent level: evaluation.py:278 ->
// rpmodule.getMatchingPrimitive :511-533).  Everything BEHIND the SIFT detector: descriptors at the selected detections / random points
// (interpolate :43-58), for every query descriptor the TOPK non-maximum-suppressed minima of its squared distance to every pixel of the
// OTHER view's feature map (:182-190, :212-214, Sampling :355-371), the validity filter, the concatenation and the weights 1 / 0.99.
// All np.random draws of getKeypoint are independent of the features, so the host pre-draws them per (pair, level) in the reference's
// call order (rputil.keypoint_plan) and hands over query points + slot tables; the feature-dependent part runs here.
//
// The [n_query, H*W] distance maps are never materialised (the reference builds three [30, 102400] float32 maps per pair and level):
//   kp_desc_kernel       descriptor of every query point (bilinear, float32 like torch)
//   kp_tile_best_kernel  one pass over a view's feature map: every 32x16-pixel tile keeps, per query, its best (heat, first index) -- heat =
//                        expf(-d/2) in float32 exactly like Sampling's np.exp(-heatmap/2), d summed over the 32 channels in order
//   kp_pick_kernel       per query: the TOPK picks from the tile bests; only tiles whose best lies inside a suppression window are redone
//   kp_assemble_kernel   per view: slot table (host coordinates / picks) -> compacted keypoint list + weights + count
// Bound: VALU (96 separately rounded float32 operations per pixel and query; the feature map is read once: 13 MB per view at 160x640).
//
// Compiled with -ffp-contract=off: the float32 distance and the bilinear gather must round like torch (no FMA fusion).
#include "common.h"
#include <limits.h>
#include <algorithm>

namespace {

constexpr int KP_TW = 32, KP_TH = 16;        // tile: 32 columns x 16 rows = 512 pixels, 2 per thread (rows r and r + 8)
constexpr int KP_MAXK = 4;
typedef unsigned long long u64;

// heat (a non-negative float32: its bit pattern is order-preserving) in the high word, ~pixel index in the low word: the maximum key is the
// largest heat and, among equal heats, the FIRST pixel in row-major order -- np.argmax's tie rule.  0 = nothing (suppressed / outside).
__device__ __forceinline__ u64 kp_key(float heat, int idx) { return ((u64)__float_as_uint(heat) << 32) | (u64)(0xFFFFFFFFu - (unsigned)idx); }
__device__ __forceinline__ int kp_key_idx(u64 k) { return (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)); }

template <int CTRL>
__device__ __forceinline__ u64 kp_dpp(u64 v) {
    const unsigned lo = (unsigned)rp_dpp<CTRL>((int)(unsigned)v), hi = (unsigned)rp_dpp<CTRL>((int)(unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 kp_max(u64 a, u64 b) { return a > b ? a : b; }
__device__ __forceinline__ u64 kp_wave_max(u64 v) {
    v = kp_max(v, kp_dpp<0xB1>(v));          // quad_perm [1,0,3,2]
    v = kp_max(v, kp_dpp<0x4E>(v));          // quad_perm [2,3,0,1]
    v = kp_max(v, kp_dpp<0x141>(v));         // row_half_mirror
    v = kp_max(v, kp_dpp<0x140>(v));         // row_mirror
    u64 r = 0;
#pragma unroll
    for (int l = 0; l < 64; l += 16) {
        const u64 x = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
        r = kp_max(r, x);
    }
    return r;
}

// rputil.interpolate :43-58 at one point per thread (the arithmetic of interpolate_kernel in geometry.hip): query `q` samples the feature
// block of image q_view[q] at the normalised point q_pt[q]; desc [nq][32]
__global__ void kp_desc_kernel(const float* __restrict__ f, size_t istride, int feat_off, int H, int W, const int* __restrict__ q_view,
                               const float* __restrict__ q_pt, float* __restrict__ desc, int nq) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = t >> 5, c = t & 31;
    if (q >= nq) return;
    const size_t hw = (size_t)H * W;
    const float* fc = f + (size_t)q_view[q] * istride + (size_t)(feat_off + c) * hw;
    const float x = q_pt[(size_t)q * 2 + 0] * (float)(W - 1), y = q_pt[(size_t)q * 2 + 1] * (float)(H - 1);
    const float x0 = floorf(x), y0 = floorf(y);
    const int xi = (int)x0, yi = (int)y0;
    const float wx0 = x0 + 1.0f - x, wy0 = y0 + 1.0f - y, wx1 = x - x0, wy1 = y - y0;
    float v = NAN;
    if (xi >= 0 && xi + 1 < W && yi >= 0 && yi + 1 < H) {
        const float v00 = fc[(size_t)yi * W + xi], v10 = fc[(size_t)(yi + 1) * W + xi];
        const float v01 = fc[(size_t)yi * W + xi + 1], v11 = fc[(size_t)(yi + 1) * W + xi + 1];
        v = ((v00 * wx0 * wy0 + v10 * wx0 * wy1) + v01 * wx1 * wy0) + v11 * wx1 * wy1;
    }
    desc[(size_t)q * 32 + c] = v;
}

// The distance of one query to this thread's two pixels (packed: .x = pixel A, .y = pixel B), channel by channel like
// feature_distance_kernel (geometry.hip): acc = 0; acc = acc + (q_c - f_c)^2.
__device__ __forceinline__ rp_v2f kp_dist2(const float* __restrict__ qd, const rp_v2f (&fv)[32]) {
    rp_v2f acc = {0.f, 0.f};
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
        const float4 q4 = *reinterpret_cast<const float4*>(qd + 4 * c4);
        const float qq[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const rp_v2f d = (rp_v2f){qq[k], qq[k]} - fv[4 * c4 + k];
            acc = acc + d * d;
        }
    }
    return acc;
}

// grid (tiles, views): queries [q_off[v], q_off[v + 1]) search the feature map of view v
__global__ __launch_bounds__(256) void kp_tile_best_kernel(const float* __restrict__ f, size_t istride, int feat_off, int H, int W,
                                                            const float* __restrict__ desc, const int* __restrict__ q_off,
                                                            u64* __restrict__ tilebest, int ntx, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int v = blockIdx.y, tile = blockIdx.x;
    const int q0 = q_off[v], nqv = q_off[v + 1] - q0;
    if (nqv <= 0) return;
    float* qs = (float*)smem;                                   // [nqv][32]
    u64* wbest = (u64*)(qs + (size_t)nqv * 32);                 // [nqv][4] per-wave bests
    for (int i = threadIdx.x; i < nqv * 8; i += 256) reinterpret_cast<float4*>(qs)[i] = reinterpret_cast<const float4*>(desc + (size_t)q0 * 32)[i];
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int r = threadIdx.x >> 5, c = threadIdx.x & 31;
    const int x = tx * KP_TW + c, ya = ty * KP_TH + r, yb = ya + 8;
    const bool va = x < W && ya < H, vb = x < W && yb < H;
    const size_t hw = (size_t)H * W;
    const float* fm = f + (size_t)v * istride + (size_t)feat_off * hw;
    const int ia = ya * W + x, ib = yb * W + x;
    rp_v2f fv[32];
#pragma unroll
    for (int ch = 0; ch < 32; ++ch) {
        fv[ch].x = va ? rp_ldg(fm + (size_t)ch * hw + ia) : 0.f;
        fv[ch].y = vb ? rp_ldg(fm + (size_t)ch * hw + ib) : 0.f;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int q = 0; q < nqv; ++q) {
        const rp_v2f d = kp_dist2(qs + (size_t)q * 32, fv);
        const u64 ka = va ? kp_key(expf(-d.x / 2), ia) : 0ull;
        const u64 kb = vb ? kp_key(expf(-d.y / 2), ib) : 0ull;
        const u64 k = kp_wave_max(kp_max(ka, kb));
        if (lane == 0) wbest[q * 4 + wave] = k;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nqv; q += 256)
        tilebest[(size_t)(q0 + q) * ntiles + tile] = kp_max(kp_max(wbest[q * 4], wbest[q * 4 + 1]), kp_max(wbest[q * 4 + 2], wbest[q * 4 + 3]));
}

// One workgroup per query: the TOPK picks of Sampling (:355-371) from the tile bests.  After pick k the window
// [y - win, min(H - 1, y + win)) x [x - win, min(W - 1, x + win)) (end exclusive, :368-370) is suppressed: the suppressed pixels hold the
// map minimum, which cannot win while one pixel is left.  A tile's best stays valid unless it lies inside a window; those tiles are redone
// without the suppressed pixels.  picks [nq][topk][2] = (x, y).
__global__ __launch_bounds__(256) void kp_pick_kernel(const float* __restrict__ f, size_t istride, int feat_off, int H, int W,
                                                       const float* __restrict__ desc, const int* __restrict__ q_map,
                                                       const u64* __restrict__ tilebest, int ntx, int ntiles, int topk, int win,
                                                       int* __restrict__ picks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* tb = (u64*)smem;                                       // [ntiles] current best of every tile
    __shared__ u64 red[4];
    __shared__ __attribute__((aligned(16))) float qd[32];
    __shared__ int wy0[KP_MAXK], wy1[KP_MAXK], wx0[KP_MAXK], wx1[KP_MAXK];
    const int q = blockIdx.x, v = q_map[q];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int t = threadIdx.x; t < ntiles; t += 256) tb[t] = tilebest[(size_t)q * ntiles + t];
    if (threadIdx.x < 32) qd[threadIdx.x] = desc[(size_t)q * 32 + threadIdx.x];
    __syncthreads();
    const size_t hw = (size_t)H * W;
    const float* fm = f + (size_t)v * istride + (size_t)feat_off * hw;
    for (int k = 0; k < topk; ++k) {
        u64 b = 0;
        for (int t = threadIdx.x; t < ntiles; t += 256) b = kp_max(b, tb[t]);
        b = kp_wave_max(b);
        if (lane == 0) red[wave] = b;
        __syncthreads();
        b = kp_max(kp_max(red[0], red[1]), kp_max(red[2], red[3]));
        // A best heat of exactly 0 (every remaining pixel underflowed: |q - f|^2 > ~207, possible with unbounded descriptors, useTanh = 0) equals
        // the map minimum the suppressed pixels were set to (rputil.py:370): the reference's argmax then returns the FIRST pixel of the whole map,
        // suppressed or not.  (Also: a map with no pixel left.)
        const int idx = (b >> 32) ? kp_key_idx(b) : 0;
        const int py = idx / W, px = idx - py * W;
        if (threadIdx.x == 0) {
            picks[((size_t)q * topk + k) * 2 + 0] = px;
            picks[((size_t)q * topk + k) * 2 + 1] = py;
            wy0[k] = max(0, py - win); wy1[k] = min(H - 1, py + win);
            wx0[k] = max(0, px - win); wx1[k] = min(W - 1, px + win);
        }
        __syncthreads();
        if (k + 1 == topk) break;
        // tiles whose current best was just suppressed: redo them over their unsuppressed pixels (wave-uniform loop over the tiles)
        for (int t = 0; t < ntiles; ++t) {
            const u64 cur = tb[t];
            if (!cur) continue;
            const int ci = kp_key_idx(cur), cy = ci / W, cx = ci - cy * W;
            if (!(cy >= wy0[k] && cy < wy1[k] && cx >= wx0[k] && cx < wx1[k])) continue;
            const int ty = t / ntx, tx = t - ty * ntx;
            const int r = threadIdx.x >> 5, c = threadIdx.x & 31;
            const int x = tx * KP_TW + c, ya = ty * KP_TH + r, yb = ya + 8;
            bool va = x < W && ya < H, vb = x < W && yb < H;
            for (int j = 0; j <= k; ++j) {
                if (ya >= wy0[j] && ya < wy1[j] && x >= wx0[j] && x < wx1[j]) va = false;
                if (yb >= wy0[j] && yb < wy1[j] && x >= wx0[j] && x < wx1[j]) vb = false;
            }
            const int ia = ya * W + x, ib = yb * W + x;
            rp_v2f fv[32];
#pragma unroll
            for (int ch = 0; ch < 32; ++ch) {
                fv[ch].x = va ? rp_ldg(fm + (size_t)ch * hw + ia) : 0.f;
                fv[ch].y = vb ? rp_ldg(fm + (size_t)ch * hw + ib) : 0.f;
            }
            const rp_v2f d = kp_dist2(qd, fv);
            const u64 ka = va ? kp_key(expf(-d.x / 2), ia) : 0ull;
            const u64 kb = vb ? kp_key(expf(-d.y / 2), ib) : 0ull;
            const u64 kk = kp_wave_max(kp_max(ka, kb));
            __syncthreads();                                    // (everyone has read tb[t] and red[])
            if (lane == 0) red[wave] = kk;
            __syncthreads();
            if (threadIdx.x == 0) tb[t] = kp_max(kp_max(red[0], red[1]), kp_max(red[2], red[3]));
            __syncthreads();
        }
    }
}

// One wave per view: the slot table in order -> compacted keypoints.  slot_kind: -1 empty, -2 a host coordinate (slot_xy), >= 0 the pick
// with that linear index (query * topk + k); picks on the last row / column are dropped (rputil.py:192-196, :216-218).  Weight 1 inside
// the observed region, 0.99 (MARKER) outside (:226-235 / :341-351: both bounds inclusive).
__global__ __launch_bounds__(64) void kp_assemble_kernel(const int* __restrict__ slot_kind, const double* __restrict__ slot_xy, int L,
                                                          const int* __restrict__ picks, int H, int W, double bx0, double bx1, double by0,
                                                          double by1, int observed_only, double* __restrict__ pts, double* __restrict__ weight,
                                                          int* __restrict__ npts) {
    const int v = blockIdx.x, lane = threadIdx.x;
    int base = 0;
    for (int s0 = 0; s0 < L; s0 += 64) {
        const int s = s0 + lane;
        const int kind = s < L ? slot_kind[(size_t)v * L + s] : -1;
        double x = 0, y = 0;
        bool ok = false;
        if (kind == -2) { x = slot_xy[((size_t)v * L + s) * 2]; y = slot_xy[((size_t)v * L + s) * 2 + 1]; ok = true; }
        else if (kind >= 0) {
            x = (double)picks[(size_t)kind * 2]; y = (double)picks[(size_t)kind * 2 + 1];
            ok = x < (double)(W - 1) && y < (double)(H - 1);
        }
        const bool inside = x >= bx0 && x <= bx1 && y >= by0 && y <= by1;
        if (observed_only) ok = ok && inside;          // doCompletion = 0 (rpmodule.py:534-537): the weight-1 keypoints, in their order
        const unsigned long long m = __ballot(ok);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (ok) {
            pts[((size_t)v * L + pos) * 2] = x; pts[((size_t)v * L + pos) * 2 + 1] = y;
            weight[(size_t)v * L + pos] = inside ? 1.0 : 0.99;
        }
        base += __popcll(m);
    }
    // padding: a valid coordinate and weight 0 (never read beyond npts; keeps padded reads in bounds)
    for (int s = base + lane; s < L; s += 64) { pts[((size_t)v * L + s) * 2] = 0.0; pts[((size_t)v * L + s) * 2 + 1] = 0.0; weight[(size_t)v * L + s] = 0.0; }
    if (lane == 0) npts[v] = base;
}

struct KpWs { size_t desc, tilebest, picks, total; };
KpWs kp_ws(int nq, int H, int W, int topk) {
    const int ntiles = ((W + KP_TW - 1) / KP_TW) * ((H + KP_TH - 1) / KP_TH);
    KpWs o; size_t off = 0;
    o.desc = off; off += rp_align((size_t)nq * 32 * sizeof(float));
    o.tilebest = off; off += rp_align((size_t)nq * ntiles * sizeof(u64));
    o.picks = off; off += rp_align((size_t)nq * topk * 2 * sizeof(int));
    o.total = off;
    return o;
}

}  // namespace

extern "C" {

size_t relpose_keypoints_reference_workspace_bytes(int32_t nq, int32_t H, int32_t W, int32_t topk) {
    if (nq <= 0 || H <= 0 || W <= 0 || topk < 1 || topk > KP_MAXK) return 0;
    return kp_ws(nq, H, W, topk).total;
}

int relpose_keypoints_reference(const float* f, int64_t image_stride, int32_t feat_off, int32_t n_views, int32_t H, int32_t W,
                                const int32_t* q_src_view, const float* q_pt, const int32_t* q_map_view, const int32_t* q_off, int32_t nq,
                                int32_t nq_view_max, int32_t topk, int32_t window, const int32_t* slot_kind, const double* slot_xy, int32_t L,
                                int32_t mask_method, int32_t flags, double* pts, double* weight, int32_t* npts, void* workspace, size_t workspace_bytes,
                                void* stream) {
    if (!f || !q_off || !slot_kind || !slot_xy || !pts || !weight || !npts) return RELPOSE_EINVAL;
    if (nq > 0 && (!q_src_view || !q_pt || !q_map_view || !workspace)) return RELPOSE_EINVAL;
    if (n_views <= 0 || H <= 1 || W <= 1 || nq < 0 || nq_view_max < 0 || nq_view_max > RELPOSE_KP_MAX_QUERIES_PER_VIEW || (nq > 0 && nq_view_max == 0) || topk < 1 ||
        topk > KP_MAXK || window < 0 || L <= 0 || feat_off < 0 || image_stride < (int64_t)(feat_off + 32) * H * W || mask_method < 0 || mask_method > 1 ||
        (flags & ~RELPOSE_KP_OBSERVED_ONLY))
        return RELPOSE_EINVAL;
    const KpWs o = kp_ws(std::max(nq, 1), H, W, topk);
    if (nq > 0 && workspace_bytes < o.total) return RELPOSE_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* desc = (float*)(ws + o.desc);
    u64* tilebest = (u64*)(ws + o.tilebest);
    int* picks = (int*)(ws + o.picks);
    const int ntx = (W + KP_TW - 1) / KP_TW, nty = (H + KP_TH - 1) / KP_TH, ntiles = ntx * nty;
    if (nq > 0) {       // (no query at all -- every view of the batch without SIFT detections --: only the slot tables are assembled)
        hipLaunchKernelGGL(kp_desc_kernel, dim3((nq * 32 + 255) / 256), dim3(256), 0, s, f, (size_t)image_stride, feat_off, H, W, q_src_view, q_pt, desc, nq);
        // 160 bytes of dynamic LDS per query of the largest view group: beyond the 64 KB default cap (409 queries) the launch needs the attribute
        // (ADVICE r5); RELPOSE_KP_MAX_QUERIES_PER_VIEW = 1000 queries = 160 000 B, inside the CU's 160 KiB
        const size_t lds1 = (size_t)nq_view_max * (32 * sizeof(float) + 4 * sizeof(u64));
        if (lds1 > 48 * 1024) RP_HIP(hipFuncSetAttribute((const void*)kp_tile_best_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL(kp_tile_best_kernel, dim3(ntiles, n_views), dim3(256), lds1, s, f, (size_t)image_stride, feat_off, H, W, desc, q_off, tilebest, ntx,
                           ntiles);
        hipLaunchKernelGGL(kp_pick_kernel, dim3(nq), dim3(256), (size_t)ntiles * sizeof(u64), s, f, (size_t)image_stride, feat_off, H, W, desc, q_map_view,
                           tilebest, ntx, ntiles, topk, window, picks);
    }
    // observed region of the weights: getKeypoint :226 (x in [H, 2H]) / getKeypoint_kinect :341 (the 88 x 66 crop, bounds inclusive)
    double bx0, bx1, by0, by1;
    if (mask_method == RELPOSE_MASK_SECOND) { bx0 = H; bx1 = 2 * H; by0 = -1e300; by1 = 1e300; }
    else { bx0 = H + H / 2 - 88 / 2; bx1 = H + H / 2 + 88 / 2; by0 = H / 2 - 66 / 2; by1 = H / 2 + 66 / 2; }
    hipLaunchKernelGGL(kp_assemble_kernel, dim3(n_views), dim3(64), 0, s, slot_kind, slot_xy, L, picks, H, W, bx0, bx1, by0, by1,
                       (flags & RELPOSE_KP_OBSERVED_ONLY) ? 1 : 0, pts, weight, npts);
    RP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
