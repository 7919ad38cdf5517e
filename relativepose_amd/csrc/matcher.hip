// Spectral-matching pose module on gfx950: batched replacement of
// RelativePoseEstimation_helper (reference RPModule/rpmodule.py:317-508).
//
// Kernel sequence per batch of scan pairs (all on one stream, no host sync; 6 dispatches):
//   affinity (affinity.hip)  rpmodule.py:342-379  N x N descriptor affinity (f32 distance in numpy's summation order, f64 weights),
//                                                  row norm, row top-K
//   pair_tile        rpmodule.py:381-451  all C(C-1)/2 correspondence pairs in 64 x 64 tiles: distance screen, compacted angle test,
//                                         survivors as a symmetric bitmap
//   pair_scan                             CSR row pointers + segment pointers of the compatibility graph, status
//   pair_fill_rows   rpmodule.py:453-467  weights of the survivors, written in deterministic order (segment layout)
//   fit_pair         rpmodule.py:212-315  IRLS + spectral rounds; one workgroup per scan pair, Lanczos eigen-solver
//
// The fit never materialises the reference's [4M] stacked arrays: a pair's IRLS weight factorises into (pair weight) x
// (per-correspondence reweighting product), so every weighted sum over 4M elements collapses to a sum over the C correspondences
// with the graph's weighted degree (DESIGN.md "fit").  The leading eigenvector comes from Lanczos on the C x C compressed graph
// instead of ARPACK on the (Ns*Nt)^2 sparse matrix.
//
// Compiled with -ffp-contract=off: the f64 threshold tests must round like numpy.
#include "matcher_internal.h"
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define RP_EPS 1e-12
#define RP_OFFSET 50.0

namespace {

struct Graph {                 // per-batch device arrays of the pair-compatibility graph
    int32_t Cmax, Wmax;
    int64_t max_edges;
    const int32_t* corres_j;   // [B, ns_max, topK]
    const double* corres_w;    // [B, ns_max, topK]
    const int32_t* keff;       // [B]
    unsigned long long* bitmap;  // [B, Cmax, Wmax]
    int32_t* upcnt;            // [B, Cmax]
    int32_t* counters;         // [B, 4]  n_dist, M, nnz(x2), unused
    int32_t* rowptr;           // [B, Cmax+1]
    int32_t* col;              // [B, max_edges]
    double* wv;                // [B, max_edges]
    double* xe;                // [B, max_edges]
    double* state;             // [B, 4, Cmax]  deg, gP, gN, rsum
    double* geo;               // [B, Cmax, 12]  sp, tp, sn, tn of every correspondence (gathered once per fit)
    int32_t* pairC;            // [B] number of correspondences of a pair whose fit is running, 0 otherwise
    // Edge storage: every row is cut into SEGMENTS of <= 32 edges; segment s lives in slot s % 64 of wave-slice s / 64, edge k % 32 of it at
    // ((s >> 6) << 11) + ((k & 31) << 6) + (s & 63): the 64 lanes of a wave, each walking its own segment, read 64 consecutive entries per step
    // (SELL-64 over segments).  (r6) Segment numbering: first every row's FULL segments (32 edges), row by row -- row c owns
    // segptr[c] .. segptr[c+1]-1, segptr[C] = their number --, then the rows' last, PARTIAL segments sorted by length class
    // (25-31, 17-24, 9-16, 1-8 edges; row order within a class): segpart[c] or -1, segpart[C] = all segments.  A wave of the edge passes thus
    // sees 64 segments of (almost always) one class and skips the 8-edge batches none of them has -- with the rows' segments numbered row by
    // row (rounds 2-5) nearly every wave ran all four batches for its shortest segments (30 % of the slots dead at N = 200).  A row's sum
    // still adds its segments' partial sums in the row's own order (full ones, then the partial one): bitwise the same results.
    // The bitmap holds the full symmetric adjacency.
    int32_t seg_cap;           // segments per pair the edge arrays are sized for (multiple of 64)
    int64_t estride;           // entries of col / wv / xe per pair
    int32_t* segptr;           // [B, Cmax+1] first FULL segment of every row; [C] = number of full segments
    int32_t* segpart;          // [B, Cmax+1] the row's partial segment (-1: none); [C] = number of segments
    int32_t* segrow;           // [B, seg_cap] row | length << 16 of every segment (the fit's segment-table word)
    double* part;              // [B, seg_cap] per-segment partial sums of the edge passes
};
#define RP_SEG 32
__device__ __forceinline__ size_t seg_edge_index(int seg, int k) { return ((size_t)(seg >> 6) << 11) + ((size_t)(k & 31) << 6) + (seg & 63); }

__device__ __forceinline__ int pair_C(const RelposeKeypoints& kp, const Graph& g, int b) {
    int ns = kp.ns[b], nt = kp.nt[b];
    if (ns < 3 || nt < 3) return 0;
    return ns * g.keff[b];
}

struct Corr { double ps[3], ns[3], pt[3], nt[3]; double f, ws, wt; };

__device__ __forceinline__ void load_corr(const RelposeKeypoints& kp, const Graph& g, int b, int topK, int keff, int c, Corr& o) {
    int i = c / keff, kk = c - i * keff;
    size_t si = (size_t)b * kp.ns_max + i;
    int j = g.corres_j[si * topK + kk];
    size_t ti = (size_t)b * kp.nt_max + j;
    o.f = g.corres_w[si * topK + kk];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o.ps[a] = kp.pc_s[si * 3 + a]; o.ns[a] = kp.normal_s[si * 3 + a];
        o.pt[a] = kp.pc_t[ti * 3 + a]; o.nt[a] = kp.normal_t[ti * 3 + a];
    }
    o.ws = kp.weight_s[si]; o.wt = kp.weight_t[ti];
}

// ------------------------------------------------------------------ pair consistency
__global__ __launch_bounds__(1024) void pair_scan_kernel(RelposeKeypoints kp, Graph g, int32_t* __restrict__ status) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int b = blockIdx.x;
    const int C = pair_C(kp, g, b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t* rp = g.rowptr + (size_t)b * (g.Cmax + 1);
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < C; base += 1024) {
        const int c = base + threadIdx.x;
        int v = 0;
        if (c < C) {
            // degree = set bits of the row; kept in upcnt for the segment pass below
            const unsigned long long* row = g.bitmap + ((size_t)b * g.Cmax + c) * g.Wmax;
            const int nw = (C + 63) >> 6;
            for (int w = 0; w < nw; ++w) v += __popcll(row[w]);
            g.upcnt[(size_t)b * g.Cmax + c] = v;
        }
        int inc = v;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { int t = __shfl_up(inc, m, 64); if (lane >= m) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int pre = carry_s;
        for (int w = 0; w < wave; ++w) pre += wsum[w];
        if (c < C) rp[c] = pre + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = pre + inc;
        __syncthreads();
    }
    const int total_edges = carry_s;
    __syncthreads();
    {                            // segment numbering (see Graph): full segments row by row, then the partial ones by length class; the row of every segment
        int32_t* sp = g.segptr + (size_t)b * (g.Cmax + 1);
        int32_t* pp = g.segpart + (size_t)b * (g.Cmax + 1);
        int32_t* sr = g.segrow + (size_t)b * g.seg_cap;
        __shared__ unsigned long long wsum2[16];
        __shared__ unsigned long long carry2_s;
        if (threadIdx.x == 0) { carry_s = 0; carry2_s = 0ull; }
        __syncthreads();
        for (int base = 0; base < C; base += 1024) {
            const int c = base + threadIdx.x;
            const int deg = (c < C) ? g.upcnt[(size_t)b * g.Cmax + c] : 0;
            const int v = deg / RP_SEG, rem = deg % RP_SEG;
            const int cls = rem ? (rem + 7) >> 3 : 0;                               // 1 .. 4 (8-edge batches of the partial segment), 0: none
            const unsigned long long key = cls ? 1ull << (16 * (cls - 1)) : 0ull;  // four 16-bit counters (<= Cmax <= 8192 rows each) in one scan
            int inc = v;
            unsigned long long inc2 = key;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const int t = __shfl_up(inc, m, 64);
                const unsigned long long t2 = __shfl_up(inc2, m, 64);
                if (lane >= m) { inc += t; inc2 += t2; }
            }
            if (lane == 63) { wsum[wave] = inc; wsum2[wave] = inc2; }
            __syncthreads();
            int pre = carry_s;
            unsigned long long pre2 = carry2_s;
            for (int w = 0; w < wave; ++w) { pre += wsum[w]; pre2 += wsum2[w]; }
            if (c < C) {
                const int first = pre + inc - v;
                sp[c] = first;
                if (first + v <= g.seg_cap) for (int i = 0; i < v; ++i) sr[first + i] = c | (RP_SEG << 16);
                // (rank of the row's partial segment within its class, class in the top bits: turned into a segment number below)
                pp[c] = cls ? (int)(((pre2 + inc2 - key) >> (16 * (cls - 1))) & 0xffffull) | (cls << 16) : -1;
            }
            __syncthreads();
            if (threadIdx.x == 1023) { carry_s = pre + inc; carry2_s = pre2 + inc2; }
            __syncthreads();
        }
        const int nfull = carry_s;
        const unsigned long long tot = carry2_s;
        const int n4 = (int)((tot >> 48) & 0xffff), n3 = (int)((tot >> 32) & 0xffff), n2 = (int)((tot >> 16) & 0xffff), n1 = (int)(tot & 0xffff);
        for (int c = threadIdx.x; c < C; c += 1024) {
            const int e = pp[c];
            if (e < 0) continue;
            const int cls = e >> 16, rank = e & 0xffff;
            const int rem = g.upcnt[(size_t)b * g.Cmax + c] % RP_SEG;
            const int s_ = nfull + (cls == 4 ? 0 : cls == 3 ? n4 : cls == 2 ? n4 + n3 : n4 + n3 + n2) + rank;     // longest class first
            pp[c] = s_;
            if (s_ < g.seg_cap) sr[s_] = c | (rem << 16);
        }
        if (threadIdx.x == 0) { sp[C] = nfull; pp[C] = nfull + n4 + n3 + n2 + n1; }
    }
    if (threadIdx.x == 0) {
        const int total = total_edges;
        rp[C] = total;
        int st = RELPOSE_OK;
        if (C < 3) st = RELPOSE_FEW_KEYPOINTS;
        else if (g.counters[b * 4 + 0] < 3) st = RELPOSE_DIST_FILTER;
        else if (g.counters[b * 4 + 1] < 3) st = RELPOSE_ANGLE_FILTER;
        else if ((int64_t)total > g.max_edges) st = RELPOSE_EDGE_OVERFLOW;
        status[b] = st;
    }
}

// ---- tiled pair consistency (rpmodule.py:381-451) ------------------------------------------------------------------
// One workgroup = one 64 x 64 tile (rows tr*64.., columns tc*64.., tr <= tc) of the C x C pair matrix of one scan pair.
//   1. the 64 + 64 correspondences of the tile are gathered to LDS once (the row-per-wave kernel gathered per pair);
//   2. every thread SCREENS 16 pairs with the distance test only (two sqrt) and the survivors (a few %) are
//      compacted into an LDS queue -- the row-per-wave kernel ran the six acos of the angle test on every wave that held one
//      survivor, i.e. at 3-5 % lane efficiency;
//   3. the queue is evaluated densely (rp_pair_eval, identical arithmetic), passing pairs set a bit in the tile's row
//      words AND in its transposed column words (LDS atomics);
//   4. the tile stores word tc of its rows and word tr of its columns: the bitmap is the full symmetric adjacency
//      matrix, every (row, word) slot written exactly once (no memset, no global atomics per edge), and the fill
//      kernel reads a row's neighbours from ONE contiguous run of words instead of a strided column walk.
__global__ __launch_bounds__(256) void pair_tile_kernel(RelposeKeypoints kp, Graph g, RpPairConsts kc, int topK) {
    __shared__ double geo[2][64][12];                  // [rows | columns][slot][ps, ns, pt, nt]
    __shared__ unsigned long long bm[2][64];           // row words, transposed (column) words
    __shared__ unsigned short queue[4096];
    __shared__ int qn;
    const int b = blockIdx.y;
    const int C = pair_C(kp, g, b);
    const int T = g.Wmax;
    int tr = 0, t = blockIdx.x;
    while (t >= T - tr) { t -= T - tr; ++tr; }
    const int tc = tr + t;
    if (tc * 64 >= C) return;                           // (tr <= tc: the row tile is inside as well)
    const int keff = g.keff[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // gather: thread = (rows | columns, slot, source | target side)
        const int which = tid >> 7, slot = (tid >> 1) & 63, side = tid & 1;
        const int c = (which ? tc : tr) * 64 + slot;
        double* o = &geo[which][slot][side * 6];
        if (c < C) {
            const int i = c / keff, kk = c - i * keff;
            const size_t si = (size_t)b * kp.ns_max + i;
            if (side == 0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) { o[a] = kp.pc_s[si * 3 + a]; o[3 + a] = kp.normal_s[si * 3 + a]; }
            } else {
                const size_t ti = (size_t)b * kp.nt_max + g.corres_j[si * topK + kk];
#pragma unroll
                for (int a = 0; a < 3; ++a) { o[a] = kp.pc_t[ti * 3 + a]; o[3 + a] = kp.normal_t[ti * 3 + a]; }
            }
        } else {
#pragma unroll
            for (int a = 0; a < 6; ++a) o[a] = 0.0;
        }
        if (tid < 128) bm[tid >> 6][tid & 63] = 0ull;
        if (tid == 0) qn = 0;
    }
    __syncthreads();
    {   // screen: this thread's column against 16 rows (the row operands are LDS broadcasts)
        const int c = tc * 64 + lane;
        const double cps[3] = {geo[1][lane][0], geo[1][lane][1], geo[1][lane][2]};
        const double cpt[3] = {geo[1][lane][6], geo[1][lane][7], geo[1][lane][8]};
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int rl = wave * 16 + i;
            const int r = tr * 64 + rl;
            bool pd = false;
            if (c > r && c < C) {
                double es[3], et[3], ds, dt, d;
                pd = rp_pair_dist(&geo[0][rl][0], &geo[0][rl][6], cps, cpt, kc, es, et, ds, dt, d);
            }
            const unsigned long long m = __ballot(pd);
            if (m) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&qn, __popcll(m));
                base = __shfl(base, 0, 64);
                if (pd) queue[base + __popcll(m & lt)] = (unsigned short)((rl << 6) | lane);
            }
        }
    }
    __syncthreads();
    const int n = qn;
    for (int q = tid; q < n; q += 256) {
        const int e = queue[q], rl = e >> 6, cl = e & 63;
        const double* a = geo[0][rl];
        const double* o = geo[1][cl];
        const RpPairEval ev = rp_pair_eval(a, a + 3, a + 6, a + 9, o, o + 3, o + 6, o + 9, kc);
        if (ev.pass_all) {
            atomicOr(&bm[0][rl], 1ull << cl);
            atomicOr(&bm[1][cl], 1ull << rl);
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int r = tr * 64 + tid;
        unsigned long long w = bm[0][tid];
        const int up = __popcll(w);
        if (tr == tc) w |= bm[1][tid];
        if (r < C) g.bitmap[((size_t)b * g.Cmax + r) * g.Wmax + tc] = w;
        const int m = rp_wave_sum_i(up);
        if (tid == 0) {
            if (n) atomicAdd(&g.counters[b * 4 + 0], n);
            if (m) atomicAdd(&g.counters[b * 4 + 1], m);
        }
    } else if (tid < 128 && tr != tc) {
        const int c = tc * 64 + (tid - 64);
        if (c < C) g.bitmap[((size_t)b * g.Cmax + c) * g.Wmax + tr] = bm[1][tid - 64];
    }
}

// Edge list of one row per wave from the symmetric bitmap: the row's set bits are expanded into an LDS list (ascending
// column = the reference's order: lower neighbours, then upper), then the edges are evaluated DENSELY, 64 per step.
__global__ __launch_bounds__(256) void pair_fill_rows_kernel(RelposeKeypoints kp, Graph g, RpPairConsts kc, int topK,
                                                              const int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned short fill_lists[];      // [4 waves][Cmax]
    const int b = blockIdx.y;
    if (status[b] != RELPOSE_OK) return;
    const int C = pair_C(kp, g, b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + wave;
    if (c >= C) return;
    unsigned short* list = fill_lists + (size_t)wave * g.Cmax;
    const int keff = g.keff[b];
    const unsigned long long* row = g.bitmap + ((size_t)b * g.Cmax + c) * g.Wmax;
    const int nw = (C + 63) >> 6;
    int deg = 0;
    for (int w0 = 0; w0 < nw; w0 += 64) {
        unsigned long long word = (w0 + lane < nw) ? row[w0 + lane] : 0ull;
        const int pc = __popcll(word);
        int inc = pc;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { const int t = __shfl_up(inc, m, 64); if (lane >= m) inc += t; }
        int off = deg + inc - pc;
        while (word) {
            list[off++] = (unsigned short)((w0 + lane) * 64 + __builtin_ctzll(word));
            word &= word - 1;
        }
        deg += __shfl(inc, 63, 64);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (deg == 0) return;
    Corr me;
    load_corr(kp, g, b, topK, keff, c, me);
    const size_t eoff = (size_t)b * g.estride;
    const int row_start = g.rowptr[(size_t)b * (g.Cmax + 1) + c];
    const int seg0 = g.segptr[(size_t)b * (g.Cmax + 1) + c];                  // the row's full segments: seg0, seg0 + 1, ...
    const int segp = g.segpart[(size_t)b * (g.Cmax + 1) + c];                 // ... and its partial one (edges nfe .. deg - 1)
    const int nfe = deg / RP_SEG * RP_SEG;
    int nz = 0;
    for (int k = lane; k < deg; k += 64) {
        const int x = list[k];
        Corr o;
        load_corr(kp, g, b, topK, keff, x, o);
        double w;
        if (x < c) {             // canonical orientation: "1" = the smaller index
            const RpPairEval ev = rp_pair_eval(o.ps, o.ns, o.pt, o.nt, me.ps, me.ns, me.pt, me.nt, kc);
            w = rp_pair_weight(ev, o.f, me.f, o.ws, me.ws, o.wt, me.wt, kc);
        } else {
            const RpPairEval ev = rp_pair_eval(me.ps, me.ns, me.pt, me.nt, o.ps, o.ns, o.pt, o.nt, kc);
            w = rp_pair_weight(ev, me.f, o.f, me.ws, o.ws, me.wt, o.wt, kc);
        }
        const size_t pos = seg_edge_index(k < nfe ? seg0 + (k >> 5) : segp, k);
        g.col[eoff + pos] = x;
        g.wv[eoff + pos] = w;
        nz += (w != 0.0) ? 1 : 0;
    }
    // (r6) the unused slots of the row's last segment: column 0, weight 0 -- an edge pass then needs no per-slot predicate: a dead slot adds
    // w * (...) = +-0 to a sum that is never -0 (seg_body)
    for (int k = deg + lane; k < ((deg + RP_SEG - 1) / RP_SEG) * RP_SEG; k += 64) {
        const size_t pos = seg_edge_index(segp, k);
        g.col[eoff + pos] = 0;
        g.wv[eoff + pos] = 0.0;
    }
    nz = rp_wave_sum_i(nz);
    if (lane == 0 && nz) atomicAdd(&g.counters[b * 4 + 2], nz);
}

// ------------------------------------------------------------------ fit
struct FitCtx {
    const double* geo;      // [C][12] of this pair
    int b, C;
    double mu;
    double* deg; double* gP; double* gN; double* rsum;
    double* red;        // LDS reduction scratch
};

__device__ __forceinline__ void corr_geom(const FitCtx& f, int c, double* sp, double* tp, double* sn, double* tn) {
    const double* g = f.geo + (size_t)c * 12;
#pragma unroll
    for (int a = 0; a < 3; ++a) { sp[a] = g[a]; tp[a] = g[3 + a]; sn[a] = g[6 + a]; tn[a] = g[9 + a]; }
}

// ------------------------------------------------------------------ single-workgroup fit (the default path)
// One workgroup per scan pair (512 threads up to 1024 correspondences, 768 beyond) runs the WHOLE fit of rpmodule.py:212-315 in one launch: status
// finalisation, geometry gather, degrees, the IRLS iterations and the five spectral rounds.  The pair's
// compatibility graph (~31 k directed edges at N = 200) is streamed from L2 once per matrix-vector product in the
// segment layout (see Graph): a lane walks one segment of <= 32 edges of one row, a wave reads 64 consecutive entries
// per step, no cross-lane reduction; every vector the products gather from lives in LDS.  The leading eigenvector
// comes from a Lanczos iteration with full re-orthogonalisation (classical Gram-Schmidt against the whole basis, a
// second pass whenever the first one cancelled more than half of the vector; basis in global scratch), checked for
// convergence every RP_LZ_CHECK steps and restarted from the Ritz vector until the residual estimate
// beta_m |s_m| <= RP_LZ_TOL * theta -- i.e. a CONVERGED eigenvector like the reference's ARPACK call
// (rpmodule.py:273), where round 1 ran a capped power iteration.  Rounds 2..5 start from the previous round's
// eigenvector.  A pair that is still not converged after RP_LZ_MAXPROD products gets RELPOSE_NOT_CONVERGED.
// All reductions have a fixed order: results are bitwise reproducible and independent of the batch.
#define RP_FIT1_THREADS 1024      // (the rounds 2-5 size of the large kernel: experiments build only since round 6)
#define RP_LZ_M 24              // Lanczos steps per cycle (basis size)
#define RP_LZ_CHECK 8           // convergence test every 8 steps
#define RP_LZ_MAXPROD 192       // products per eigen-solve before giving up
#define RP_LZ_TOL 1e-13
#ifndef RP_TRI_ROUNDS
#define RP_TRI_ROUNDS 4         // 64-way multisection rounds of the tridiagonal eigenvalue (6 bits each) before the Newton polish (>= 9: no polish)
#endif
#define RP_FIT1_MAXC 4500       // LDS: 3 vectors of C doubles + 2 x (C + 1) ints
#define RP_FIT1_MAXC_HU 1024    // ... + the {h, u} pairs of the products (16 bytes per correspondence more): the 512-thread kernel's sizes.  (Measured at
                                // Cmax = 2000 / 1024 threads: the pairs take the LDS of the basis vectors and buy nothing there: 5.63 -> 5.69 ms.)

struct Fit1 {                   // LDS layout + per-pair pointers of the single-workgroup fit
    long long* prof;            // optional [16] cycle counters of block 0 (RELPOSE_FIT_PROF=1, experiments build)
    double* vec;                // [C] current Lanczos vector / eigenvector (gather source of the products)
    double* hh;                 // [C] h = relu(50 - r)
    double* yy;                 // [C] product output
    int32_t* sp;                // [C + 1] first full segment of every row (Graph::segptr)
    int32_t* pp;                // [C + 1] the row's partial segment or -1 (Graph::segpart)
    int nfull;                  // segments below this number are full (32 edges)
    int32_t* meta;              // [meta_cap] LDS: row | length << 16 of the pair's first meta_cap segments (constant over the fit; 0 entries: helpers, global layout)
    int meta_cap;
    double* red;                // [160] reduction scratch
    double* cbuf;               // [RP_LZ_M + 1] Gram-Schmidt coefficients
    double* tri;                // [4 * (RP_LZ_M + 1)] alpha, beta, s of the tridiagonal solve (the fourth block is spare)
    const int32_t* col; const double* wv; double* xe;      // this pair's edges (global, segment layout)
    __amdgpu_buffer_rsrc_t rs_col, rs_wv, rs_xe;           // ... as buffer descriptors (seg_body's loads)
    const int32_t* segrow; double* part;
    double* part2;              // partial sums of the products done with helper workgroups (only ever written write-through)
    double* V;                  // [(RP_LZ_M + 1), Cmax] Lanczos basis (global scratch)
    double* hu;                 // LDS [C][2]: {h_c, u_c} interleaved, the gather source of the leader's products (seg_body HU); null in the global layout
    double* partl;              // LDS: the partial sums of the pair's first meta_cap segments (one workgroup per pair; round 6) -- `part` beyond
    double* Vl;                 // LDS copy of basis vectors 0 .. KL-1 (round 6: what is left of the CU's 160 KB behind the vectors and the segment
    int KL;                     // table; a multiple of 4).  The re-orthogonalisation reads those from LDS, the rest from global: same sums, same order
    int C, Cmax, nseg, tri_rounds, max_prod;
    int fixed_checks;           // RELPOSE_TUNE_FIT_FIXED_CHECKS: convergence test every RP_LZ_CHECK steps (the round-2/3 rule; A/B switch)
    struct FitCtl* ctl;         // helper workgroups (G > 1): the pair's control block; xu = the published vectors [2][Cmax] (u, then h), part2 behind them
    double* xu;
    int G;
    unsigned* epoch;            // LDS (leader): [0] products published so far, [1] h versions published, [2] claimed chunk
};

// ---- helper workgroups for the matrix-vector products ----------------------------------------------------------------------------
// The products are 66 % (N = 200) to 79 % (N = 400) of the fit and one CU's vector-memory + LDS-gather + f64 pipes are what bounds
// them, so G - 1 HELPER workgroups per scan pair take chunks of segments of every product.  The leader (blockIdx.x == 0) runs the
// whole fit as before; per product it publishes the vector (write-through stores), then ONE control word {product number, h version,
// next chunk}; everyone -- leader included -- claims chunks (nseg / G segments each) with a compare-and-swap on that word, writes
// the per-segment partial sums write-through and counts the chunk done; the leader waits for all chunks and adds the partials up
// per row in segment order.  Properties:
//   * identical results for any G and any timing: a partial sum belongs to a segment, not to whoever computed it;
//   * no co-residency assumption: the leader never waits for a workgroup that has not claimed work (a helper that starts late, or
//     never, just takes nothing; a helper that sees no leader within ~30 ms gives up), and it takes every chunk itself if alone;
//   * a helper loads the vector while its claim is in flight: a successful claim means the product is still running, so the leader
//     cannot have started to overwrite the vector (it waits for every claimed chunk first);
//   * cross-CU visibility by the write-through / L1-bypassing forms only (8-byte relaxed agent-scope atomics for payload and
//     control words, vmcnt(0) before every publish: cdna_hip_programming.md Guideline 16, R1), in memory that is NEVER touched by
//     ordinary stores: the partial sums of these products have their own array (part2) -- the leader's plain-stored partials of the
//     other passes sit dirty in its XCD's L2 and would shadow (and later overwrite) what a helper on another XCD wrote through;
//     all polled words are zeroed by a memset node in front of the launch.
struct FitCtl {
    unsigned long long claim;   // {product << 40 | h version << 32 | next chunk}; product 0 = leader not there yet, 0xffffff = fit finished
    unsigned long long done;    // {product << 40 | chunks finished}
    unsigned long long pad[6];
};
#define RP_FIT_DONE 0xffffffu
typedef unsigned long long rp_u64;
__device__ __forceinline__ void rp_st_sc1(double* p, double v) { __hip_atomic_store((rp_u64*)p, (rp_u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double rp_ld_sc1(const double* p) { return __longlong_as_double((long long)__hip_atomic_load((const rp_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void rp_st_sc1(rp_u64* p, rp_u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ rp_u64 rp_ld_sc1(const rp_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rp_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Formal side of the protocol (HSA memory model): every flag store / read-modify-write that publishes data is preceded by an agent-scope
// RELEASE fence, every flag observation that licenses reading such data is followed by an agent-scope ACQUIRE fence (fences rather than
// ordered atomics: the polls stay relaxed, one invalidate per observation instead of one per poll).  The write-through payload and
// vmcnt(0) drains above are what makes it work on gfx950; the fences are what makes it a data-race-free program.  RP_FIT_FENCES=0 builds
// without them (A/B, matcher alone at B=32, same box: leader + 3 helpers 2.87 -> 3.00-3.07 ms at N=200, 4.81 -> 4.96 at N=400; leader + 7: 3.14 ->
// 3.37-3.44 and 4.50 -> 4.83; one workgroup per pair -- the throughput configurations -- unchanged: no flag is touched).
#ifndef RP_FIT_FENCES
#define RP_FIT_FENCES 1
#endif
__device__ __forceinline__ void rp_release_agent() { if (RP_FIT_FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__device__ __forceinline__ void rp_acquire_agent() { if (RP_FIT_FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// (chunk geometry of the distributed products: rp_fit_chunk_size / _count / _begin in rp_math.h, CPU-tested)

// One pass over the pair's edges: thread <-> segment (<= 32 edges of one row, read with stride 64 so that a wave's loads
// are 64 consecutive entries), sequential accumulation per segment, then every row adds up its segments in order.
//   MODE 0: val = w                                   (weighted degrees)
//   MODE 1: val = (base * (h[r] + h[cc])) * u[cc]      (rpmodule.py:262-267; base = w, or mu * xe for 'spectral' rounds > 0)
//   MODE 2: val = x = relu(u[r] * u[cc]) * w           (rpmodule.py:277-280); x is stored to xe when store_x
// RP_SEG_CFG = (register batches of 8 edges in flight per lane: 2, or 4 = the whole segment) | (edges whose LDS gathers are in flight together: 8 in
// the 512-thread kernel, 4 in the 1024-thread one with its 128 VGPRs) << 4
// SCALED (MODE 1 only): base = mu * xe, the 'spectral' method's rounds > 0 -- a compile-time choice: as a run-time one it cost every edge of
// every method a multiplication and two selects
// HU (MODE 1, one workgroup per pair, LDS layout; round 6): h and the Lanczos vector are gathered as ONE 16-byte entry {h_c, u_c} of Fit1::hu
// (kept beside the two dense vectors by lanczos_top) -- one LDS instruction and one address per edge instead of two.
template <int MODE, int RP_SEG_CFG, bool SC1, bool SCALED = false, bool HU = false>
__device__ __forceinline__ void seg_body(const Fit1& f, int sgm, double mu_xe, bool store_x) {
    static_assert(!SCALED || MODE == 1, "SCALED is a MODE 1 variant");
    static_assert(!HU || (MODE == 1 && !SC1), "HU is a MODE 1 variant of the leader's own passes");
    const double* base = SCALED ? f.xe : f.wv;
    constexpr int U = 8;                                  // edges per register batch; two batches in flight
    constexpr bool PRED = SCALED;                         // per-slot predicate on the sums: only where the loaded weights are not zero-filled (xe)
    constexpr int RP_SEG_DEPTH = RP_SEG_CFG & 15, GQ = RP_SEG_CFG >> 4;
    static_assert((RP_SEG_DEPTH == 2 || RP_SEG_DEPTH == 4) && (GQ == 2 || GQ == 4 || GQ == 8), "seg_body configuration");
    {
        // the segment's row and length: one LDS word (filled once per fit) instead of a global load of the row followed by three
        // dependent LDS reads in front of every segment's first edge load
        int r, len;
        if (sgm < f.meta_cap) { const int mw = f.meta[sgm]; r = mw & 0xffff; len = mw >> 16; }
        else {
            const int mw = f.segrow[sgm]; r = mw & 0xffff; len = mw >> 16;
        }
        const size_t e0 = seg_edge_index(sgm, 0);
        const double hr = (MODE == 1) ? f.hh[r] : 0.0, ur = (MODE == 2) ? f.vec[r] : 0.0;
        double acc = 0.0;
        int cc[RP_SEG_DEPTH][U]; double w[RP_SEG_DEPTH][U];
        // dead slots (beyond a shorter segment's length) of a loaded batch are loaded and gathered like everybody's (no branch around the loads;
        // rounds 2-5 clamped their address to the segment's first entry and predicated the sums, round 6 zero-fills them when the rows are
        // written).  Gating whole batches with a ballot, or processing all 32 slots branch-free, measured 20-30 % slower.
        // (r6) buffer loads: the lane's byte offset of slot 0 in a VGPR, the slot's distance (a compile-time constant after unrolling) in the
        // instruction's scalar / immediate offset -- no 64-bit address arithmetic per edge (it was half of an edge's vector instructions).  A dead
        // slot's load returns whatever the slot holds (or 0 beyond the pair's region: the descriptor's range check); `consume` never uses it.
        const int vc = (int)e0 * 4, vw = (int)e0 * 8;
        const __amdgpu_buffer_rsrc_t rs_w = SCALED ? f.rs_xe : f.rs_wv;
        auto load = [&](int buf, int kb) {
#pragma unroll
            for (int q = 0; q < U; ++q) {
                // (the slot's offset inside the batch folds into the instruction's 12-bit immediate, the batch's offset is one scalar per batch)
                cc[buf][q] = (int)__builtin_amdgcn_raw_buffer_load_b32(f.rs_col, vc + q * 64 * 4, kb * 64 * 4, 0);
                typedef unsigned rp_u32x2 __attribute__((__vector_size__(8)));
                const rp_u32x2 wv2 = (rp_u32x2)__builtin_amdgcn_raw_buffer_load_b64(rs_w, vw + q * 64 * 8, kb * 64 * 8, 0);
                w[buf][q] = __builtin_bit_cast(double, wv2);
            }
        };
        // The gathers of GQ edges are issued together, for dead slots too (column 0: a valid index); with `if (live) { gather; multiply; add }`
        // per edge every edge paid its own LDS round trip in a serial chain.  Same operations in the same order on the live slots, +-0 added for
        // a dead one (PRED, the 'spectral' rounds: a select on the sum instead): bitwise the same sums.
        auto consume = [&](int buf, int kb) {
#pragma unroll
            for (int q0 = 0; q0 < U; q0 += GQ) {
                double hv[GQ], uv[GQ];
#pragma unroll
                for (int q = 0; q < GQ; ++q) {
                    // (a dead slot of a loaded batch holds column 0, weight 0 -- pair_fill_rows_kernel; the 'spectral' rounds' xe is not zero-filled: PRED)
                    const int c = (!PRED || kb + q0 + q < len) ? cc[buf][q0 + q] : 0;
                    if (HU) { const double2 p = *reinterpret_cast<const double2*>(f.hu + 2 * c); hv[q] = p.x; uv[q] = p.y; }
                    else {
                        hv[q] = (MODE == 1) ? f.hh[c] : 0.0;
                        uv[q] = (MODE != 0) ? f.vec[c] : 0.0;
                    }
                }
#pragma unroll
                for (int q = 0; q < GQ; ++q) {
                    const bool live = kb + q0 + q < len;
                    const double wq = w[buf][q0 + q];
                    double val;
                    if (MODE == 0) val = wq;
                    else if (MODE == 1) {
                        const double bb = SCALED ? mu_xe * wq : wq;
                        val = (bb * (hr + hv[q])) * uv[q];
                    } else {
                        double x = ur * uv[q];
                        x = (x < 0.0 ? 0.0 : x) * wq;
                        if (store_x && live) *(RP_GLOBAL double*)(f.xe + e0 + (size_t)(kb + q0 + q) * 64) = x;
                        val = x;
                    }
                    const double sum = acc + val;
                    acc = (!PRED || live) ? sum : acc;                   // (!PRED: val is +-0 on a dead slot and acc is never -0: the same bits)
                }
            }
        };
        static_assert(RP_SEG == 4 * U, "seg_pass is unrolled for 4 batches");
        if (RP_SEG_DEPTH == 4) {          // the whole segment in flight (64 loads per lane) before the first gather
            load(0, 0);
            if (len > U) load(1, U);
            if (len > 2 * U) load(2, 2 * U);
            if (len > 3 * U) load(3, 3 * U);
            consume(0, 0);
            if (len > U) consume(1, U);
            if (len > 2 * U) consume(2, 2 * U);
            if (len > 3 * U) consume(3, 3 * U);
        } else {
            load(0, 0);
            if (len > U) load(1, U);
            consume(0, 0);
            if (len > 2 * U) load(0, 2 * U);
            if (len > U) consume(1, U);
            if (len > 3 * U) load(1 % RP_SEG_DEPTH, 3 * U);
            if (len > 2 * U) consume(0, 2 * U);
            if (len > 3 * U) consume(1 % RP_SEG_DEPTH, 3 * U);
        }
        if (SC1) rp_st_sc1(f.part2 + sgm, acc);            // (helpers / a leader with helpers: write-through, read by the leader's row sums)
        else if (sgm < f.meta_cap) f.partl[sgm] = acc;     // (LDS: no store -> L2 -> load round trip between the edge pass and the row sums)
        else *(RP_GLOBAL double*)(f.part + sgm) = acc;
    }
}
// every row adds up its segments' partial sums in segment order (SC1: from part2 with L1-bypassing loads -- their latency is a
// trip to the memory side, so two rows x 8 partial sums are in flight per thread)
template <bool SC1>
__device__ __forceinline__ void seg_row_sums(const Fit1& f, double* out) {
    if (SC1) {
        for (int r0 = threadIdx.x; r0 < f.C; r0 += 2 * blockDim.x) {
            const int r1 = r0 + blockDim.x;
            const bool two = r1 < f.C;
            const int a0 = f.sp[r0], a1 = f.sp[r0 + 1], b0 = two ? f.sp[r1] : 0, b1 = two ? f.sp[r1 + 1] : 0;
            double acc0 = 0.0, acc1 = 0.0;
            for (int k = 0; a0 + k < a1 || b0 + k < b1; k += 8) {
                double v[8], w[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    v[q] = a0 + k + q < a1 ? rp_ld_sc1(f.part2 + a0 + k + q) : 0.0;
                    w[q] = b0 + k + q < b1 ? rp_ld_sc1(f.part2 + b0 + k + q) : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (a0 + k + q < a1) acc0 += v[q];
                    if (b0 + k + q < b1) acc1 += w[q];
                }
            }
            const int p0 = f.pp[r0], p1 = two ? f.pp[r1] : -1;                     // the rows' partial segments come last, as they always did
            const double v0 = p0 >= 0 ? rp_ld_sc1(f.part2 + p0) : 0.0, v1 = p1 >= 0 ? rp_ld_sc1(f.part2 + p1) : 0.0;
            if (p0 >= 0) acc0 += v0;
            if (p1 >= 0) acc1 += v1;
            out[r0] = acc0;
            if (two) out[r1] = acc1;
        }
    } else {
        for (int r = threadIdx.x; r < f.C; r += blockDim.x) {
            double acc = 0.0;
            const int a = f.sp[r], b = f.sp[r + 1], bl = min(b, f.meta_cap);
            for (int sgm = a; sgm < bl; ++sgm) acc += f.partl[sgm];                                    // (same order: LDS-resident segments come first)
            for (int sgm = max(a, f.meta_cap); sgm < b; ++sgm) acc += *(RP_GLOBAL const double*)(f.part + sgm);
            const int p = f.pp[r];                                                 // the row's partial segment comes last, as it always did
            if (p >= 0) acc += p < f.meta_cap ? f.partl[p] : *(RP_GLOBAL const double*)(f.part + p);
            out[r] = acc;
        }
    }
    __syncthreads();
}
template <int MODE, int RP_SEG_DEPTH, bool HU = false>
__device__ __forceinline__ void seg_pass(const Fit1& f, double* out, double mu_xe, bool store_x) {
    const long long t0_ = f.prof ? (long long)__builtin_readcyclecounter() : 0;
    if (MODE == 1 && mu_xe != 0.0) { for (int sgm = threadIdx.x; sgm < f.nseg; sgm += blockDim.x) seg_body<1, RP_SEG_DEPTH, false, true, HU>(f, sgm, mu_xe, store_x); }
    else for (int sgm = threadIdx.x; sgm < f.nseg; sgm += blockDim.x) seg_body<MODE, RP_SEG_DEPTH, false, false, HU>(f, sgm, mu_xe, store_x);
    const long long t1_ = f.prof ? (long long)__builtin_readcyclecounter() : 0;
    __syncthreads();
    const long long t2_ = f.prof ? (long long)__builtin_readcyclecounter() : 0;
    seg_row_sums<false>(f, out);
    if (f.prof && threadIdx.x == 0 && blockIdx.y == 0) {     // (experiments build) thread 0's segments | barrier = the slowest wave + the partial sums' stores | row sums
        f.prof[12] += t1_ - t0_; f.prof[13] += t2_ - t1_; f.prof[14] += (long long)__builtin_readcyclecounter() - t2_;
    }
}

// claim-and-process loop of one product (leader and helpers).  word = the product's control word with next chunk = 1.
// Shape matters: ONE thread-0 block per iteration (count the finished chunk, claim the next), every barrier and the loop exit in
// wave-uniform control flow (the chunk index goes through readfirstlane).  With a second thread-0 block at the end of the body the
// compiler threads thread 0 from there straight into the next claim, the loop becomes irreducible, and wave 0's other lanes reach the
// barrier -- and read the chunk index -- before lane 0 has claimed anything (seen as a memory fault on a garbage chunk index).
//   LEADER: starts with chunk 0, then takes whatever is unclaimed and leaves when every other chunk is counted done;
//   helpers: LOADER() = their vector loads, issued while the first claim is in flight; they leave when nothing is left to claim.
template <int DEPTH, bool LEADER, class LOADER>
__device__ __forceinline__ void fit_work_loop(const Fit1& f, rp_u64 word, int* s_chunk, LOADER loader) {
    const int csz = rp_fit_chunk_size(f.nseg, f.G);
    const unsigned nchunks = (unsigned)rp_fit_chunk_count(f.nseg, csz);
    int prev = -1;
    for (int it = 0;; ++it) {
        if (threadIdx.x == 0) {
            if (prev > 0) { rp_release_agent(); __hip_atomic_fetch_add(&f.ctl->done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // (its stores were drained below; chunk 0 is not counted)
            int got = -1;
            if (LEADER && it == 0) got = 0;
            else {
                const long long t0 = (long long)__builtin_readcyclecounter();
                rp_u64 cur = (!LEADER && it == 0) ? word : rp_ld_sc1(&f.ctl->claim);
                for (;;) {
                    if ((cur >> 32) != (word >> 32)) break;                              // (helpers: the product is over)
                    if ((unsigned)cur < nchunks) {
                        const rp_u64 old = atomicCAS(&f.ctl->claim, cur, cur + 1);      // (device scope, relaxed)
                        if (old == cur) { got = (int)(unsigned)cur; break; }
                        cur = old;
                        continue;
                    }
                    if (!LEADER) break;                                                 // nothing left to claim
                    // leader: every chunk is claimed; wait until the nchunks - 1 claimed ones are counted done
                    if (rp_ld_sc1(&f.ctl->done) == ((word >> 40 << 40) | (rp_u64)(nchunks - 1))) { rp_acquire_agent(); break; }     // (then reads part2)
                    __builtin_amdgcn_s_sleep(1);
                    // ~2 s without the claimed chunks being counted: never a hang -- and never a silently wrong product either: -2 makes the
                    // leader redo every chunk itself (below) and finish the fit without helpers
                    if ((long long)__builtin_readcyclecounter() - t0 > (1ll << 32)) { got = -2; break; }
                }
            }
            *s_chunk = got;
        }
        if (it == 0) loader();
        __syncthreads();
        const int ch = __builtin_amdgcn_readfirstlane(*s_chunk);
        if (LEADER && ch == -2) {
            // A claimed chunk was not counted in time (its workgroup descheduled, or lost): entries of part2 may be stale.  A partial sum
            // belongs to its segment, not to whoever computed it, so the leader recomputes every chunk but its own chunk 0 -- a late helper
            // storing the same values over them is harmless -- tells the helpers to leave and runs the remaining products alone (they
            // use `part`, which no helper ever writes; a late bump of ctl->done is never read again).
            for (int sgm = rp_fit_chunk_begin(1, csz) + threadIdx.x; sgm < f.nseg; sgm += blockDim.x) seg_body<1, DEPTH, true>(f, sgm, 0.0, false);
            rp_drain_stores();
            if (threadIdx.x == 0) { f.epoch[3] = 1; rp_st_sc1(&f.ctl->claim, (rp_u64)RP_FIT_DONE << 40); }
            break;
        }
        if (ch < 0) break;
        const int s0 = rp_fit_chunk_begin(ch, csz), s1 = min(f.nseg, rp_fit_chunk_begin(ch + 1, csz));
        for (int sgm = s0 + threadIdx.x; sgm < s1; sgm += blockDim.x) seg_body<1, DEPTH, true>(f, sgm, 0.0, false);
        rp_drain_stores();
        __syncthreads();        // every wave's partial sums are out (and everyone has read *s_chunk) before thread 0 counts the chunk
        prev = ch;
    }
    __syncthreads();            // *s_chunk may be rewritten by the caller
}

// the leader's product with helpers: publish u (f.vec), work + wait, row sums.  (h is published by fit_publish_h once per round.)
template <int DEPTH>
__device__ __forceinline__ void fit_dist_product(const Fit1& f, double* out, int* s_chunk) {
    const bool pr = f.prof && threadIdx.x == 0 && blockIdx.y == 0;
    const long long ta_ = pr ? (long long)__builtin_readcyclecounter() : 0;
    const unsigned e = f.epoch[0] + 1;
    const rp_u64 word = ((rp_u64)e << 40) | ((rp_u64)(f.epoch[1] & 0xff) << 32) | 1ull;
    for (int c = threadIdx.x; c < f.C; c += blockDim.x) rp_st_sc1(f.xu + c, f.vec[c]);
    if (threadIdx.x == 0) rp_st_sc1(&f.ctl->done, (rp_u64)e << 40);
    rp_drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) { f.epoch[0] = e; rp_release_agent(); rp_st_sc1(&f.ctl->claim, word); }
    const long long tb_ = pr ? (long long)__builtin_readcyclecounter() : 0;
    fit_work_loop<DEPTH, true>(f, word, s_chunk, [] {});
    const long long td_ = pr ? (long long)__builtin_readcyclecounter() : 0;
    seg_row_sums<true>(f, out);
    if (pr) { f.prof[8] += tb_ - ta_; f.prof[9] += td_ - tb_; f.prof[11] += (long long)__builtin_readcyclecounter() - td_; }
}
__device__ __forceinline__ void fit_publish_h(const Fit1& f) {
    for (int c = threadIdx.x; c < f.C; c += blockDim.x) rp_st_sc1(f.xu + f.Cmax + c, f.hh[c]);
    rp_drain_stores();
    if (threadIdx.x == 0) f.epoch[1] += 1;      // (the version travels in the next product's control word)
    __syncthreads();
}

__device__ __forceinline__ double rp_readlane_f64(double v, int l) {       // l: wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// fast a / b for the Sturm counts (sign and rough size matter, not the last bits)
__device__ __forceinline__ double rp_fast_div(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return a * r;
}

// Largest eigenpair of the symmetric tridiagonal T (alpha[0..m), beta[0..m-1)) on ONE wave: eigenvalue by 64-way
// multisection on Sturm counts, eigenvector by inverse iteration with theta shifted just above the spectrum
// (theta I - T is then positive definite: LDL^T without pivoting).  s is normalised.  Returns theta.
__device__ double tridiag_top(const double* alpha, const double* beta, int m, double* s, int rounds, long long* prof) {
    const int lane = threadIdx.x & 63;
    const long long tp0_ = prof ? (long long)__builtin_readcyclecounter() : 0;
    double lo = -INFINITY, hi = -INFINITY, scale = 0.0;
    for (int k = 0; k < m; ++k) {
        const double bl = k > 0 ? fabs(beta[k - 1]) : 0.0, br = k < m - 1 ? fabs(beta[k]) : 0.0;
        lo = fmax(lo, alpha[k]);                                       // lambda_max >= every diagonal entry
        hi = fmax(hi, alpha[k] + bl + br);                              // Gershgorin
        scale = fmax(scale, fabs(alpha[k]) + bl + br);
    }
    if (!(scale > 0.0)) { for (int k = lane; k < m; k += 64) s[k] = (k == 0) ? 1.0 : 0.0; return 0.0; }
    const double tiny = scale * 1e-300 + 1e-300;
    lo -= scale * 1e-15; hi += scale * 1e-15;
    for (int round = 0; round < rounds; ++round) {
        const double x = lo + (hi - lo) * ((double)(lane + 1) / 64.0);          // lane 63 tests hi itself
        int cnt = 0;
        double q = alpha[0] - x;
        if (q == 0.0) q = -tiny;
        cnt += q < 0.0;
        for (int k = 1; k < m; ++k) {
            q = (alpha[k] - x) - rp_fast_div(beta[k - 1] * beta[k - 1], q);
            if (q == 0.0) q = -tiny;
            cnt += q < 0.0;
        }
        const unsigned long long all = __ballot(cnt == m);                     // x is above the whole spectrum
        const int first = all ? __ffsll((long long)all) - 1 : 63;
        const double nlo = (first == 0) ? lo : lo + (hi - lo) * ((double)first / 64.0);
        const double nhi = lo + (hi - lo) * ((double)(first + 1) / 64.0);
        lo = nlo; hi = nhi;
    }
    // polish: lambda_max is the largest root of the last pivot d_m(x) of x I - T, which is increasing and concave to the right of
    // lambda_max(T_{m-1}); Newton from the LEFT end of the bracket therefore climbs monotonically to the root.  d and d' by the
    // pivot recurrence d_k = (x - alpha_k) - beta_{k-1}^2 / d_{k-1}.  (All lanes compute the same numbers.)
    double theta = hi;
    const long long tp1_ = prof ? (long long)__builtin_readcyclecounter() : 0;
    if (rounds < 9) {
        double x = lo;
        for (int it = 0; it < 8; ++it) {
            double d = x - alpha[0], dd = 1.0;
            bool ok = d > 0.0;
            for (int k = 1; k < m && ok; ++k) {
                const double b2 = beta[k - 1] * beta[k - 1], id = 1.0 / d;
                dd = 1.0 + b2 * dd * id * id;
                d = (x - alpha[k]) - b2 * id;
                ok = (k == m - 1) || d > 0.0;                                  // inner pivots must stay positive (x > lambda_max(T_{m-1}))
            }
            if (!ok) break;                                                     // bracket end below lambda_max(T_{m-1}): keep the multisection result
            const double xn = x - d / dd;
            if (!(xn > x) || !(xn <= hi)) { if (xn == x) theta = x; break; }
            x = xn; theta = x;
            if (fabs(d) <= 4e-16 * scale * dd) break;
        }
        if (!(theta >= lo && theta <= hi)) theta = hi;
    }
    const long long tp2_ = prof ? (long long)__builtin_readcyclecounter() : 0;
    {
        // Inverse iteration, one matrix row per LANE (m <= 64): the serial sweeps of the bidiagonal solves pass s_k from lane to lane
        // through v_readlane (a scalar register, ~10 cycles) -- the same operations in the same order as the one-lane loop over LDS
        // arrays this replaces, whose every step paid an LDS store -> load round trip (554 k of the 856 k cycles of all tests of a fit
        // at N = 200; profiles/r04_matcher.txt).  The pivots are computed redundantly by every lane (uniform operands, like the polish).
        const double sh = theta + scale * 4e-16;
        double idk = 0.0, idprev = 0.0;                  // reciprocal pivots of theta I - T: idk = lane's own
        for (int k = 0; k < m; ++k) {
            double d = (k == 0) ? sh - alpha[0] : (sh - alpha[k]) - (beta[k - 1] * beta[k - 1]) * idprev;
            if (k == 0) { if (!(d > tiny)) d = tiny; }
            else if (!(d > scale * 1e-18)) d = scale * 1e-18;
            idprev = 1.0 / d;
            if (lane == k) idk = idprev;
        }
        const double idm1 = __shfl_up(idk, 1);
        // (theta I - T) = L D L^T with L unit lower bidiagonal, l_k = -beta_k / d_k
        const double ck = (lane >= 1 && lane < m) ? beta[lane - 1] * idm1 : 0.0;        // forward:  s_k += (beta_{k-1} / d_{k-1}) s_{k-1}
        const double ek = (lane < m - 1) ? beta[lane] * idk : 0.0;                      // backward: s_k += (beta_k / d_k) s_{k+1}
        double sk = 1.0;
        for (int it = 0; it < 3; ++it) {
            for (int k = 1; k < m; ++k) { const double prev = rp_readlane_f64(sk, k - 1); if (lane == k) sk = sk + ck * prev; }      // L z = b
            sk = sk * idk;                                                                                                        // D
            for (int k = m - 2; k >= 0; --k) { const double nxt = rp_readlane_f64(sk, k + 1); if (lane == k) sk = sk + ek * nxt; }  // L^T
            const double sq = sk * sk;
            double nn = 0.0;
            for (int k = 0; k < m; ++k) nn += rp_readlane_f64(sq, k);
            nn = 1.0 / sqrt(nn);
            sk *= nn;
        }
        if (lane < m) s[lane] = sk;
    }
    if (prof && lane == 0 && blockIdx.y == 0) {      // (experiments build) multisection | Newton polish | inverse iteration
        prof[3] += tp1_ - tp0_; prof[10] += tp2_ - tp1_; prof[15] += (long long)__builtin_readcyclecounter() - tp2_;
    }
    return theta;
}

// Leading eigenvector of the pair's matrix (seg_pass MODE 1) into f.vec, starting from the unit vector already in f.vec;
// returns the number of products.  *converged = 0 when the residual estimate stays above tolerance.
// (out of line: the logarithms are evaluated a few times per eigen-solve and must not cost the Lanczos loop registers)
__device__ __attribute__((noinline)) double lz_rate(double r_a, int m_a, double r_b, int m_b, double prev) { return rp_lz_rate(r_a, m_a, r_b, m_b, prev); }
__device__ __attribute__((noinline)) int lz_steps_to_check(double r, double lrate, int max_steps) { return rp_lz_steps_to_check(r, lrate, RP_LZ_TOL, max_steps); }

template <int DEPTH, int VB, bool HU>    // HU: the products gather {h, u} pairs (Fit1::hu, kept up to date here); VB: basis-vector entries a lane keeps in flight in the re-orthogonalisation (16 in the 512-thread kernel, 8 at 1024 threads: 128 VGPRs)
__device__ int lanczos_top(const Fit1& f, double mu_xe, int* converged, double* lrate) {
    const int C = f.C, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    int nprod = 0;
    *converged = 1;
    while (true) {
        if (f.KL > 0) { for (int c = tid; c < C; c += blockDim.x) f.Vl[c] = f.vec[c]; }
        else for (int c = tid; c < C; c += blockDim.x) f.V[c] = f.vec[c];
        if constexpr (HU) for (int c = tid; c < C; c += blockDim.x) *reinterpret_cast<double2*>(f.hu + 2 * c) = make_double2(f.hh[c], f.vec[c]);
        __syncthreads();
        int m = 0;
        double beta_last = 0.0, theta = 0.0;
        bool done = false;
        // where the next convergence test goes (rp_lz_steps_to_check in rp_math.h; every thread computes the same numbers from the
        // same LDS values): (m_a, r_a) = the last data point of the relative residual estimate
        int next_check = RP_LZ_CHECK, m_a = 0;
        double r_a = 0.0;
        for (int j = 0; j < RP_LZ_M && !done; ++j) {
            long long t0_ = f.prof ? (long long)__builtin_readcyclecounter() : 0;
            if (f.G > 1 && f.epoch[3] == 0) fit_dist_product<DEPTH>(f, f.yy, (int*)(f.epoch + 2));     // yy = A v_j with the helper workgroups
            else seg_pass<1, DEPTH, HU>(f, f.yy, mu_xe, false);                            // yy = A v_j   (barriers inside)
            ++nprod;
            long long t1_ = f.prof ? (long long)__builtin_readcyclecounter() : 0;
            // classical Gram-Schmidt against v_0..v_j (its coefficient of v_j is alpha_j); a second pass only when the
            // first one removed more than half of the vector (Daniel-Gragg-Kaufman-Stewart criterion)
            double alpha = 0.0, nrm_before = 0.0, nrm_after = 0.0;
            for (int pass = 0; pass < 2; ++pass) {
                for (int i = wave; i <= j; i += nw) {
                    const double* vi = f.V + (size_t)i * f.Cmax;
                    double d = 0.0;
                    if (i < f.KL) {                                        // (wave-uniform) an LDS-resident basis vector: the same sum in the same order
                        const double* vl = f.Vl + (size_t)i * f.Cmax;
                        for (int c0 = lane; c0 < C; c0 += 64 * 8) {       // 8 + 8 LDS reads in flight, then accumulated in order
                            double v8[8], y8[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) { const bool ok = c0 + 64 * q < C; v8[q] = ok ? vl[c0 + 64 * q] : 0.0; y8[q] = ok ? f.yy[c0 + 64 * q] : 0.0; }
#pragma unroll
                            for (int q = 0; q < 8; ++q) if (c0 + 64 * q < C) d += v8[q] * y8[q];
                        }
                    } else
                    for (int c0 = lane; c0 < C; c0 += 64 * VB) {          // VB loads in flight (one L2 round trip for C <= 1024 with VB = 16), then accumulated in order
                        double v8[VB];
#pragma unroll
                        for (int q = 0; q < VB; ++q) v8[q] = (c0 + 64 * q < C) ? rp_ldg(vi + c0 + 64 * q) : 0.0;
#pragma unroll
                        for (int q = 0; q < VB; ++q) if (c0 + 64 * q < C) d += v8[q] * f.yy[c0 + 64 * q];
                    }
                    d = rp_wave_sum(d);
                    if (lane == 0) f.cbuf[i] = d;
                }
                double nn[2] = {0.0, 0.0};
                if (pass == 0) for (int c = tid; c < C; c += blockDim.x) nn[0] += f.yy[c] * f.yy[c];
                __syncthreads();
                alpha += f.cbuf[j];
                if constexpr (VB == 16) {
                    // two entries of the vector per thread at a time, 8 basis vectors' entries of each in flight (the same sums in the
                    // same order as below; half the L2 round trips)
                    for (int c = tid; c < C; c += 2 * blockDim.x) {
                        const int c2 = c + blockDim.x;
                        const bool two = c2 < C;
                        double acc = f.yy[c], acc2 = two ? f.yy[c2] : 0.0;
                        const int jl = min(j + 1, f.KL);                  // vectors 0 .. jl-1 from LDS (4 at a time), the rest from global (8 in flight)
                        for (int i0 = 0; i0 < jl; i0 += 4) {
                            double v4[4], w4[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                v4[q] = (i0 + q < jl) ? f.Vl[(size_t)(i0 + q) * f.Cmax + c] : 0.0;
                                w4[q] = (two && i0 + q < jl) ? f.Vl[(size_t)(i0 + q) * f.Cmax + c2] : 0.0;
                            }
#pragma unroll
                            for (int q = 0; q < 4; ++q) if (i0 + q < jl) { acc -= f.cbuf[i0 + q] * v4[q]; acc2 -= f.cbuf[i0 + q] * w4[q]; }
                        }
                        for (int i0 = f.KL; i0 <= j; i0 += 8) {
                            double v8[8], w8[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                v8[q] = (i0 + q <= j) ? rp_ldg(f.V + (size_t)(i0 + q) * f.Cmax + c) : 0.0;
                                w8[q] = (two && i0 + q <= j) ? rp_ldg(f.V + (size_t)(i0 + q) * f.Cmax + c2) : 0.0;
                            }
#pragma unroll
                            for (int q = 0; q < 8; ++q) if (i0 + q <= j) { acc -= f.cbuf[i0 + q] * v8[q]; acc2 -= f.cbuf[i0 + q] * w8[q]; }
                        }
                        f.yy[c] = acc;
                        nn[1] += acc * acc;
                        if (two) { f.yy[c2] = acc2; nn[1] += acc2 * acc2; }
                    }
                } else
                for (int c = tid; c < C; c += blockDim.x) {
                    double acc = f.yy[c];
                    const int jl = min(j + 1, f.KL);
                    for (int i0 = 0; i0 < jl; i0 += 4) {
                        double v4[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v4[q] = (i0 + q < jl) ? f.Vl[(size_t)(i0 + q) * f.Cmax + c] : 0.0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (i0 + q < jl) acc -= f.cbuf[i0 + q] * v4[q];
                    }
                    for (int i0 = f.KL; i0 <= j; i0 += 8) {                // 8 basis vectors' entries in flight
                        double v8[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v8[q] = (i0 + q <= j) ? rp_ldg(f.V + (size_t)(i0 + q) * f.Cmax + c) : 0.0;
#pragma unroll
                        for (int q = 0; q < 8; ++q) if (i0 + q <= j) acc -= f.cbuf[i0 + q] * v8[q];
                    }
                    f.yy[c] = acc;
                    nn[1] += acc * acc;
                }
                rp_block_sum<2>(nn, f.red);                                        // (barriers inside)
                if (pass == 0) nrm_before = nn[0];
                nrm_after = nn[1];
                if (pass == 0 && nrm_after > 0.25 * nrm_before) break;            // |w'| > |w| / 2: orthogonal enough
            }
            const double beta = sqrt(nrm_after);
            if (tid == 0) { f.tri[j] = alpha; f.tri[(RP_LZ_M + 1) + j] = beta; }
            if (f.prof && tid == 0 && blockIdx.y == 0) { f.prof[0] += t1_ - t0_; f.prof[1] += (long long)__builtin_readcyclecounter() - t1_; f.prof[6] += 1; }
            m = j + 1;
            beta_last = beta;
            const double anorm = fabs(alpha) + beta;
            const bool invariant = !(beta > 1e-14 * anorm);                        // invariant subspace (or the zero matrix): exact
            if (!invariant && j + 1 < RP_LZ_M) {
                const double inv = 1.0 / beta;
                double* vn = f.V + (size_t)(j + 1) * f.Cmax;
                if (j + 1 < f.KL) {
                    double* vln = f.Vl + (size_t)(j + 1) * f.Cmax;
                    for (int c = tid; c < C; c += blockDim.x) { const double v = f.yy[c] * inv; vln[c] = v; f.vec[c] = v; if (HU) f.hu[2 * c + 1] = v; }
                } else
                for (int c = tid; c < C; c += blockDim.x) { const double v = f.yy[c] * inv; rp_stg(vn + c, v); f.vec[c] = v; if (HU) f.hu[2 * c + 1] = v; }
            }
            __syncthreads();
            if (j == 0 && !f.fixed_checks) {                                       // the start vector's own residual: free
                r_a = beta / fabs(alpha); m_a = 1;
                next_check = (r_a <= RP_LZ_TOL) ? 1 : 1 + lz_steps_to_check(r_a, *lrate, RP_LZ_CHECK - 1);
            }
            if (invariant || m == RP_LZ_M || m == next_check || nprod >= f.max_prod) {
                // Ritz pair of the m x m tridiagonal matrix, residual estimate beta_m |s_m|
                long long t2_ = f.prof ? (long long)__builtin_readcyclecounter() : 0;
                if (wave == 0) {
                    const double th = tridiag_top(f.tri, f.tri + (RP_LZ_M + 1), m, f.tri + 2 * (RP_LZ_M + 1), f.tri_rounds, f.prof);
                    if (lane == 0) f.red[159] = th;
                }
                __syncthreads();
                theta = f.red[159];
                if (f.prof && tid == 0 && blockIdx.y == 0) f.prof[2] += (long long)__builtin_readcyclecounter() - t2_;
                const double resid = invariant ? 0.0 : beta_last * fabs(f.tri[2 * (RP_LZ_M + 1) + m - 1]);
                if (invariant || m == RP_LZ_M || resid <= RP_LZ_TOL * fabs(theta) || nprod >= f.max_prod) {
                    done = true;
                    if (resid <= RP_LZ_TOL * fabs(theta)) beta_last = 0.0;         // flag: converged
                } else if (f.fixed_checks) next_check = m + RP_LZ_CHECK;
                else {
                    const double r_b = resid / fabs(theta);
                    *lrate = lz_rate(r_a, m_a, r_b, m, *lrate);
                    r_a = r_b; m_a = m;
                    next_check = m + lz_steps_to_check(r_b, *lrate, RP_LZ_CHECK);
                }
                __syncthreads();
            }
        }
        if (!(theta > 0.0) && m == 1) {
            // A v0 = 0 (zero matrix / no active edges): keep the start vector, like the round-1 solver
            if (f.KL > 0) { for (int c = tid; c < C; c += blockDim.x) f.vec[c] = f.Vl[c]; }
            else for (int c = tid; c < C; c += blockDim.x) f.vec[c] = f.V[c];
            __syncthreads();
            return nprod;
        }
        // u = sum_i s_i v_i, normalised
        const double* sv = f.tri + 2 * (RP_LZ_M + 1);
        double nn[1] = {0.0};
        for (int c = tid; c < C; c += blockDim.x) {
            double acc = 0.0;
            const int ml = min(m, f.KL);
            for (int i0 = 0; i0 < ml; i0 += 4) {
                double v4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v4[q] = (i0 + q < ml) ? f.Vl[(size_t)(i0 + q) * f.Cmax + c] : 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (i0 + q < ml) acc += sv[i0 + q] * v4[q];
            }
            for (int i0 = f.KL; i0 < m; i0 += 8) {
                double v8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v8[q] = (i0 + q < m) ? rp_ldg(f.V + (size_t)(i0 + q) * f.Cmax + c) : 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) if (i0 + q < m) acc += sv[i0 + q] * v8[q];
            }
            f.yy[c] = acc;
            nn[0] += acc * acc;
        }
        rp_block_sum<1>(nn, f.red);
        const double inv = 1.0 / sqrt(nn[0]);
        for (int c = tid; c < C; c += blockDim.x) f.vec[c] = f.yy[c] * inv;
        __syncthreads();
        if (beta_last == 0.0) return nprod;                                        // converged (or exact)
        if (nprod + RP_LZ_CHECK > f.max_prod) break;
    }
    *converged = 0;
    return nprod;
}

// Horn's quaternion for the single-workgroup fit, operands and result in LDS, out of line so that its temporaries do not
// set the register budget of the whole kernel: Newton on the characteristic quartic + adjugate eigenvector (rp_math.h),
// the Jacobi solver only when the leading eigenvalue is not separated.
__device__ __attribute__((noinline)) void horn_lds(const double* m9, double* r9) {
    double M[3][3], Rl[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) M[a][bb] = m9[a * 3 + bb];
    rp_horn_rotation_fast(M, Rl);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) r9[a * 3 + bb] = Rl[a][bb];
}

// fit_solve for the single-workgroup kernel: same arithmetic and summation order as fit_solve, Horn through horn_lds
template <int EPT>
__device__ void fit_solve1(const FitCtx& f, bool reweight, double* Rt /* LDS [12]: R row-major, t */) {
    // EPT = 1: the first correspondence of every thread lives in registers across the three passes (12 doubles each); beyond that
    // (C > blockDim) -- and with EPT = 0 for all of them -- the passes re-read them from the gathered geometry table (L2-resident)
    const int nloc = min(EPT, (f.C - (int)threadIdx.x + (int)blockDim.x - 1) / (int)blockDim.x);
    double gsp[EPT + 1][3], gtp[EPT + 1][3], gsn[EPT + 1][3], gtn[EPT + 1][3], gdeg[EPT + 1], ggP[EPT + 1], ggN[EPT + 1];      // (+ 1: no zero-length arrays)
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int c = threadIdx.x + k * blockDim.x;
        if (k < nloc) { corr_geom(f, c, gsp[k], gtp[k], gsn[k], gtn[k]); gdeg[k] = f.deg[c]; ggP[k] = f.gP[c]; ggN[k] = f.gN[c]; }
    }
    const int cont = threadIdx.x + EPT * blockDim.x;          // first correspondence of this thread that is not cached
    double s7[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        if (k < nloc) {
            const double wp = f.mu * gdeg[k] * ggP[k];
            s7[0] += wp;
#pragma unroll
            for (int a = 0; a < 3; ++a) { s7[1 + a] += wp * gsp[k][a]; s7[4 + a] += wp * gtp[k][a]; }
        }
    }
    for (int c = cont; c < f.C; c += blockDim.x) {
        double sp[3], tp[3], sn[3], tn[3];
        corr_geom(f, c, sp, tp, sn, tn);
        const double wp = f.mu * f.deg[c] * f.gP[c];
        s7[0] += wp;
#pragma unroll
        for (int a = 0; a < 3; ++a) { s7[1 + a] += wp * sp[a]; s7[4 + a] += wp * tp[a]; }
    }
    rp_block_sum<7>(s7, f.red);
    const double den = s7[0] + RP_EPS;
    double ms[3], mt[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { ms[a] = s7[1 + a] / den; mt[a] = s7[4 + a] / den; }
    double m9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        if (k < nloc) {
            const double wp = f.mu * gdeg[k] * ggP[k], wn = gdeg[k] * ggN[k];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb)
                    m9[a * 3 + bb] += (gsp[k][a] - ms[a]) * ((gtp[k][bb] - mt[bb]) * wp) + gsn[k][a] * (gtn[k][bb] * wn);
        }
    }
    for (int c = cont; c < f.C; c += blockDim.x) {
        double sp[3], tp[3], sn[3], tn[3];
        corr_geom(f, c, sp, tp, sn, tn);
        const double d = f.deg[c];
        const double wp = f.mu * d * f.gP[c], wn = d * f.gN[c];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb)
                m9[a * 3 + bb] += (sp[a] - ms[a]) * ((tp[bb] - mt[bb]) * wp) + sn[a] * (tn[bb] * wn);
    }
    rp_block_sum<9>(m9, f.red);
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 9; ++q) f.red[128 + q] = m9[q];
        horn_lds(f.red + 128, Rt);
    }
    __syncthreads();
    double R[3][3], t[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { R[a][0] = Rt[a * 3 + 0]; R[a][1] = Rt[a * 3 + 1]; R[a][2] = Rt[a * 3 + 2]; }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) t[a] = -((R[a][0] * ms[0] + R[a][1] * ms[1]) + R[a][2] * ms[2]) + mt[a];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) Rt[9 + a] = t[a];
    }
    auto residual = [&](int c, const double* sp, const double* tp, const double* sn, const double* tn, double gP, double gN) {
        double rP = 0.0, rN = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double x0 = sp[0] - ms[0], x1 = sp[1] - ms[1], x2 = sp[2] - ms[2];
            const double ep = ((R[a][0] * x0 + R[a][1] * x1) + R[a][2] * x2) - (tp[a] - mt[a]);
            const double en = ((R[a][0] * sn[0] + R[a][1] * sn[1]) + R[a][2] * sn[2]) - tn[a];
            rP += ep * ep; rN += en * en;
        }
        rP *= f.mu;
        f.rsum[c] = rP + rN;
        if (reweight) { f.gP[c] = gP / (1.0 + rP); f.gN[c] = gN / (1.0 + rN); }
    };
#pragma unroll
    for (int k = 0; k < EPT; ++k)
        if (k < nloc) residual(threadIdx.x + k * blockDim.x, gsp[k], gtp[k], gsn[k], gtn[k], ggP[k], ggN[k]);
    for (int c = cont; c < f.C; c += blockDim.x) {
        double sp[3], tp[3], sn[3], tn[3];
        corr_geom(f, c, sp, tp, sn, tn);
        residual(c, sp, tp, sn, tn, f.gP[c], f.gN[c]);
    }
    __syncthreads();
}

__device__ __forceinline__ void write_pose_lds(double* out, const double* Rt) {
    if (threadIdx.x < 16) {
        const int a = threadIdx.x >> 2, q = threadIdx.x & 3;
        out[threadIdx.x] = a == 3 ? (q == 3 ? 1.0 : 0.0) : (q == 3 ? Rt[9 + a] : Rt[a * 3 + q]);
    }
}

#ifndef RP_FIT_EPT
#define RP_FIT_EPT(T_) ((T_) == 512 ? 1 : 0)
#endif
// LAYOUT 0: everything in LDS incl. the {h, u} pairs of the products (Cmax <= RP_FIT1_MAXC_HU); 2: the same without the pairs (up to RP_FIT1_MAXC: they
// would not fit); 1 = GVEC: the three per-correspondence vectors + row / segment pointers in global memory
template <int THREADS, int LAYOUT>
__global__ __launch_bounds__(THREADS) void fit_pair_kernel(RelposeKeypoints kp, Graph g, RpPairConsts kc, int topK, int method,
                                                                    double* __restrict__ lz_basis, double* __restrict__ gvec, int32_t* __restrict__ status,
                                                                    double* __restrict__ pose, double* __restrict__ trace,
                                                                    int32_t* __restrict__ counts_out, int32_t* __restrict__ eig_iters_out,
                                                                    long long* __restrict__ prof, int tri_rounds, FitCtl* __restrict__ ctl_all,
                                                                    double* __restrict__ xu_all, int meta_cap, int basis_lds) {
    constexpr bool GVEC = (LAYOUT == 1), HUL = (LAYOUT == 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[160];
    __shared__ double Rt[12];
    __shared__ int st_s;
    __shared__ unsigned cl_s[4];        // helper-workgroup protocol: leader [0] products / [1] h versions published; helper [0] / [1] the control word as polled (high / low half); [2] claimed chunk; leader [3] helpers given up on (a claimed chunk did not arrive in time)
#ifndef RP_SEG_GQ
#define RP_SEG_GQ(T_) ((T_) == 512 ? 4 : 2)      // measured (matcher alone, B=32): N=200 3.63 ms with per-edge gathers, 3.48 / 3.45 with 8 / 4 together; N=400 6.47, 6.63 / 6.41 / 6.34 with 8 / 4 / 2
#endif
#ifndef RP_SEG_NB
#define RP_SEG_NB(T_) 2
#endif
    constexpr int DEPTH = RP_SEG_NB(THREADS) | (RP_SEG_GQ(THREADS) << 4);      // seg_body's RP_SEG_CFG (depth 4 = the whole segment in flight: measured 13 % slower at 512 threads, spills at 1024)
    const int b = blockIdx.y, tid = threadIdx.x;
    const int G = gridDim.x;            // workgroups per scan pair: 1 leader + G - 1 helpers for the matrix-vector products
    const int C = pair_C(kp, g, b);
    FitCtl* ctl = ctl_all + b;
    if (blockIdx.x > 0) {
        // ---- helper: wait for products, take chunks, leave when the leader says so (or never shows up)
        if constexpr (!GVEC) {
            Fit1 f;
            f.C = C; f.Cmax = g.Cmax;
            f.ctl = ctl; f.xu = xu_all + (size_t)b * (2 * (size_t)g.Cmax + g.seg_cap); f.part2 = f.xu + 2 * (size_t)g.Cmax; f.G = G;
            f.vec = (double*)smem; f.hh = f.vec + g.Cmax;                        // the published u and h
            f.sp = g.segptr + (size_t)b * (g.Cmax + 1);      // (global: read in place)
            f.pp = g.segpart + (size_t)b * (g.Cmax + 1);
            f.meta = nullptr; f.meta_cap = 0;
            const size_t eoffh = (size_t)b * g.estride;
            f.col = g.col + eoffh; f.wv = g.wv + eoffh; f.xe = g.xe + eoffh;
            f.rs_col = __builtin_amdgcn_make_buffer_rsrc((void*)f.col, 0, (int)(g.estride * 4), 0x00020000);
            f.rs_wv = __builtin_amdgcn_make_buffer_rsrc((void*)f.wv, 0, (int)(g.estride * 8), 0x00020000);
            f.rs_xe = __builtin_amdgcn_make_buffer_rsrc((void*)f.xe, 0, (int)(g.estride * 8), 0x00020000);
            f.segrow = g.segrow + (size_t)b * g.seg_cap; f.part = g.part + (size_t)b * g.seg_cap;
            f.nfull = f.sp[C]; f.nseg = f.pp[C];
            unsigned last = 0, hseen = 0;       // product number / h version seen last
            for (;;) {
                if (tid == 0) {
                    const long long t0 = (long long)__builtin_readcyclecounter();
                    rp_u64 w;
                    while ((unsigned)((w = rp_ld_sc1(&ctl->claim)) >> 40) == last) {
                        __builtin_amdgcn_s_sleep(2);
                        if ((long long)__builtin_readcyclecounter() - t0 > (last == 0 ? (1ll << 26) : (1ll << 32))) { w = (rp_u64)RP_FIT_DONE << 40; break; }   // no leader within ~30 ms: leave
                    }
                    rp_acquire_agent();         // (then everyone loads the published vectors)
                    cl_s[0] = (unsigned)(w >> 32); cl_s[1] = (unsigned)w;
                }
                __syncthreads();
                const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)cl_s[0]);       // {product, h version}
                const unsigned cl_next = (unsigned)__builtin_amdgcn_readfirstlane((int)cl_s[1]);  // next unclaimed chunk as polled
                __syncthreads();
                if ((hi >> 8) == RP_FIT_DONE) return;
                last = hi >> 8;
                const bool newh = (hi & 0xff) != hseen;
                hseen = hi & 0xff;
                fit_work_loop<DEPTH, false>(f, ((rp_u64)hi << 32) | (unsigned)cl_next, (int*)&cl_s[2], [&] {
                    for (int c = tid; c < C; c += blockDim.x) f.vec[c] = rp_ld_sc1(f.xu + c);
                    if (newh) for (int c = tid; c < C; c += blockDim.x) f.hh[c] = rp_ld_sc1(f.xu + g.Cmax + c);
                });
            }
        }
        return;
    }
    if (tid == 0) { cl_s[0] = 0; cl_s[1] = 0; cl_s[3] = 0; }
    if (tid == 0) {
        int st = status[b];
        if (st == RELPOSE_OK && g.counters[b * 4 + 2] < 1) st = RELPOSE_ZERO_WEIGHT;
        st_s = st;
        if (counts_out) {
            counts_out[b * 4 + 0] = g.counters[b * 4 + 0];
            counts_out[b * 4 + 1] = g.counters[b * 4 + 1];
            counts_out[b * 4 + 2] = g.counters[b * 4 + 2] / 2;
            counts_out[b * 4 + 3] = (kp.ns[b] >= 3 && kp.nt[b] >= 3) ? g.keff[b] : 0;
        }
    }
    __syncthreads();
    if (st_s != RELPOSE_OK) {                             // identity, like the reference's early returns
        if (tid == 0 && G > 1) rp_st_sc1(&ctl->claim, (rp_u64)RP_FIT_DONE << 40);
        if (tid == 0) status[b] = st_s;
        if (tid < 16) pose[(size_t)b * 16 + tid] = (tid % 5 == 0) ? 1.0 : 0.0;
        if (trace && tid < 96) trace[(size_t)b * 96 + tid] = ((tid % 16) % 5 == 0) ? 1.0 : 0.0;
        if (eig_iters_out && tid < 5) eig_iters_out[b * 5 + tid] = 0;
        return;
    }
    Fit1 f;
    f.prof = prof;
    f.tri_rounds = tri_rounds & 0xff;
    f.max_prod = (tri_rounds >> 8) & 0xffff;
    f.fixed_checks = (tri_rounds >> 24) & 1;
    const long long tstart_ = prof ? (long long)__builtin_readcyclecounter() : 0;
    f.C = C; f.Cmax = g.Cmax;
    f.ctl = ctl; f.xu = xu_all + (size_t)b * (2 * (size_t)g.Cmax + g.seg_cap); f.part2 = f.xu + 2 * (size_t)g.Cmax; f.G = G; f.epoch = &cl_s[0];
    const int32_t* spg = g.segptr + (size_t)b * (g.Cmax + 1);
    const int32_t* ppg = g.segpart + (size_t)b * (g.Cmax + 1);
    // (LDS layout: the {h, u} pairs first -- 16-byte aligned for their ds_read_b128 gathers --, then the tridiagonal scratch, the vectors, ...)
    f.hu = HUL ? (double*)smem : nullptr;
    f.tri = (double*)smem + (HUL ? 2 * (size_t)g.Cmax : 0); f.cbuf = f.tri + 4 * (RP_LZ_M + 1);
    // (a compile-time choice: with a run-time one the compiler no longer knows the address space and emits FLAT accesses for the
    // LDS layout -- measured +50 % on the whole kernel)
    if constexpr (GVEC) {   // more correspondences than LDS holds: the three vectors in global scratch, row / segment pointers read in place
        f.vec = gvec + (size_t)b * 3 * g.Cmax; f.hh = f.vec + g.Cmax; f.yy = f.hh + g.Cmax;
        f.sp = const_cast<int32_t*>(spg); f.pp = const_cast<int32_t*>(ppg);
        f.meta = nullptr; f.meta_cap = 0;
    } else {
        f.vec = f.cbuf + (RP_LZ_M + 1); f.hh = f.vec + g.Cmax; f.yy = f.hh + g.Cmax;
        f.sp = (int32_t*)(f.yy + g.Cmax); f.pp = f.sp + (g.Cmax + 1);
        f.meta = f.pp + (g.Cmax + 1); f.meta_cap = 0;          // (set once the table is filled, below)
    }
    // the first basis_lds Lanczos vectors live in LDS behind the segment table (meta_cap is even: 8-byte aligned); 0 in the global layout
    f.KL = GVEC ? 0 : basis_lds;
    f.partl = GVEC ? nullptr : (double*)(f.meta + meta_cap);
    f.Vl = GVEC ? nullptr : f.partl + meta_cap;
    f.red = red;
    const size_t eoff = (size_t)b * g.estride;
    f.col = g.col + eoff; f.wv = g.wv + eoff; f.xe = g.xe + eoff;
    f.rs_col = __builtin_amdgcn_make_buffer_rsrc((void*)f.col, 0, (int)(g.estride * 4), 0x00020000);
    f.rs_wv = __builtin_amdgcn_make_buffer_rsrc((void*)f.wv, 0, (int)(g.estride * 8), 0x00020000);
    f.rs_xe = __builtin_amdgcn_make_buffer_rsrc((void*)f.xe, 0, (int)(g.estride * 8), 0x00020000);
    f.segrow = g.segrow + (size_t)b * g.seg_cap; f.part = g.part + (size_t)b * g.seg_cap;
    f.V = lz_basis + (size_t)b * (RP_LZ_M + 1) * g.Cmax;
    if constexpr (!GVEC) {
        for (int c = tid; c <= C; c += blockDim.x) { f.sp[c] = spg[c]; f.pp[c] = ppg[c]; }
    }
    __syncthreads();
    f.nfull = f.sp[C]; f.nseg = f.pp[C];
    if constexpr (!GVEC) {
        const int32_t* segrow_g = g.segrow + (size_t)b * g.seg_cap;
        const int nm = min(f.nseg, meta_cap);
        for (int sgm = tid; sgm < nm; sgm += blockDim.x) {
            f.meta[sgm] = segrow_g[sgm];
        }
        __syncthreads();
        f.meta_cap = nm;
    }
    FitCtx fc;
    fc.b = b; fc.C = C; fc.mu = kc.mu;
    fc.deg = g.state + ((size_t)b * 4 + 0) * g.Cmax; fc.gP = g.state + ((size_t)b * 4 + 1) * g.Cmax;
    fc.gN = g.state + ((size_t)b * 4 + 2) * g.Cmax; fc.rsum = g.state + ((size_t)b * 4 + 3) * g.Cmax;
    fc.red = red;
    fc.geo = g.geo + (size_t)b * g.Cmax * 12;
    {   // geometry of every correspondence (gathered once), weighted degrees of the raw pair weights
        const int keff = g.keff[b];
        double* geo = g.geo + (size_t)b * g.Cmax * 12;
        for (int idx = tid; idx < C * 12; idx += blockDim.x) {
            const int c = idx / 12, gl = idx - c * 12;
            const int i = c / keff, kk = c - i * keff;
            const size_t si = (size_t)b * kp.ns_max + i;
            const int j = g.corres_j[si * topK + kk];
            const size_t ti = (size_t)b * kp.nt_max + j;
            const int a = gl % 3, what = gl / 3;
            geo[idx] = what == 0 ? kp.pc_s[si * 3 + a] : what == 1 ? kp.pc_t[ti * 3 + a] : what == 2 ? kp.normal_s[si * 3 + a] : kp.normal_t[ti * 3 + a];
        }
        seg_pass<0, DEPTH>(f, f.yy, 0.0, false);
        for (int c = tid; c < C; c += blockDim.x) { fc.deg[c] = f.yy[c]; fc.gP[c] = 1.0; fc.gN[c] = 1.0; fc.rsum[c] = 0.0; }
        __syncthreads();
    }
    const bool irls0 = (method == RELPOSE_FIT_IRLS_SM || method == RELPOSE_FIT_IRLS);
    long long ti_ = f.prof ? (long long)__builtin_readcyclecounter() : 0;
    for (int it = 0; it < (irls0 ? 5 : 1); ++it) fit_solve1<RP_FIT_EPT(THREADS)>(fc, irls0, Rt);
    if (f.prof && tid == 0 && b == 0) { f.prof[4] += (long long)__builtin_readcyclecounter() - ti_; f.prof[7] = ti_ - tstart_; }
    write_pose_lds(pose + (size_t)b * 16, Rt);
    if (trace) write_pose_lds(trace + (size_t)b * 96, Rt);
    int all_converged = 1;
    if (method == RELPOSE_FIT_IRLS_SM || method == RELPOSE_FIT_SPECTRAL) {
        const bool sm = (method == RELPOSE_FIT_IRLS_SM);
        const double u0 = 1.0 / sqrt((double)C);
        for (int c = tid; c < C; c += blockDim.x) f.vec[c] = u0;          // round 0 starts from the uniform vector
        double lrate = 0.0;                                                // decay of the residual estimate per Lanczos step (log), carried from round to round; 0 = not known yet
        for (int round = 0; round < 5; ++round) {
            for (int c = tid; c < C; c += blockDim.x) { const double v = RP_OFFSET - fc.rsum[c]; f.hh[c] = v < 0.0 ? 0.0 : v; }
            __syncthreads();
            if (G > 1) fit_publish_h(f);
            int conv = 1;
#ifndef RP_FIT_VB
#define RP_FIT_VB(T_) ((T_) == 512 ? 16 : 8)
#endif
            const int np = lanczos_top<DEPTH, RP_FIT_VB(THREADS), HUL>(f, (!sm && round > 0) ? kc.mu : 0.0, &conv, &lrate);       // rounds > 0: warm start from f.vec
            all_converged &= conv;
            if (eig_iters_out && tid == 0) eig_iters_out[b * 5 + round] = np;
            long long tf_ = f.prof ? (long long)__builtin_readcyclecounter() : 0;
            seg_pass<2, DEPTH>(f, f.yy, 0.0, !sm);                                    // x per edge, new weighted degrees
            if (f.prof && tid == 0 && b == 0) f.prof[5] += (long long)__builtin_readcyclecounter() - tf_;
            for (int c = tid; c < C; c += blockDim.x) { fc.deg[c] = f.yy[c]; fc.gP[c] = 1.0; fc.gN[c] = 1.0; }
            __syncthreads();
            long long tj_ = f.prof ? (long long)__builtin_readcyclecounter() : 0;
            for (int it = 0; it < (sm ? 5 : 1); ++it) fit_solve1<RP_FIT_EPT(THREADS)>(fc, sm, Rt);
            if (f.prof && tid == 0 && b == 0) f.prof[4] += (long long)__builtin_readcyclecounter() - tj_;
            write_pose_lds(pose + (size_t)b * 16, Rt);
            if (trace) write_pose_lds(trace + (size_t)b * 96 + (round + 1) * 16, Rt);
        }
    } else if (trace) {
        for (int q = 1; q < 6; ++q) write_pose_lds(trace + (size_t)b * 96 + q * 16, Rt);
    }
    if (tid == 0 && G > 1) rp_st_sc1(&ctl->claim, (rp_u64)RP_FIT_DONE << 40);          // helpers leave
    if (tid == 0) status[b] = all_converged ? RELPOSE_OK : RELPOSE_NOT_CONVERGED;
}

bool kp_ok(const RelposeKeypoints* kp, const RelposeParams* p) {
    return kp && p && kp->B > 0 && kp->ns_max > 0 && kp->nt_max > 0 && kp->nt_max <= RELPOSE_MAX_TARGETS && p->topK >= 1 && p->topK <= RP_MAXK &&
           kp->ns && kp->nt && kp->pc_s && kp->pc_t && kp->normal_s && kp->normal_t && kp->feat_s && kp->feat_t &&
           kp->weight_s && kp->weight_t;
}

// The fit keeps 3 vectors of C doubles + the row / segment pointers of a pair in LDS up to RP_FIT1_MAXC correspondences; beyond
// that (or with RELPOSE_TUNE_FIT_GLOBAL_VECTORS) the same kernel keeps them in global scratch (gvec; rowptr / segptr in place).
static bool fit_in_lds(int32_t Cmax) { return Cmax <= RP_FIT1_MAXC && g_rp_tune[RELPOSE_TUNE_FIT_GLOBAL_VECTORS] == 0; }
static bool fit_hu(int32_t Cmax, bool in_lds) { return in_lds && Cmax <= RP_FIT1_MAXC_HU; }
static size_t fit_lds_bytes(int32_t Cmax, bool in_lds) {
    return (size_t)(5 * (RP_LZ_M + 1)) * 8 + 16 + (in_lds ? (size_t)Cmax * (fit_hu(Cmax, true) ? 40 : 24) + (size_t)(Cmax + 1) * 8 : 0);     // (40 = three vectors + the {h, u} pairs; 8 = the rows' full-segment and partial-segment tables)
    // NB: whoever launches layout 0 must have sized the LDS with fit_hu() true, i.e. Cmax <= RP_FIT1_MAXC_HU (the dispatch below guarantees it)
}
// entries of the LDS segment table (row | length of a pair's first segments) behind that: enough for ~3 segments per row, within the CU's 160 KB
static int fit_meta_cap(int32_t Cmax, int32_t seg_cap, bool in_lds) {
    if (!in_lds) return 0;
    const long long room = (160 * 1024 - 2048 - (long long)fit_lds_bytes(Cmax, true)) / 12;     // 4 bytes of table + 8 of partial sum per segment
    long long want = 3ll * Cmax + 512;
    if (want > seg_cap) want = seg_cap;
    if (want > room) want = room;
    want &= ~1ll;                                           // (even: the LDS basis vectors behind the table stay 8-byte aligned)
    return want < 0 ? 0 : (int)want;
}
// Lanczos basis vectors kept in LDS behind that (round 6): whatever the CU's 160 KB still hold, a multiple of 4, at most the whole basis.
// N = 200 (Cmax = 1000): 12 of a cycle's <= 24 vectors; N = 400 (Cmax = 2000): 4.  RELPOSE_FIT_BASIS_LDS=0 (experiments build) = the round-5 layout.
static int fit_basis_lds(int32_t Cmax, int meta_cap, bool in_lds) {
    if (!in_lds) return 0;
    static const int cap_env = RP_ENV("RELPOSE_FIT_BASIS_LDS") ? atoi(RP_ENV("RELPOSE_FIT_BASIS_LDS")) : RP_LZ_M;
    const long long room = 160 * 1024 - 2048 - (long long)fit_lds_bytes(Cmax, true) - (long long)meta_cap * 12;
    long long k = room / ((long long)Cmax * 8);
    if (k > cap_env) k = cap_env;
    if (k > RP_LZ_M) k = RP_LZ_M;
    k &= ~3ll;
    return k < 0 ? 0 : (int)k;
}
#define RP_MAX_CORRES RELPOSE_MAX_CORRESPONDENCES      // correspondences per pair (ns_max * topK): the fill kernel's row lists are uint16 in 8 * Cmax bytes of LDS

struct WsLayout {
    size_t corres_j, corres_w, keff, bitmap, upcnt, counters, rowptr, col, wv, xe, state, geo, lz, gvec, segptr, segpart, segrow, part, ctl, xu, total;
    int32_t Cmax, Wmax, seg_cap;
    int64_t max_edges, estride;
};

WsLayout ws_layout(int32_t B, int32_t ns_max, int32_t topK, int64_t max_edges) {
    WsLayout L;
    L.Cmax = ns_max * topK;
    L.Wmax = (L.Cmax + 63) / 64;
    const int64_t worst = (int64_t)L.Cmax * (L.Cmax - 1);
    L.max_edges = (max_edges <= 0 || max_edges > worst) ? worst : max_edges;
    if (L.max_edges < 16) L.max_edges = 16;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += rp_align(bytes); return r; };
    L.corres_j = take((size_t)B * ns_max * topK * 4);
    L.corres_w = take((size_t)B * ns_max * topK * 8);
    L.keff = take((size_t)B * 4);
    L.bitmap = take((size_t)B * L.Cmax * L.Wmax * 8);
    L.upcnt = take((size_t)B * L.Cmax * 4);
    L.counters = take((size_t)B * 4 * 4);
    L.rowptr = take((size_t)B * (L.Cmax + 1) * 4);
    // segment layout: every row wastes less than one 32-entry segment; slices of 64 segments
    L.seg_cap = (int32_t)(((L.max_edges + RP_SEG - 1) / RP_SEG + L.Cmax + 63) / 64 * 64);
    L.estride = (int64_t)L.seg_cap * RP_SEG;
    L.col = take((size_t)B * L.estride * 4);
    L.wv = take((size_t)B * L.estride * 8);
    L.xe = take((size_t)B * L.estride * 8);
    L.segptr = take((size_t)B * (L.Cmax + 1) * 4);
    L.segpart = take((size_t)B * (L.Cmax + 1) * 4);
    L.segrow = take((size_t)B * L.seg_cap * 4);
    L.part = take((size_t)B * L.seg_cap * 8);
    L.state = take((size_t)B * 4 * L.Cmax * 8);
    L.geo = take((size_t)B * L.Cmax * 12 * 8);
    L.lz = take((size_t)B * (RP_LZ_M + 1) * L.Cmax * 8);        // Lanczos basis
    L.gvec = take((size_t)B * 3 * L.Cmax * 8);                  // the fit's three per-correspondence vectors when they do not live in LDS
    L.ctl = take((size_t)B * sizeof(FitCtl));                   // helper workgroups: control block per pair (zeroed per call)
    L.xu = take((size_t)B * (2 * (size_t)L.Cmax + L.seg_cap) * 8);     // ... the published vectors u, h and the products' partial sums
                                                                // (memory that is only ever written write-through: see FitCtl)
    L.total = o;
    return L;
}

}  // namespace

int32_t g_rp_tune[RELPOSE_TUNE_COUNT] = {0};

extern "C" {

int relpose_set_tuning(int32_t key, int32_t value) {
    if (key < 0 || key >= RELPOSE_TUNE_COUNT) return RELPOSE_EINVAL;
    const int32_t old = g_rp_tune[key];
    g_rp_tune[key] = value;
    return old;
}

void relpose_default_params(RelposeParams* p) {
    p->distThre = 0.08; p->distSepThre = 1.5 * 0.08; p->angleThre = 45 / 180. * M_PI;
    p->sigmaAngle1 = 0.523 / 2; p->sigmaAngle2 = 0.523 / 2; p->sigmaDist = 0.08 / 2; p->sigmaFeat = 0.01;
    p->mu = 0.3; p->topK = 5; p->method = RELPOSE_FIT_IRLS_SM;
}

const char* relpose_version(void) { return "relpose-hip 0.1 (gfx950)"; }

#ifdef RP_EXPERIMENTS
int relpose_stream_create_cu_limited(void** stream_out, int32_t n_cus) {
    if (!stream_out) return RELPOSE_EINVAL;
    int dev = 0, ncu = 0;
    RP_HIP(hipGetDevice(&dev));
    RP_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    hipStream_t s = nullptr;
    if (n_cus <= 0 || n_cus >= ncu) {
        RP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    } else {
        uint32_t mask[32];
        const int words = std::min(32, (ncu + 31) / 32);
        for (int w = 0; w < words; ++w) {
            const int lo = 32 * w;
            mask[w] = n_cus >= lo + 32 ? 0xffffffffu : (n_cus > lo ? ((1u << (n_cus - lo)) - 1u) : 0u);
        }
        RP_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
    }
    *stream_out = (void*)s;
    return 0;
}

int relpose_stream_destroy(void* stream) {
    if (!stream) return RELPOSE_EINVAL;
    RP_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
#endif

size_t relpose_match_workspace_bytes(int32_t B, int32_t ns_max, int32_t nt_max, int32_t topK, int64_t max_edges) {
    if (nt_max > RELPOSE_MAX_TARGETS) return 0;
    if (B <= 0 || ns_max <= 0 || topK < 1 || topK > RP_MAXK || (int64_t)ns_max * topK > RP_MAX_CORRES) return 0;
    return ws_layout(B, ns_max, topK, max_edges).total;
}

int relpose_affinity_topk(const RelposeParams* p, const RelposeKeypoints* kp, float* wij, int32_t* corres_j, double* corres_w,
                          int32_t* k_eff, void* stream) {
    if (!kp_ok(kp, p) || !corres_j || !corres_w || !k_eff) return RELPOSE_EINVAL;
    return rp_launch_affinity(*p, *kp, wij, corres_j, corres_w, k_eff, (hipStream_t)stream, 0);
}

int relpose_match_pairs(const RelposeParams* p, const RelposeKeypoints* kp, void* workspace, size_t workspace_bytes, int64_t max_edges,
                        double* pose, int32_t* status, const RelposeMatchDebug* dbg, void* stream) {
    RelposeMatchArgs a;
    memset(&a, 0, sizeof(a));
    a.struct_size = (uint32_t)sizeof(a);
    a.params_host = p; a.kp_host = kp; a.workspace = workspace; a.workspace_bytes = workspace_bytes; a.max_edges = max_edges;
    a.pose = pose; a.status = status; a.debug_host = dbg; a.stream = stream;
    return relpose_match_pairs_ex(&a);
}

int relpose_match_pairs_ex(const RelposeMatchArgs* args) {
    if (!args || args->struct_size < offsetof(RelposeMatchArgs, stream) + sizeof(void*)) return RELPOSE_EINVAL;
    const RelposeParams* p = args->params_host;
    const RelposeKeypoints* kp = args->kp_host;
    void* workspace = args->workspace;
    const size_t workspace_bytes = args->workspace_bytes;
    const int64_t max_edges = args->max_edges;
    double* pose = args->pose;
    int32_t* status = args->status;
    const RelposeMatchDebug* dbg = args->debug_host;
    void* stream = args->stream;
    const int call_cluster = args->fit_cluster;
    const int call_affinity = args->struct_size >= offsetof(RelposeMatchArgs, affinity_kernel) + sizeof(int32_t) ? args->affinity_kernel : 0;
    if (call_cluster < 0 || call_cluster > 8 || call_affinity < 0 || call_affinity > 4) return RELPOSE_EINVAL;
    if (!kp_ok(kp, p) || !workspace || !pose || !status) return RELPOSE_EINVAL;
    if (p->method < 0 || p->method > 3) return RELPOSE_EINVAL;
    if ((int64_t)kp->ns_max * p->topK > RP_MAX_CORRES) return RELPOSE_EINVAL;
    const WsLayout L = ws_layout(kp->B, kp->ns_max, p->topK, max_edges);
    if (workspace_bytes < L.total) return RELPOSE_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int32_t* cj = (int32_t*)(ws + L.corres_j);
    double* cw = (double*)(ws + L.corres_w);
    int32_t* keff = (int32_t*)(ws + L.keff);
    Graph g;
    g.Cmax = L.Cmax; g.Wmax = L.Wmax; g.max_edges = L.max_edges;
    g.corres_j = cj; g.corres_w = cw; g.keff = keff;
    g.bitmap = (unsigned long long*)(ws + L.bitmap);
    g.upcnt = (int32_t*)(ws + L.upcnt); g.counters = (int32_t*)(ws + L.counters);
    g.rowptr = (int32_t*)(ws + L.rowptr); g.col = (int32_t*)(ws + L.col);
    g.wv = (double*)(ws + L.wv); g.xe = (double*)(ws + L.xe); g.state = (double*)(ws + L.state); g.geo = (double*)(ws + L.geo);
    g.seg_cap = L.seg_cap; g.estride = L.estride;
    g.segptr = (int32_t*)(ws + L.segptr); g.segpart = (int32_t*)(ws + L.segpart); g.segrow = (int32_t*)(ws + L.segrow); g.part = (double*)(ws + L.part);
    RP_HIP(hipMemsetAsync(ws + L.counters, 0, (size_t)kp->B * 16, s));
    int rc = rp_launch_affinity(*p, *kp, dbg ? dbg->wij : nullptr, cj, cw, keff, s, call_affinity);
    if (rc) return rc;
    const RpPairConsts kc = rp_make_consts(*p);
    dim3 grid_rows((L.Cmax + 3) / 4, kp->B);
    hipLaunchKernelGGL(pair_tile_kernel, dim3((unsigned)(L.Wmax * (L.Wmax + 1) / 2), kp->B), dim3(256), 0, s, *kp, g, kc, p->topK);
    RP_CHECK_LAUNCH();
    hipLaunchKernelGGL(pair_scan_kernel, dim3(kp->B), dim3(1024), 0, s, *kp, g, status);
    RP_CHECK_LAUNCH();
    // (at the capacity limit the row lists are exactly 64 KB of dynamic LDS, the default cap: raise it explicitly)
    if ((size_t)4 * L.Cmax * sizeof(unsigned short) > 48 * 1024)
        RP_HIP(hipFuncSetAttribute((const void*)pair_fill_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)4 * L.Cmax * sizeof(unsigned short))));
    hipLaunchKernelGGL(pair_fill_rows_kernel, grid_rows, dim3(256), (size_t)4 * L.Cmax * sizeof(unsigned short), s, *kp, g, kc, p->topK, status);
    RP_CHECK_LAUNCH();
    // ---- fit: ONE launch, one workgroup per pair (fit_pair_kernel)
    double* trace = dbg ? dbg->trace : nullptr;
    int32_t* eig_iters = dbg ? dbg->eig_iters : nullptr;
    const int m = p->method;
    {
        const bool in_lds = fit_in_lds(L.Cmax);
        const int meta_cap = fit_meta_cap(L.Cmax, L.seg_cap, in_lds);
        const int basis_lds = fit_basis_lds(L.Cmax, meta_cap, in_lds);
        const size_t lds = fit_lds_bytes(L.Cmax, in_lds) + (size_t)meta_cap * 12 + (size_t)basis_lds * L.Cmax * 8;
        double* gvec = in_lds ? nullptr : (double*)(ws + L.gvec);
        static long long* prof = nullptr;
        if (RP_ENV("RELPOSE_FIT_PROF")) {
            if (!prof) RP_HIP(hipMalloc((void**)&prof, 128));
            RP_HIP(hipMemsetAsync(prof, 0, 128, s));
        }
        // (multisection rounds | product budget << 8); RELPOSE_TUNE_FIT_MAX_PRODUCTS is a test hook: a tiny budget forces RELPOSE_NOT_CONVERGED
        const int tri_rounds = (RP_ENV("RELPOSE_TRI_ROUNDS") ? atoi(RP_ENV("RELPOSE_TRI_ROUNDS")) : RP_TRI_ROUNDS) |
                               ((g_rp_tune[RELPOSE_TUNE_FIT_MAX_PRODUCTS] > 0 ? g_rp_tune[RELPOSE_TUNE_FIT_MAX_PRODUCTS] : RP_LZ_MAXPROD) << 8) |
                               ((g_rp_tune[RELPOSE_TUNE_FIT_FIXED_CHECKS] != 0 ? 1 : 0) << 24);
        // workgroup size: 512 threads (no register spills: IRLS twice as fast) while every thread still owns at most two correspondences; beyond,
        // 768 in the LDS layout (round 6: three waves per SIMD and a 170-register budget -- measured at N = 400, B = 32, same box: 640 threads 5.56 ms,
        // 768 5.38, 896 5.76, 1024 5.66: the edge passes are no slower than with 16 waves and the IRLS spills less; at N = 200: 384 threads 3.12,
        // 512 2.82, 640 2.97, 768 2.98), in the global layout too (N = 400 forced into it: 12.83 vs 13.15 ms at 1024).  RELPOSE_FIT_THREADS = 1024
        // brings the rounds 2-5 kernel back (experiments build).
        const int fit_threads = RP_ENV("RELPOSE_FIT_THREADS") ? atoi(RP_ENV("RELPOSE_FIT_THREADS")) : (L.Cmax <= 1024 ? 512 : 768);
        // helper workgroups per pair for the matrix-vector products (see FitCtl): a LATENCY tool.  Alone on the chip the matcher of 32
        // N = 400 pairs drops from 7.7 to 5.1 ms with 7 helpers per pair, but helpers sit on a CU each for the whole fit, mostly
        // polling, and inside the pipeline they take those CUs from the SCNet kernels of the other slot: configs[2] 462 -> 400 pairs/s,
        // configs[1] 497 -> 475 with 3 helpers (measured, round 3).  So by default only small batches -- at most 32 workgroups, an
        // eighth of the chip: even 2 workgroups per pair at 32 pairs cost configs[2] 1.7 % (505 -> 496) -- get helpers (up to 7 beyond 1024 correspondences per pair, up to 3 from 512), never the 'spectral'
        // method (its per-round edge weights are written by the leader with plain stores) or the global layout.
        // RELPOSE_TUNE_FIT_CLUSTER forces a size (1 = none).
        int G = 1;
        if (in_lds && m != RELPOSE_FIT_SPECTRAL) {
            const int want = call_cluster > 0 ? call_cluster : g_rp_tune[RELPOSE_TUNE_FIT_CLUSTER];     // the call's own choice wins over the test knob
            if (want > 0) G = want > 8 ? 8 : want;
            else if (L.Cmax >= 512) { G = L.Cmax > 1024 ? 8 : 4; while (G > 1 && (long long)kp->B * G > 32) G >>= 1; }
        }
        if (G > 1) RP_HIP(hipMemsetAsync(ws + L.ctl, 0, (size_t)kp->B * sizeof(FitCtl), s));
#define RP_FIT_LAUNCH(T_, G_)                                                                                                          \
        {                                                                                                                              \
            RP_HIP(hipFuncSetAttribute((const void*)fit_pair_kernel<T_, G_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
            hipLaunchKernelGGL((fit_pair_kernel<T_, G_>), dim3(G, kp->B), dim3(T_), lds, s, *kp, g, kc, p->topK, m, (double*)(ws + L.lz), gvec,  \
                               status, pose, trace, dbg ? dbg->counts : nullptr, eig_iters, prof, tri_rounds, (FitCtl*)(ws + L.ctl),      \
                               (double*)(ws + L.xu), meta_cap, basis_lds);                                                              \
        }
        // (512 threads <=> Cmax <= 1024 <=> the {h, u} pairs when in LDS: layout 0; the 1024-thread kernel's LDS layout is 2, without them)
        if (fit_threads == 512 && (fit_hu(L.Cmax, in_lds) || !in_lds)) { if (in_lds) RP_FIT_LAUNCH(512, 0) else RP_FIT_LAUNCH(512, 1) }
#ifdef RP_EXPERIMENTS
        else if (fit_threads == 1024) { if (in_lds) RP_FIT_LAUNCH(1024, 2) else RP_FIT_LAUNCH(1024, 1) }
#endif
        else if (!in_lds) RP_FIT_LAUNCH(768, 1)
        else RP_FIT_LAUNCH(768, 2)
#undef RP_FIT_LAUNCH
        RP_CHECK_LAUNCH();
        if (prof) {       // experiments build only: synchronises
            long long h[16];
            RP_HIP(hipStreamSynchronize(s));
            RP_HIP(hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "[fit prof, pair 0, cycles] products %lld (%lld calls) reorth+norm %lld tridiag %lld irls %lld finish %lld setup %lld | with helpers: publish %lld own chunks %lld wait %lld row sums %lld | edge passes of one workgroup: thread 0's segments %lld barrier %lld row sums %lld | tridiagonal: multisection %lld newton %lld (= 'wait' column without helpers) inverse iteration %lld\n", h[0], h[6], h[1], h[2], h[4], h[5], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[3], h[10], h[15]);
        }
    }
    if (dbg && dbg->corres_j)
        RP_HIP(hipMemcpyAsync(dbg->corres_j, cj, (size_t)kp->B * kp->ns_max * p->topK * 4, hipMemcpyDeviceToDevice, s));
    if (dbg && dbg->corres_w)
        RP_HIP(hipMemcpyAsync(dbg->corres_w, cw, (size_t)kp->B * kp->ns_max * p->topK * 8, hipMemcpyDeviceToDevice, s));
    return 0;
}

}  // extern "C"
