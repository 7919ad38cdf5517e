// N x N descriptor affinity + row top-K on gfx950 (reference RPModule/rpmodule.py:342-379), batched over scan pairs.
//
//   wij[i,j]  = exp(-dij / (2 (sigma_ij / 5)^2)), rows L2-normalised (zero-norm rows -> 0)     rpmodule.py:354-363
//   dij       = |fs_i/100 - ft_j/100|^2 in float32, summed in numpy's order (8 strided partial sums + a fixed tree)
//   corres    = the K largest entries of every row, ties to the smaller j                        rpmodule.py:367-379
//
// Three kernels with identical results, chosen by problem size (rp_launch_affinity):
//   affinity_rows_kernel   small batches: a lane owns up to 8 targets in registers, the row is broadcast through SGPRs
//   affinity_tile_kernel   large batches: approximate distances on the matrix pipe (fp16 MFMA) find the few entries per row
//                          that matter; only those get the exact numpy-order arithmetic; wij rows are written once
//   affinity_lds_kernel    nt_max > 512: targets transposed in LDS (up to RELPOSE_MAX_TARGETS = 1152 targets: 136 B each + 128 B < 160 KB of LDS)
// The bound is HBM by SURVEY 8(d)'s definition: (Ns + Nt) * 33 * 4 + Ns * Nt * 4 algorithmic bytes per pair-step.
//
// Compiled with -ffp-contract=off: the float32 distance must round like numpy.
#include "matcher_internal.h"
#include <limits.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <mutex>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
// ---- LDS variant (nt_max > 512) ----------------------------------------------------------------------------------------
__device__ __forceinline__ float desc_dist(float fs, const float* ftT, int ldt, int j) {
    float r[8];
#pragma unroll
    for (int c = 0; c < RP_FEAT; ++c) {
        float s = __shfl(fs, c, 64);
        float df = s - ftT[c * ldt + j];
        float sq = df * df;
        if (c < 8) r[c] = sq; else r[c & 7] = r[c & 7] + sq;
    }
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

template <bool WRITE_WIJ>
__global__ __launch_bounds__(256) void affinity_lds_kernel(RelposeKeypoints kp, RpPairConsts kc, int topK, int rows_per_block,
                                                             float* __restrict__ wij, int32_t* __restrict__ corres_j,
                                                             double* __restrict__ corres_w, int32_t* __restrict__ keff_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y;
    const int ns = kp.ns[b], nt = kp.nt[b];
    const int ntp = (kp.nt_max + 63) & ~63;
    const int ldt = ntp + 1;
    double* wt_s = (double*)smem;                       // [ntp]
    float* ftT = (float*)(smem + (size_t)ntp * 8);      // [32][ldt]
    const int keff = (ns >= 3 && nt >= 3) ? min(topK, nt - 1) : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) keff_out[b] = keff;
    if (keff == 0) return;
    const float* ft = kp.feat_t + (size_t)b * kp.nt_max * RP_FEAT;
    for (int idx = threadIdx.x; idx < ntp * RP_FEAT; idx += 256) {
        int j = idx >> 5, c = idx & 31;
        ftT[c * ldt + j] = (j < nt) ? ft[(size_t)j * RP_FEAT + c] / 100.0f : 0.0f;
    }
    for (int j = threadIdx.x; j < ntp; j += 256) wt_s[j] = (j < nt) ? kp.weight_t[(size_t)b * kp.nt_max + j] : 0.0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int rr = wave; rr < rows_per_block; rr += 4) {
        const int i = blockIdx.x * rows_per_block + rr;
        if (i >= ns) break;
        const size_t si = (size_t)b * kp.ns_max + i;
        const float fs = (lane < RP_FEAT) ? kp.feat_s[si * RP_FEAT + lane] / 100.0f : 0.0f;
        const double wsi = kp.weight_s[si];
        double te[RP_MAXK];
        int tj[RP_MAXK];
#pragma unroll
        for (int q = 0; q < RP_MAXK; ++q) { te[q] = -INFINITY; tj[q] = INT_MAX; }
        double sumsq = 0.0;
        for (int j0 = 0; j0 < nt; j0 += 64) {
            const int j = j0 + lane;
            const bool valid = j < nt;
            const int jj = valid ? j : 0;
            const float d = desc_dist(fs, ftT, ldt, jj);
            const double den = (wsi * wt_s[jj] == 1.0) ? kc.den_both : kc.den_other;
            const double e = (-(double)d) / den;
            const double w = exp(e);
            if (valid) {
                sumsq += w * w;
                if (e > te[RP_MAXK - 1]) {            // strict: equal e keeps the smaller (earlier) j
                    te[RP_MAXK - 1] = e; tj[RP_MAXK - 1] = j;
#pragma unroll
                    for (int q = RP_MAXK - 1; q > 0; --q) {
                        if (te[q] > te[q - 1]) {
                            double t0 = te[q]; te[q] = te[q - 1]; te[q - 1] = t0;
                            int t1 = tj[q]; tj[q] = tj[q - 1]; tj[q - 1] = t1;
                        }
                    }
                }
            }
        }
        const double nm = sqrt(rp_wave_sum(sumsq));
        for (int k = 0; k < keff; ++k) {
            double be = te[0];
            int bj = tj[0];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                double oe = rp_shfl_xor_d(be, m);
                int oj = __shfl_xor(bj, m, 64);
                if (oe > be || (oe == be && oj < bj)) { be = oe; bj = oj; }
            }
            if (tj[0] == bj && bj != INT_MAX) {       // this lane owned the winner: pop it
#pragma unroll
                for (int q = 0; q < RP_MAXK - 1; ++q) { te[q] = te[q + 1]; tj[q] = tj[q + 1]; }
                te[RP_MAXK - 1] = -INFINITY; tj[RP_MAXK - 1] = INT_MAX;
            }
            if (lane == 0) {
                const bool ok = bj >= 0 && bj < nt;
                corres_j[si * topK + k] = ok ? bj : 0;
                corres_w[si * topK + k] = (ok && nm != 0.0) ? exp(be) / nm : 0.0;
            }
        }
        if (WRITE_WIJ) {
            float* row = wij + si * kp.nt_max;
            for (int j0 = 0; j0 < nt; j0 += 64) {
                const int j = j0 + lane;
                const bool valid = j < nt;
                const int jj = valid ? j : 0;
                const float d = desc_dist(fs, ftT, ldt, jj);
                const double den = (wsi * wt_s[jj] == 1.0) ? kc.den_both : kc.den_other;
                const double w = exp((-(double)d) / den);
                if (valid) row[j] = (nm != 0.0) ? (float)(w / nm) : 0.0f;
            }
        }
    }
}

// ---- register-resident variant (the default for nt_max <= 512) -----------------------------------------------------
// A lane OWNS up to T targets (j = t*64 + lane) with their scaled 32-float descriptors in VGPRs; a wave walks over
// `rows_per_wave` source rows, broadcasting the row's descriptor through SGPRs (v_readlane), so an entry costs no LDS
// traffic at all: only the 32 x {sub, mul, add} of the numpy-order float32 distance, as packed fp32 math over two target
// slots.  Per row: e = -d/den (float64 division replaced by Markstein's exact q + fma(rem, 1/den, q) sequence; den takes
// two values), the K winners by K rounds of {per-lane best, DPP wave maximum, owner pops}, exp() only in the target slots
// where some lane is within 110 of the row maximum (everything below is < 2^-150 relative: exactly 0 in the float32 wij and
// invisible in the float64 row norm), the norm, the K outputs (exp + divide on K lanes in parallel) and, if wanted, wij.
struct AffConsts { double den[2], rden[2]; int exact_div; };
// wij entries more than RP_AFF_WINDOW below the row's best exponent (< e^-75 = 2.7e-33 of the row maximum) are written as exact zeros and
// left out of the float64 row norm (they change it by < 1e-65 relative): exp() is evaluated only inside the window
#define RP_AFF_WINDOW 75.0        // [0] = other, [1] = both observed

__device__ __forceinline__ double rp_wave_max_d(double v) {
    // butterfly inside every row of 16 lanes (DPP), then the four row results through SGPRs
    v = fmax(v, rp_dpp_d<0xB1>(v));          // quad_perm [1,0,3,2]
    v = fmax(v, rp_dpp_d<0x4E>(v));          // quad_perm [2,3,0,1]
    v = fmax(v, rp_dpp_d<0x141>(v));         // row_half_mirror
    v = fmax(v, rp_dpp_d<0x140>(v));         // row_mirror
    const double a = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0), __builtin_amdgcn_readlane(__double2loint(v), 0));
    const double b = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16), __builtin_amdgcn_readlane(__double2loint(v), 16));
    const double c = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 32), __builtin_amdgcn_readlane(__double2loint(v), 32));
    const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 48), __builtin_amdgcn_readlane(__double2loint(v), 48));
    return fmax(fmax(a, b), fmax(c, d));
}

// FIXUP: only the rows that affinity_gram_kernel marked (corres_j[row][0] == RP_AFF_REDO: more candidates than its per-lane
// stack holds) are processed; a wave without marked rows exits before staging anything.
#define RP_AFF_REDO (-1)
template <int TP, bool WRITE_WIJ, bool FIXUP = false>       // TP = pairs of target slots per lane (targets <= 128 * TP)
__global__ __launch_bounds__(256) void affinity_rows_kernel(RelposeKeypoints kp, AffConsts ac, int topK, int rows_per_wave,
                                                             float* __restrict__ wij, int32_t* __restrict__ corres_j,
                                                             double* __restrict__ corres_w, int32_t* __restrict__ keff_out) {
    constexpr int T = 2 * TP;
    const int b = blockIdx.y;
    const int ns = kp.ns[b], nt = kp.nt[b];
    const int keff = (ns >= 3 && nt >= 3) ? min(topK, nt - 1) : 0;
    if (!FIXUP && blockIdx.x == 0 && threadIdx.x == 0) keff_out[b] = keff;
    if (keff == 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + wave) * rows_per_wave;
    if (row0 >= ns) return;
    if (FIXUP) {       // one coalesced look at the wave's markers (rows_per_wave <= 64): a wave without marked rows exits before staging anything
        const int r = row0 + lane;
        const bool mark = lane < rows_per_wave && r < ns && corres_j[((size_t)b * kp.ns_max + (r < ns ? r : row0)) * topK] == RP_AFF_REDO;
        if (!__ballot(mark)) return;
    }
    // ---- this lane's targets: descriptors / 100 (float32 division like numpy), observed-weight flags
    rp_v2f ft[TP][RP_FEAT];
    double wt[T];
    const float* ftg = kp.feat_t + (size_t)b * kp.nt_max * RP_FEAT;
#pragma unroll
    for (int p = 0; p < TP; ++p) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int j = (2 * p + h2) * 64 + lane;
            const bool ok = j < nt;
            const float* src = ftg + (size_t)(ok ? j : 0) * RP_FEAT;
            wt[2 * p + h2] = ok ? kp.weight_t[(size_t)b * kp.nt_max + j] : 0.0;
#pragma unroll
            for (int c4 = 0; c4 < RP_FEAT / 4; ++c4) {
                const float4 v = rp_ldg4(src + 4 * c4);
                const float q0 = ok ? v.x / 100.0f : 0.0f, q1 = ok ? v.y / 100.0f : 0.0f, q2 = ok ? v.z / 100.0f : 0.0f, q3 = ok ? v.w / 100.0f : 0.0f;
                if (h2 == 0) { ft[p][4 * c4].x = q0; ft[p][4 * c4 + 1].x = q1; ft[p][4 * c4 + 2].x = q2; ft[p][4 * c4 + 3].x = q3; }
                else { ft[p][4 * c4].y = q0; ft[p][4 * c4 + 1].y = q1; ft[p][4 * c4 + 2].y = q2; ft[p][4 * c4 + 3].y = q3; }
            }
        }
    }
    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const int i = row0 + rr;
        if (i >= ns) break;
        const size_t si = (size_t)b * kp.ns_max + i;
        if (FIXUP && corres_j[si * topK] != RP_AFF_REDO) continue;
        const float fsl = kp.feat_s[si * RP_FEAT + (lane & 31)] / 100.0f;
        const double wsi = kp.weight_s[si];
        float sc[RP_FEAT];                       // the row's descriptor, wave-uniform (SGPRs)
#pragma unroll
        for (int c = 0; c < RP_FEAT; ++c) sc[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fsl), c));
        // ---- numpy-order float32 squared distances (8 strided partial sums + fixed tree), two target slots per packed op
        double e[T];
#pragma unroll
        for (int p = 0; p < TP; ++p) {
            rp_v2f r8[8];
#pragma unroll
            for (int c = 0; c < RP_FEAT; ++c) {
                const rp_v2f sv = {sc[c], sc[c]};
                const rp_v2f df = sv - ft[p][c];
                const rp_v2f sq = df * df;
                if (c < 8) r8[c] = sq; else r8[c & 7] = r8[c & 7] + sq;
            }
            const rp_v2f d2 = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int t = 2 * p + h2;
                const double x = (double)(h2 ? d2.y : d2.x);
                const bool cls = (wsi * wt[t] == 1.0);
                const double den = cls ? ac.den[1] : ac.den[0], rd = cls ? ac.rden[1] : ac.rden[0];
                double q = x * rd;                                    // Markstein: rd = RN(1/den), one correction step = RN(x/den)
                const double rem = __builtin_fma(-q, den, x);
                q = __builtin_fma(rem, rd, q);
                e[t] = (t * 64 + lane < nt) ? -q : -INFINITY;
            }
        }
        // ---- K winners: largest e, ties to the smaller j
        double ek[T];
#pragma unroll
        for (int t = 0; t < T; ++t) ek[t] = e[t];
        double be[RP_MAXK];
        int bj[RP_MAXK];
#pragma unroll
        for (int k = 0; k < RP_MAXK; ++k) {
            be[k] = -INFINITY; bj[k] = INT_MAX;
            if (k < keff) {
                double lb = ek[0];
                int lt = 0;
#pragma unroll
                for (int t = 1; t < T; ++t) if (ek[t] > lb) { lb = ek[t]; lt = t; }
                const double mx = rp_wave_max_d(lb);
                const unsigned long long cand = __ballot(lb == mx);
                const int jl = lt * 64 + lane;
                int owner = __ffsll((long long)cand) - 1;
                int jwin = __builtin_amdgcn_readlane(jl, owner);
                if (cand & (cand - 1)) {                              // several lanes hold the same e: the smallest j wins
                    unsigned long long rest = cand & (cand - 1);
                    while (rest) {
                        const int l2 = __ffsll((long long)rest) - 1;
                        const int j2 = __builtin_amdgcn_readlane(jl, l2);
                        if (j2 < jwin) { jwin = j2; owner = l2; }
                        rest &= rest - 1;
                    }
                }
                be[k] = mx; bj[k] = (mx == -INFINITY) ? INT_MAX : jwin;
                if (lane == owner) {
#pragma unroll
                    for (int t = 0; t < T; ++t) if (t == lt) ek[t] = -INFINITY;
                }
            }
        }
        // ---- exp only where it can matter, row norm
        const double emax = be[0];
        double w[T];
        double sumsq = 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const bool need = e[t] >= emax - RP_AFF_WINDOW;       // false for the padding (-inf)
            w[t] = 0.0;
            if (__ballot(need)) {
                const double v = exp(e[t]);
                w[t] = need ? v : 0.0;
            }
            sumsq += w[t] * w[t];
        }
        const double nm = sqrt(rp_wave_sum(sumsq));
        const double inm = (nm != 0.0) ? 1.0 / nm : 0.0;
        {   // the K outputs: lane k takes winner k (exp + one division per lane, all K in parallel)
            double mybe = -INFINITY;
            int mybj = INT_MAX;
#pragma unroll
            for (int k = 0; k < RP_MAXK; ++k) if (k == lane) { mybe = be[k]; mybj = bj[k]; }
            if (lane < keff) {
                const bool ok = mybj >= 0 && mybj < nt;
                corres_j[si * topK + lane] = ok ? mybj : 0;
                corres_w[si * topK + lane] = (ok && nm != 0.0) ? exp(mybe) / nm : 0.0;
            }
        }
        if (WRITE_WIJ) {
            float* row = wij + si * kp.nt_max;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int j = t * 64 + lane;
                if (j < nt) row[j] = (float)(w[t] * inm);
            }
        }
    }
}


// ---- tile (MFMA) variant: exact work only where it can matter ----------------------------------------------------------------
// The numpy-order float32 distance costs 95 separately rounded operations per entry, but only a handful of entries per row
// matter: the K winners, whatever lies within RP_AFF_NORM_WINDOW of the row maximum (the float64 row norm) and, when wij is
// materialised, within RP_AFF_WINDOW of it (everything else is an exact 0 in the float32 wij).  A wave owns up to 32 source rows
// and finds those entries from APPROXIMATE distances on the matrix pipe: v_mfma_f32_32x32x16_f16 with the TARGETS as M and the
// source rows as N -- in the C layout lane (n, h) holds entries of ITS OWN row n (16 targets per 32-target tile, lanes n and
// n + 32 share a row), so row-wise selection is in-lane work and the approximate values never leave registers.
// With err bounding |e~ - e| (fp16 operand rounding, derived in the kernel) every true winner has e~ >= kth~ - 2 err, where kth~ is
// the K-th largest of the row's group-of-8 maxima (a lower bound of the K-th largest e~), and every entry inside a window W of
// the true maximum has e~ >= max~ - W - 2 err.
// A row whose candidates do not fit the wave's queue / the lane's stack, or whose data the bound does not cover (weights outside
// [0, 1], |descriptor| >= 1e4, NaN), is marked RP_AFF_REDO and redone by affinity_rows_kernel<FIXUP> (launched right behind,
// normally a no-op).  The kernel itself (round 4: second generation) is described at affinity_tile_kernel.
#define AT_WAVES 8             // waves per workgroup = tiles of 32 source rows (small batches launch 2 or 4)
#define AT_LDT 36              // LDS row stride of the target descriptors (floats): conflict-free b128 reads
#define AT_ISTK 16             // important-entry stack per lane (half a row), uint16 target indices; 24 beyond 256 targets
#define RP_AFF_NORM_WINDOW 24.0      // squares below e^-48 (1.4e-21) of the largest are left out of the float64 row norm: 512 of them stay under half an ulp
#define RP_AFF_TILE_MIN_TILES 1024   // 32-row tiles in the batch from which rp_launch_affinity picks the tile kernel

// feat / 100 like numpy: rp_div100_fast (rp_math.h) when every lane's value is in its verified range, the division otherwise
__device__ __forceinline__ float rp_div100(float x) {
    if (__all(rp_div100_ok(x))) return rp_div100_fast(x);
    return x / 100.0f;
}
__device__ __forceinline__ float4 rp_div100(float4 v) { return make_float4(rp_div100(v.x), rp_div100(v.y), rp_div100(v.z), rp_div100(v.w)); }
__device__ __forceinline__ unsigned rp_pkrtz(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ float rp_bperm_f(int byte_idx, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_idx, __float_as_int(v))); }
// LDS traffic between the lanes of ONE wave: LDS operations of a wave execute in issue order, so the lanes only need the compiler to
// keep the order -- wavefront-scope fences.  (A workgroup-scope release fence also waits for the wave's global STORES, s_waitcnt
// vmcnt(0): with those in the write phase every staging chunk waited for the previous chunk's stores to be acknowledged.)
__device__ __forceinline__ void rp_wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int KL>
__device__ __forceinline__ void rp_list_push(float (&te)[KL], float x) {       // sorted (descending) list of the KL largest values
    float prev = te[0];
    te[0] = fmaxf(prev, x);
#pragma unroll
    for (int k = 1; k < KL; ++k) { const float cur = te[k]; te[k] = __builtin_amdgcn_fmed3f(prev, cur, x); prev = cur; }
}
// RN(d / den) by Markstein's correction of d * RN(1/den), negated: the exponent of one entry
__device__ __forceinline__ double rp_exponent(float d, double den, double rd) {
    const double x = (double)d;
    double q = x * rd;
    const double rem = __builtin_fma(-q, den, x);
    q = __builtin_fma(rem, rd, q);
    return -q;
}

// ---- the tile kernel (round 4: second generation) ---------------------------------------------------------------------------
// Reorganised around what the counters of round 3's version showed (profiles/r03_affinity_pmc.txt: 2.1x the algorithmic HBM
// traffic, VALU-issue / latency bound at 1.8 waves per SIMD):
//  1. wij rows are written ONCE.  The first version zero-filled a wave's rows with streaming stores and then overwrote the window
//     entries with 4-byte scatter stores: partial-line read-modify-writes (FETCH 3.4x, WRITE 1.7x the algorithmic bytes).  Here the
//     window values of a wave are gathered first (5 items per lane, in registers), then the wave's rows go through a small LDS
//     staging buffer, a few rows at a time -- zeros, the rows' items scattered over them, coalesced 16-byte stores of finished lines.
//  2. The sweeps work in DISTANCE space.  The targets are staged sorted by their weight class (weight == 1 first), so that the
//     denominator of the exponent -- which takes two values, rpmodule.py:357-359 -- is uniform per 32-target MFMA tile except for the
//     one tile holding the class boundary; -|t|^2/2 enters the product as two extra K slots (fp16 hi + lo) of a third MFMA, so the
//     accumulator IS g = s.t - |t|^2/2 = (|s|^2 - D)/2 and the selection is one v_cmp per entry against a per-class threshold (the first
//     version spent two FMAs per entry and sweep on e~ = B_j (|s|^2 - 2 s.t) + A_j, three sweeps).
//  3. ONE selection sweep and a POOLED exact evaluation.  Winners, norm-window and wij-window entries are nested sets ("e~ >= some
//     threshold"), so one mask per lane covers them all; the candidates of the wave's rows go through a ballot-compacted queue and are
//     evaluated 64 at a time, one per lane whatever row they belong to (exact distance only: 4 bytes per item back to LDS).  The
//     owner lanes then rank their rows' candidates (exponent from the stored distance: 8 instructions) and evaluate exp() for the K
//     winners only; the norm is the winners' share plus -- rarely -- the other candidates inside the norm window.
//     (The first version ran distance + float64 exp + insertion for max-over-lanes candidates per lane: ~60 % of its instructions.)
// A wave owns rpw <= 32 consecutive source rows, rpw chosen so that the waves of a workgroup are equally loaded (200 rows = 8 x 25;
// the first version ran 6 full waves, one quarter-full and one idle).
// Results: corres_j / corres_w identical in meaning to affinity_rows_kernel's (same exact arithmetic on a superset of the entries
// that matter; the float64 norm is added up winners first), wij zeros below e^-75 of the row maximum exactly like the row kernel.
// max(a, b) as med3(a, b, +inf): clang puts a canonicalising v_max x, x in front of every fmaxf operand that it cannot prove quiet
// (MFMA results); v_med3 takes them as they are.  (Inline-asm v_max3 on accumulator registers is not safe: the hazard recognizer
// does not put the MFMA-result wait states in front of inline asm.)
__device__ __forceinline__ float rp_max_nc(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, INFINITY); }
#define AT2_CAPW 448           // candidate items per wave (25 rows x (5 winners + ~6 window entries + slack))
#define AT2_IPL (AT2_CAPW / 64)
#define AT2_PADG (-60000.0f)   // -|t|^2/2 of a padding target: below every threshold, finite in fp16

template <bool WRITE_WIJ, int KL>
__global__ __launch_bounds__(AT_WAVES * 64, 4) void affinity_tile_kernel(RelposeKeypoints kp, AffConsts ac, int topK, int ntp, int rows_per_block, int rpw,
                                                                           float* __restrict__ wij, int32_t* __restrict__ corres_j,
                                                                           double* __restrict__ corres_w, int32_t* __restrict__ keff_out, int istk_cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NT = blockDim.x;
    const int b = blockIdx.y;
    const int ns = kp.ns[b], nt = kp.nt[b];
    const int keff = (ns >= 3 && nt >= 3) ? min(topK, nt - 1) : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) keff_out[b] = keff;
    if (keff == 0) return;
    if (blockIdx.x * rows_per_block >= ns) return;
    float* ftT = (float*)smem;                                   // [ntp][AT_LDT] scaled target descriptors (float32, exact), class-sorted
    unsigned* tpack = (unsigned*)(ftT + (size_t)ntp * AT_LDT);   // [ntp] fp16 {hi, lo * 1024} of -|t|^2 / 2
    unsigned short* perm = (unsigned short*)(tpack + ntp);       // [ntp] sorted position -> target index
    unsigned short* posof = perm + ntp;                          // [ntp] target index -> sorted position (staging only)
    int* misc = (int*)(posof + ntp);                             // [0] max |t|^2 (float bits), [1] pair not covered by the bound, [2] class-1 targets
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, n = lane & 31;
    const int wbytes = AT2_CAPW * 6 + istk_cap * 128;            // per-wave area: queue u16[CAPW] | dist f32[CAPW] | istk u16[cap][64]
    char* warea = (char*)(misc + 4) + (size_t)wave * wbytes;
    float* resd = (float*)warea;                                 // exact distance of item `slot`
    unsigned short* queue = (unsigned short*)(resd + AT2_CAPW);  // item `slot` = (row slot << 9) | sorted target position
    unsigned short* istk = queue + AT2_CAPW;                     // [istk_cap][64] slots of the lane's own candidates
    // this lane's source row: loads issued before the staging
    const int i0w = blockIdx.x * rows_per_block + wave * rpw;
    const int nrows = max(0, min(min(rpw, rows_per_block - wave * rpw), ns - i0w));       // rows of this wave
    const int i = i0w + n;
    const bool rowok = n < nrows;
    const size_t si = (size_t)b * kp.ns_max + (rowok ? i : 0);
    float4 fsraw[RP_FEAT / 4];
#pragma unroll
    for (int c4 = 0; c4 < RP_FEAT / 4; ++c4) fsraw[c4] = rowok ? rp_ldg4(kp.feat_s + si * RP_FEAT + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const double wsi = rowok ? rp_ldg(kp.weight_s + si) : 0.0;
    {   // ---- stage the pair's targets, sorted by weight class
        if (tid == 0) { misc[0] = 0; misc[1] = 0; }
        if (wave == 0) {
            // one wave ranks the targets: class 1 (weight == 1: the "both observed" denominator for rows of weight 1) first, in index order
            double wt[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int j = lane + 64 * k; wt[k] = (j < nt) ? rp_ldg(kp.weight_t + (size_t)b * kp.nt_max + j) : 0.0; }
            int n1 = 0;
            bool bad = false;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = lane + 64 * k;
                n1 += __popcll(__ballot(j < nt && wt[k] == 1.0));
                if (j < nt && !(wt[k] >= 0.0 && wt[k] <= 1.0)) bad = true;
            }
            int r1 = 0, r0 = n1;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = lane + 64 * k;
                const bool v = j < nt, c1 = v && wt[k] == 1.0, c0 = v && !c1;
                const unsigned long long b1 = __ballot(c1), b0 = __ballot(c0);
                const unsigned long long below = (1ull << lane) - 1ull;
                const int p = c1 ? r1 + __popcll(b1 & below) : r0 + __popcll(b0 & below);
                if (v) { posof[j] = (unsigned short)p; perm[p] = (unsigned short)j; }
                r1 += __popcll(b1); r0 += __popcll(b0);
            }
            for (int j = nt + lane; j < ntp; j += 64) perm[j] = (unsigned short)j;       // padding positions
            if (lane == 0) misc[2] = n1;
            if (bad) misc[1] = 1;
        }
        __syncthreads();
        const float* ftg = kp.feat_t + (size_t)b * kp.nt_max * RP_FEAT;
        for (int idx = tid; idx < ntp * (RP_FEAT / 4); idx += NT) {
            const int j = idx >> 3, c4 = idx & 7;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int row = j;
            if (j < nt) { v = rp_ldg4(ftg + (size_t)j * RP_FEAT + 4 * c4); row = posof[j]; }
            *reinterpret_cast<float4*>(&ftT[row * AT_LDT + 4 * c4]) = rp_div100(v);
        }
        __syncthreads();
        float mx = 0.f;
        bool bad = false;
        for (int j = tid; j < ntp; j += NT) {
            float a = 0.f;
            for (int c = 0; c < RP_FEAT; ++c) a += ftT[j * AT_LDT + c] * ftT[j * AT_LDT + c];
            const bool ok = j < nt;
            if (ok && !(a < 1e8f)) bad = true;
            const float g0 = ok ? -0.5f * a : AT2_PADG;
            const _Float16 hi = (_Float16)g0;
            const _Float16 lo = ok ? (_Float16)((g0 - (float)hi) * 1024.0f) : (_Float16)0.0f;
            tpack[j] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
            if (ok) mx = fmaxf(mx, a);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        if (lane == 0) atomicMax(&misc[0], __float_as_int(mx));
        if (bad) misc[1] = 1;
        __syncthreads();
    }
    const float ntmax = __int_as_float(misc[0]);
    const bool pair_bad = misc[1] != 0;
    const int n1 = misc[2];                                                     // sorted positions [0, n1) are the weight-1 targets
    if (nrows <= 0) return;
    // ---- the source row: |s|^2 and the fp16 halves this lane feeds to the MFMAs.  The 32 exact features are NOT kept across the sweeps
    // (with them the sweeps spill into scratch memory inside their loops): the pooled evaluation re-reads the row (L2-resident).
    float nsq = 0.f;
    f16x8 bf0, bf1, bfx;
    {
        float fs[RP_FEAT];
#pragma unroll
        for (int c4 = 0; c4 < RP_FEAT / 4; ++c4) {
            const float4 v = rp_div100(fsraw[c4]);
            fs[4 * c4] = v.x; fs[4 * c4 + 1] = v.y; fs[4 * c4 + 2] = v.z; fs[4 * c4 + 3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < RP_FEAT; ++c) nsq += fs[c] * fs[c];
        // (all 16 packed pairs, then selects between VALUES: a select between fs[] elements becomes a dynamically indexed load and
        // sends the whole array to scratch memory)
        unsigned P[16], p[8];
#pragma unroll
        for (int q = 0; q < 16; ++q) P[q] = rp_pkrtz(fs[2 * q], fs[2 * q + 1]);
#pragma unroll
        for (int q = 0; q < 4; ++q) { p[q] = h ? P[4 + q] : P[q]; p[4 + q] = h ? P[12 + q] : P[8 + q]; }
        const u32x4 lo = {p[0], p[1], p[2], p[3]}, hi = {p[4], p[5], p[6], p[7]};
        bf0 = __builtin_bit_cast(f16x8, lo);
        bf1 = __builtin_bit_cast(f16x8, hi);
        const u32x4 one = {h ? 0u : rp_pkrtz(1.0f, 0.0009765625f), 0u, 0u, 0u};     // K slots 0, 1 of the third MFMA: 1 and 2^-10
        bfx = __builtin_bit_cast(f16x8, one);
    }
    const bool rowone = wsi == 1.0;
    const bool row_bad = pair_bad || !(wsi >= 0.0 && wsi <= 1.0) || !(nsq < 1e8f);
    const float nrd0 = -(float)ac.rden[0], nrd1 = -(float)ac.rden[1];
    // e~ = B D~ with D~ = |s|^2 - 2 g, B = -1/den of the entry's class: e~ = alpha g + beta
    const float B1 = rowone ? nrd1 : nrd0, B0 = nrd0;
    const float al1 = -2.0f * B1, be1 = B1 * nsq, al0 = -2.0f * B0, be0 = B0 * nsq;
    const int ntiles = ntp / 32;                                                // even
    const int tmix = (n1 & 31) ? (n1 >> 5) : -1;                                // the tile that holds the class boundary

    // g of one 32-target tile for this lane's row: acc[r] belongs to sorted position 32 T + 8 (r >> 2) + 4 h + (r & 3)
    auto tile_g = [&](int T) {
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* row = &ftT[(32 * T + n) * AT_LDT + 8 * h];
        const float4 a0 = *reinterpret_cast<const float4*>(row), a1 = *reinterpret_cast<const float4*>(row + 4);
        const float4 a2 = *reinterpret_cast<const float4*>(row + 16), a3 = *reinterpret_cast<const float4*>(row + 20);
        const u32x4 k0 = {rp_pkrtz(a0.x, a0.y), rp_pkrtz(a0.z, a0.w), rp_pkrtz(a1.x, a1.y), rp_pkrtz(a1.z, a1.w)};
        const u32x4 k1 = {rp_pkrtz(a2.x, a2.y), rp_pkrtz(a2.z, a2.w), rp_pkrtz(a3.x, a3.y), rp_pkrtz(a3.z, a3.w)};
        const u32x4 k2 = {h ? 0u : tpack[32 * T + n], 0u, 0u, 0u};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, k0), bf0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, k1), bf1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, k2), bfx, acc, 0, 0, 0);
        return acc;
    };
    auto pos_of = [&](int T, int r) { return 32 * T + 8 * (r >> 2) + 4 * h + (r & 3); };

    // ---- P1: the row maximum and the KL largest group-of-8 maxima of the half row, as exponents (sorted, te[0] = max)
    float te[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) te[k] = -INFINITY;
    for (int T = 0; T < ntiles; ++T) {
        const floatx16 g = tile_g(T);
        float gm[2];
        if (T != tmix) {
            const bool c1 = 32 * T < n1;
            const float al = c1 ? al1 : al0, be = c1 ? be1 : be0;
#pragma unroll
            for (int g8 = 0; g8 < 2; ++g8) {
                const float q0 = rp_max_nc(rp_max_nc(g[8 * g8], g[8 * g8 + 1]), rp_max_nc(g[8 * g8 + 2], g[8 * g8 + 3]));
                const float q1 = rp_max_nc(rp_max_nc(g[8 * g8 + 4], g[8 * g8 + 5]), rp_max_nc(g[8 * g8 + 6], g[8 * g8 + 7]));
                gm[g8] = __builtin_fmaf(al, rp_max_nc(q0, q1), be);
            }
        } else {
#pragma unroll
            for (int g8 = 0; g8 < 2; ++g8) {
                float m8 = -INFINITY;
#pragma unroll
                for (int r = 8 * g8; r < 8 * g8 + 8; ++r) {
                    const bool c1 = pos_of(T, r) < n1;
                    m8 = fmaxf(m8, __builtin_fmaf(c1 ? al1 : al0, g[r], c1 ? be1 : be0));
                }
                gm[g8] = m8;
            }
        }
        rp_list_push<KL>(te, gm[0]);
        rp_list_push<KL>(te, gm[1]);
    }
    {   // merge with the other half of the row (lane ^ 32)
        float ot[KL];
#pragma unroll
        for (int k = 0; k < KL; ++k) ot[k] = __shfl_xor(te[k], 32, 64);
#pragma unroll
        for (int kk = 0; kk < KL; ++kk) rp_list_push<KL>(te, ot[kk]);
    }
    float kth = te[0];
#pragma unroll
    for (int k = 1; k < KL; ++k) if (k == keff - 1) kth = te[k];
    const float emax_a = te[0];
    // |e~ - e|: as in affinity_tile_kernel (fp16 operands rounded toward zero, float32 accumulation), plus the two extra K slots
    // (|t|^2/2 to 2^-21 relative) and the float32 roundings of alpha g + beta and of the thresholds below
    const float rdmax = fmaxf(-nrd0, -nrd1);
    const float errd = 2.2e-3f * (nsq + ntmax) + 3.5e-7f * (sqrtf(nsq) + sqrtf(ntmax));
    const float err = errd * rdmax + 4e-6f * (fmaxf(fabsf(kth), fabsf(emax_a)) + rdmax * nsq) + 1e-30f;
    // important entries (possible winners + norm window: ranked by the row's owner lanes) and, for a materialised wij, the window
    // entries (a value each, nothing else): nested sets, so the sweep tests the wider threshold and a candidate the narrower one
    const float thr_imp = fminf(kth, emax_a - (float)RP_AFF_NORM_WINDOW) - 2.0f * err;
    const float thr = WRITE_WIJ ? fminf(thr_imp, emax_a - (float)RP_AFF_WINDOW - 2.0f * err) : thr_imp;
    const double need_norm = (double)emax_a - RP_AFF_NORM_WINDOW - 3.0 * (double)err;
    // the same thresholds in g: e~ >= thr  <=>  g >= (thr - beta) / alpha (alpha > 0); rows that are not rowok select nothing
    const float tg1 = rowok ? (thr - be1) / al1 - 1e-6f * fabsf((thr - be1) / al1) : INFINITY;
    const float tg0 = rowok ? (thr - be0) / al0 - 1e-6f * fabsf((thr - be0) / al0) : INFINITY;
    const float ti1 = (thr_imp - be1) / al1 - 1e-6f * fabsf((thr_imp - be1) / al1);
    const float ti0 = (thr_imp - be0) / al0 - 1e-6f * fabsf((thr_imp - be0) / al0);

    // candidate masks of the tile pair (T, T + 1) for this lane's row: bit 31 - (16 u + r) <-> sorted position pos_of(T + u, r);
    // m = candidates (e~ >= thr), mi = the important ones among them
    auto pair_masks = [&](int T, unsigned& m, unsigned& mi) {
        m = 0; mi = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const floatx16 g = tile_g(T + u);
            if (T + u != tmix) {
                // g >= t  <=>  the sign bit of g - t is clear (g, t finite; x - y with x == y gives +0): one packed subtraction per two
                // entries and one v_alignbit per entry ({m, d} >> 31 = (m << 1) | sign(d)) -- a v_cmp + v_cndmask + shift / or chain through
                // VCC is 3 instructions and a hazard nop per entry.  m collects the "below" bits.
                const bool c1 = 32 * (T + u) < n1;
                const float tg = c1 ? tg1 : tg0, ti = c1 ? ti1 : ti0;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const rp_v2f d = (rp_v2f){g[r], g[r + 1]} - (rp_v2f){tg, tg};
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(d.x), 31);
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(d.y), 31);
                }
                if (WRITE_WIJ) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const rp_v2f d = (rp_v2f){g[r], g[r + 1]} - (rp_v2f){ti, ti};
                        mi = __builtin_amdgcn_alignbit(mi, __float_as_uint(d.x), 31);
                        mi = __builtin_amdgcn_alignbit(mi, __float_as_uint(d.y), 31);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool c1 = pos_of(T + u, r) < n1;
                    m = (m << 1) | (g[r] >= (c1 ? tg1 : tg0) ? 0u : 1u);
                    if (WRITE_WIJ) mi = (mi << 1) | (g[r] >= (c1 ? ti1 : ti0) ? 0u : 1u);
                }
            }
        }
        m = ~m;                                                                 // candidates = not below
        mi = WRITE_WIJ ? ~mi : m;
    };
    auto load_source_row = [&](float (&fs)[RP_FEAT]) {       // the 32 exact features of this lane's row again (L2-resident)
        long long soff = (long long)(si * RP_FEAT);
        asm volatile("" : "+v"(soff));                                          // (a second load the compiler cannot merge with the first)
#pragma unroll
        for (int c4 = 0; c4 < RP_FEAT / 4; ++c4) {
            const float4 v = rp_div100(rowok ? rp_ldg4(kp.feat_s + soff + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f));
            fs[4 * c4] = v.x; fs[4 * c4 + 1] = v.y; fs[4 * c4 + 2] = v.z; fs[4 * c4 + 3] = v.w;
        }
    };

    // ---- pooled exact distances: one queue item per lane whatever row it belongs to -- the numpy-order float32 distance, the source row
    // gathered from its owner lane 8 features at a time
    auto exact_dist_item = [&](const float (&fs)[RP_FEAT], int item) {
        const int pidx = (item >> 9) << 2;
        const float* tj = &ftT[(item & 511) * AT_LDT];
        float r8[8];
#pragma unroll
        for (int c8 = 0; c8 < RP_FEAT / 8; ++c8) {
            const float4 ta = *reinterpret_cast<const float4*>(tj + 8 * c8), tb = *reinterpret_cast<const float4*>(tj + 8 * c8 + 4);
            const float tt[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float df = rp_bperm_f(pidx, fs[8 * c8 + k]) - tt[k];
                const float sq = df * df;
                if (c8 == 0) r8[k] = sq; else r8[k] = r8[k] + sq;
            }
            __builtin_amdgcn_sched_barrier(0);                                  // (8 gathers in flight, not 32: registers)
        }
        return ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
    };

    auto evaluate = [&](const float (&fs)[RP_FEAT], int from, int count) {          // queue items [from, from + count): distance -> resd
        const bool act = lane < count;
        const float d = exact_dist_item(fs, act ? queue[from + lane] : 0);
        if (act) resd[from + lane] = d;
    };
    // ---- P2: ONE selection sweep; candidates into the wave's queue.  Important candidates (ranked by their row's owner lanes) fill it
    // from the front, window-only ones (a wij value each, nothing else) from the back; when the two meet the window is DENSE (a wide
    // sigmaFeat: a large part of the row lies within e^-75 of its maximum): the window-only items are dropped and the wave writes its
    // rows the other way (below) -- the important ones always have room up to the whole queue.
    int cnt = 0, qimp = 0, qwin = 0;                                            // qimp / qwin: wave-uniform
    bool over = false, dense = false;                                           // dense: wave-uniform
    // exclusive prefix sum over the lanes of a small per-lane count (<= 32) + the wave total, from ballots of its bit planes
    auto lane_prefix = [&](int c, int& total) {
        int ex = 0; total = 0;
#pragma unroll
        for (int bpl = 0; bpl < 6; ++bpl) {
            const unsigned long long bm = __ballot((c >> bpl) & 1);
            if (bm) {
                ex += (__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0))) << bpl;
                total += __popcll(bm) << bpl;
            }
        }
        return ex;
    };
    for (int T = 0; T < ntiles; T += 2) {
        unsigned m, mi;
        pair_masks(T, m, mi);
        // (padding positions >= nt carry g = AT2_PADG: they only pass a threshold of -inf -- rows with fewer than K finite groups -- and are
        // dropped here, before the slots are counted)
        if (32 * (T + 2) > nt) {
            unsigned vm = 0;
#pragma unroll
            for (int idx = 0; idx < 32; ++idx) vm = (vm << 1) | (pos_of(T + (idx >> 4), idx & 15) < nt ? 1u : 0u);
            m &= vm;
        }
        mi &= m;
        // slots: the wave's important candidates of this tile pair take [qimp, qimp + ni) lane by lane, the window-only ones the next nw
        // slots from the back; every lane then writes its own (one ballot per bit plane of the counts instead of two per candidate)
        int ni, nw;
        int si_ = qimp + lane_prefix(__popc(mi), ni);
        int sw_ = lane_prefix(__popc(m & ~mi), nw);
        if (qimp + ni + qwin + (dense ? 0 : nw) > AT2_CAPW) { dense = true; qwin = 0; }
        sw_ += qwin;
        while (m) {
            const int bit = __builtin_ctz(m);
            m &= m - 1;
            const int idx = 31 - bit;
            const int jc = pos_of(T + (idx >> 4), idx & 15);
            if ((mi >> bit) & 1u) {
                if (si_ < AT2_CAPW) queue[si_] = (unsigned short)((n << 9) | jc);
                if (si_ < AT2_CAPW && cnt < istk_cap) istk[cnt * 64 + lane] = (unsigned short)si_; else over = true;
                ++si_; ++cnt;
            } else if (!dense) {
                queue[AT2_CAPW - 1 - sw_] = (unsigned short)((n << 9) | jc);
                ++sw_;
            }
        }
        qimp = min(qimp + ni, AT2_CAPW);
        if (!dense) qwin += nw;
    }
    rp_wave_lds_order();
    {   // the exact distances of the queued candidates, 64 at a time, one per lane whatever row they belong to
        float fs[RP_FEAT];
        load_source_row(fs);
        for (int from = 0; from < qimp; from += 64) evaluate(fs, from, min(64, qimp - from));
        for (int from = AT2_CAPW - qwin; from < AT2_CAPW; from += 64) evaluate(fs, from, min(64, AT2_CAPW - from));
    }
    rp_wave_lds_order();
    const int ncand = over ? 0 : cnt;

    // ---- the owner lanes rank their candidates: sorted list of the KL best by (e descending, target index ascending)
    double le[KL];
    int lj[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) { le[k] = -INFINITY; lj[k] = INT_MAX; }
    int nnorm = 0;                                                              // own candidates inside the norm window
    for (int c = 0; __ballot(c < ncand); ++c) {
        const bool v = c < ncand;
        const int slot = v ? istk[c * 64 + lane] : 0;
        const int jp = queue[slot] & 511;
        const bool cls = rowone && jp < n1;
        double e = rp_exponent(resd[slot], cls ? ac.den[1] : ac.den[0], cls ? ac.rden[1] : ac.rden[0]);
        int jj = v ? (int)perm[jp] : INT_MAX;
        if (!v) e = -INFINITY;
        nnorm += (e >= need_norm) ? 1 : 0;
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            const bool better = (e > le[k]) || (e == le[k] && jj < lj[k]);
            const double te_ = le[k]; const int tj_ = lj[k];
            le[k] = better ? e : te_; lj[k] = better ? jj : tj_;
            e = better ? te_ : e; jj = better ? tj_ : jj;
        }
    }
    {   // merge the two halves of the row
        double oe[KL]; int oj[KL];
#pragma unroll
        for (int k = 0; k < KL; ++k) { oe[k] = rp_shfl_xor_d(le[k], 32); oj[k] = __shfl_xor(lj[k], 32, 64); }
#pragma unroll
        for (int kk = 0; kk < KL; ++kk) {
            double e = oe[kk]; int jj = oj[kk];
#pragma unroll
            for (int k = 0; k < KL; ++k) {
                const bool better = (e > le[k]) || (e == le[k] && jj < lj[k]);
                const double te_ = le[k]; const int tj_ = lj[k];
                le[k] = better ? e : te_; lj[k] = better ? jj : tj_;
                e = better ? te_ : e; jj = better ? tj_ : jj;
            }
        }
    }
    // exp() of the winners; the norm = the winners inside the norm window, in rank order ...
    // (both lanes of a row hold the same list: half 0 evaluates the even ranks, half 1 the odd ones, then they swap)
    double lw[KL];
    double sumsq = 0.0;
    int inlist = 0;
    {
        constexpr int KH = (KL + 1) / 2;
        double wh[KH];
#pragma unroll
        for (int q = 0; q < KH; ++q) {
            const int k0 = 2 * q, k1 = (2 * q + 1 < KL) ? 2 * q + 1 : 2 * q;
            wh[q] = exp(h ? le[k1] : le[k0]);                                   // (exp(-inf) = 0 for an empty entry)
        }
#pragma unroll
        for (int q = 0; q < KH; ++q) {
            const double wo = rp_shfl_xor_d(wh[q], 32);
            lw[2 * q] = h ? wo : wh[q];
            if (2 * q + 1 < KL) lw[2 * q + 1] = h ? wh[q] : wo;
        }
    }
#pragma unroll
    for (int k = 0; k < KL; ++k)
        if (le[k] >= need_norm) { sumsq += lw[k] * lw[k]; ++inlist; }
    // ... plus, rarely, candidates inside the norm window that are not among the KL best: half 0's first, then half 1's, in queue order
    const int nrow_norm = nnorm + __shfl_xor(nnorm, 32, 64);
    if (__ballot(nrow_norm > inlist)) {
        double extra = 0.0;
        const double elast = le[KL - 1]; const int jlast = lj[KL - 1];
        for (int c = 0; __ballot(c < ncand && nrow_norm > inlist); ++c) {
            const bool v = c < ncand && nrow_norm > inlist;
            const int slot = v ? istk[c * 64 + lane] : 0;
            const int jp = queue[slot] & 511;
            const bool cls = rowone && jp < n1;
            const double e = rp_exponent(resd[slot], cls ? ac.den[1] : ac.den[0], cls ? ac.rden[1] : ac.rden[0]);
            const int jj = (int)perm[jp];
            const bool want = v && e >= need_norm && ((e < elast) || (e == elast && jj > jlast));
            if (__ballot(want)) { const double w = exp(e); if (want) extra += w * w; }
        }
        const double xo = rp_shfl_xor_d(extra, 32);
        sumsq += (h == 0) ? extra + xo : xo + extra;
    }
    const double nm = sqrt(sumsq);
    const int partner_over = __shfl_xor((int)over, 32, 64);
    const bool redo = over || partner_over != 0 || row_bad;
    if (rowok) {       // half h writes the ranks k with k & 1 == h (the float64 divisions are the cost here)
        if (redo) { if (h == 0) corres_j[si * topK] = RP_AFF_REDO; }
        else {
#pragma unroll
            for (int q = 0; q < (KL + 1) / 2; ++q) {
                const int k = 2 * q + h;
                const int jk = (2 * q + 1 < KL) ? (h ? lj[2 * q + 1] : lj[2 * q]) : lj[2 * q];
                const double wk = (2 * q + 1 < KL) ? (h ? lw[2 * q + 1] : lw[2 * q]) : lw[2 * q];
                if (k < keff && k < KL) {
                    const bool ok = jk >= 0 && jk < nt;
                    corres_j[si * topK + k] = ok ? jk : 0;
                    corres_w[si * topK + k] = (ok && nm != 0.0) ? wk / nm : 0.0;
                }
            }
        }
    }
    if (!WRITE_WIJ) return;

    // ---- window values: wij = exp(e - e_max) * (exp(e_max) / norm) of one queued / re-found candidate, evaluated by whichever lane holds it
    const float rsf = (redo || !(nm > 0.0)) ? 0.f : (float)(lw[0] / nm);
    const double emx = le[0];
    auto window_value = [&](int rown, int jp, float d) {
        const int pidx = rown << 2;
        const float rs_i = rp_bperm_f(pidx, rsf);
        const int one_i = __builtin_amdgcn_ds_bpermute(pidx, rowone ? 1 : 0);
        const double em_i = __hiloint2double(__builtin_amdgcn_ds_bpermute(pidx, __double2hiint(emx)), __builtin_amdgcn_ds_bpermute(pidx, __double2loint(emx)));
        const bool cls = one_i && jp < n1;
        const double x = rp_exponent(d, cls ? ac.den[1] : ac.den[0], cls ? ac.rden[1] : ac.rden[0]) - em_i;
        const double tl = x * 1.4426950408889634;
        const double nr = __builtin_rint(tl);
        const float p2 = __builtin_amdgcn_exp2f((float)(tl - nr));
        float val = ldexpf(p2 * rs_i, (int)fmax(nr, -300.0));
        if (!(x >= -RP_AFF_WINDOW && x <= 0.0)) val = 0.f;                      // exact zeros below the window, like affinity_rows_kernel (and for rows without a list)
        return val;
    };
    float* wbase = wij + ((size_t)b * kp.ns_max + i0w) * kp.nt_max;
    const int ntm = kp.nt_max;
    const bool vec4 = (ntm & 3) == 0;
    if (!dense) {
        // sparse window (the usual case): item idx = lane + 64 q of the queue's two ends -> (row, target, value) in registers, then the
        // wave's rows a few at a time through LDS: zeros, the rows' items over them, coalesced stores of the finished lines -- every
        // 128-byte line of wij is stored exactly once
        const int nitem = qimp + qwin;
        int it_rj[AT2_IPL];
        float it_v[AT2_IPL];
#pragma unroll
        for (int q = 0; q < AT2_IPL; ++q) {
            const int idx = lane + 64 * q;
            const bool act = idx < nitem;
            const int slot = act ? (idx < qimp ? idx : AT2_CAPW - 1 - (idx - qimp)) : 0;
            const int item = queue[slot];
            const float val = window_value(item >> 9, item & 511, act ? resd[slot] : 0.f);
            it_rj[q] = act ? (((item >> 9) << 9) | (int)perm[item & 511]) : -1;
            it_v[q] = val;
        }
        rp_wave_lds_order();                                                     // every lane has its items: the wave's LDS area becomes the staging buffer
        float* stage = (float*)warea;
        const int crow = max(1, min(nrows, (wbytes / 4) / ntm));                // rows per chunk
        for (int r0 = 0; r0 < nrows; r0 += crow) {
            const int rows = min(crow, nrows - r0), nfl = rows * ntm;
            if (vec4) for (int q4 = lane; q4 < (nfl >> 2); q4 += 64) *reinterpret_cast<float4*>(stage + 4 * q4) = make_float4(0.f, 0.f, 0.f, 0.f);
            else for (int q1 = lane; q1 < nfl; q1 += 64) stage[q1] = 0.f;
            rp_wave_lds_order();
#pragma unroll
            for (int q = 0; q < AT2_IPL; ++q) {
                const int rr = (it_rj[q] >> 9) - r0;
                if (it_rj[q] >= 0 && rr >= 0 && rr < rows) stage[rr * ntm + (it_rj[q] & 511)] = it_v[q];
            }
            rp_wave_lds_order();
            float* dst = wbase + (size_t)r0 * ntm;
            if (vec4) for (int q4 = lane; q4 < (nfl >> 2); q4 += 64) rp_stg4(dst + 4 * q4, *reinterpret_cast<const float4*>(stage + 4 * q4));
            else for (int q1 = lane; q1 < nfl; q1 += 64) rp_stg(dst + q1, stage[q1]);
            rp_wave_lds_order();
        }
        return;
    }
    // ---- dense window: more window entries than the queue holds.  The rows are zero-filled with coalesced stores, then the targets are
    // swept once more and every candidate goes through a ring of 128, 64 at a time: exact distance, value, a 4-byte store over the zero
    // (round 3's write path: partial-line stores, but a dense window means most lines are touched anyway)
    {
        const int nfl = nrows * ntm;
        if (vec4) for (int q4 = lane; q4 < (nfl >> 2); q4 += 64) rp_stg4(wbase + 4 * q4, make_float4(0.f, 0.f, 0.f, 0.f));
        else for (int q1 = lane; q1 < nfl; q1 += 64) rp_stg(wbase + q1, 0.f);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the zeros of this wave have landed: same-address order
        float fs[RP_FEAT];
        load_source_row(fs);
        int qhead = 0, qn = 0;                                                  // ring state (wave-uniform); slots [0, 128) of the queue
        auto flush = [&](int count) {
            rp_wave_lds_order();
            const bool act = lane < count;
            const int item = act ? queue[(qhead + lane) & 127] : 0;
            const float d = exact_dist_item(fs, item);
            const float val = window_value(item >> 9, item & 511, d);
            if (act) rp_stg(wbase + (size_t)(item >> 9) * ntm + perm[item & 511], val);
            qhead = (qhead + 64) & 127;
        };
        for (int T = 0; T < ntiles; T += 2) {
            unsigned m, mi;
            pair_masks(T, m, mi);
            if (!rowok || redo) m = 0;
            while (__ballot(m != 0)) {
                int jc = INT_MAX;
                if (m) { const int bit = __builtin_ctz(m); m &= m - 1; const int idx = 31 - bit; jc = pos_of(T + (idx >> 4), idx & 15); }
                const bool push = jc < nt;
                const unsigned long long bm = __ballot(push);
                const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
                if (push) queue[(qhead + qn + pos) & 127] = (unsigned short)((n << 9) | jc);
                qn += __popcll(bm);
                if (qn >= 64) { flush(64); qn -= 64; }
            }
        }
        if (qn) flush(qn);
    }
}

// ---- pool variant (round 5): the tile kernel's stages as SEPARATE dense launches ----------------------------------------------------------
// The tile kernel runs five dependent stages per workgroup (staging | P1 | selection + pooled exact distances | ranking | exp + row
// stores) at 77 KB of LDS and 118 VGPRs: 2 workgroups per CU, every stage a latency chain (profiles/r04_affinity_pmc.txt).  Here each stage
// is a launch of its own, shaped for what it does:
//   aff3_screen_kernel  the SAME approximate screen (fp16 MFMA, the same error bound and thresholds as affinity_tile_kernel) with the targets
//                       staged as fp16 only (20 KB of LDS); the candidates of every half row go into a global pool as ONE contiguous
//                       segment per lane (wave-level reservation with one atomic), the scaled float32 descriptors are written out once,
//                       and -- for a materialised wij -- the wave zero-fills its rows with streaming stores while the matrix pipe works;
//   aff3_exact_kernel   one lane per pooled candidate: numpy-order float32 distance, Markstein exponent (dense, no queues or ballots);
//   aff3_rank_kernel    one lane per source ROW (64 different rows per wave): rank, exp of the K winners, float64 norm over the norm
//                       window, outputs, and the window values of the row scattered over the zeros;
// rows the bound does not cover / that do not fit the pool are marked RP_AFF_REDO and redone by affinity_rows_kernel<FIXUP> as before.
// Results: corres_j identical, corres_w / wij to round-off of the tile kernel's (the norm window is taken from the exact row maximum).
struct A3Scratch {
    unsigned* ctl;             // (unused)
    uint2* hdr;                // [waves] {pool base, candidates} of every screen wave
    unsigned short* seg;       // [waves][64] exclusive prefix of the lanes' candidate counts
    float* fs_sc;              // [B * ns_max * 32] feat_s / 100 (float32 division like numpy)
    float* ft_sc;              // [B * nt_max * 32]
    unsigned* item_row;        // [waves][cap] global source row of candidate g
    unsigned* item_tgt;        // [waves][cap] (global target row << 1) | denominator class
    double* item_e;            // [waves][cap] exact exponent
    unsigned cap;              // pool slots per screen wave (a multiple of 64)
    unsigned nwaves;
};
#define A3_LDB 80              // LDS row stride of the fp16 target descriptors (bytes): conflict-free b128 reads

// exclusive prefix over the lanes of a per-lane count < 512 + the wave total (ballots of the bit planes)
__device__ __forceinline__ int a3_lane_prefix(int c, int& total) {
    int ex = 0; total = 0;
#pragma unroll
    for (int bpl = 0; bpl < 9; ++bpl) {
        const unsigned long long bm = __ballot((c >> bpl) & 1);
        if (bm) {
            ex += (__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0))) << bpl;
            total += __popcll(bm) << bpl;
        }
    }
    return ex;
}

template <bool WRITE_WIJ, int KL, int TPM>      // TPM: tile pairs of 64 targets (ntp <= 128 * TPM... ntp = 64 * tile pairs <= 64 * TPM)
__global__ __launch_bounds__(AT_WAVES * 64, 4) void aff3_screen_kernel(RelposeKeypoints kp, AffConsts ac, int topK, int ntp, int rows_per_block, int rpw,
                                                                         float* __restrict__ wij, int32_t* __restrict__ corres_j,
                                                                         int32_t* __restrict__ keff_out, A3Scratch sc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NT = blockDim.x;
    const int b = blockIdx.y;
    const int ns = kp.ns[b], nt = kp.nt[b];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, n = lane & 31;
    const int wv = (blockIdx.y * gridDim.x + blockIdx.x) * (NT >> 6) + wave;       // this wave's header
    const int keff = (ns >= 3 && nt >= 3) ? min(topK, nt - 1) : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) keff_out[b] = keff;
    if (keff == 0 || blockIdx.x * rows_per_block >= ns) {
        if (lane == 0) sc.hdr[wv] = make_uint2(0u, 0u);
        return;
    }
    char* th = smem;                                             // [ntp][A3_LDB] fp16 target descriptors, class-sorted
    unsigned* tpack = (unsigned*)(th + (size_t)ntp * A3_LDB);    // [ntp] fp16 {hi, lo * 1024} of -|t|^2 / 2
    unsigned short* perm = (unsigned short*)(tpack + ntp);       // [ntp] sorted position -> target index
    unsigned short* posof = perm + ntp;                          // [ntp] target index -> sorted position
    int* misc = (int*)(posof + ntp);                             // [0] max |t|^2, [1] pair not covered by the bound, [2] class-1 targets
    const int i0w = blockIdx.x * rows_per_block + wave * rpw;
    const int nrows = max(0, min(min(rpw, rows_per_block - wave * rpw), ns - i0w));
    const int i = i0w + n;
    const bool rowok = n < nrows;
    const size_t si = (size_t)b * kp.ns_max + (rowok ? i : 0);
    float4 fsraw[RP_FEAT / 4];
#pragma unroll
    for (int c4 = 0; c4 < RP_FEAT / 4; ++c4) fsraw[c4] = rowok ? rp_ldg4(kp.feat_s + si * RP_FEAT + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const double wsi = rowok ? rp_ldg(kp.weight_s + si) : 0.0;
    if (WRITE_WIJ && nrows > 0) {
        // the wave's rows of wij: zeros now (streaming stores that drain under the sweeps), the window values over them later (aff3_rank_kernel)
        float* wbase = wij + ((size_t)b * kp.ns_max + i0w) * kp.nt_max;
        const int nfl = nrows * kp.nt_max;
        if ((kp.nt_max & 3) == 0) for (int q4 = lane; q4 < (nfl >> 2); q4 += 64) rp_stg4(wbase + 4 * q4, make_float4(0.f, 0.f, 0.f, 0.f));
        else for (int q1 = lane; q1 < nfl; q1 += 64) rp_stg(wbase + q1, 0.f);
    }
    {   // ---- stage the pair's targets, sorted by weight class, as fp16 (rounded toward zero like the tile kernel's operands)
        if (tid == 0) { misc[0] = 0; misc[1] = 0; }
        if (wave == 0) {
            double wt[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int j = lane + 64 * k; wt[k] = (j < nt) ? rp_ldg(kp.weight_t + (size_t)b * kp.nt_max + j) : 0.0; }
            int n1 = 0;
            bool bad = false;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = lane + 64 * k;
                n1 += __popcll(__ballot(j < nt && wt[k] == 1.0));
                if (j < nt && !(wt[k] >= 0.0 && wt[k] <= 1.0)) bad = true;
            }
            int r1 = 0, r0 = n1;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = lane + 64 * k;
                const bool v = j < nt, c1 = v && wt[k] == 1.0, c0 = v && !c1;
                const unsigned long long b1 = __ballot(c1), b0 = __ballot(c0);
                const unsigned long long below = (1ull << lane) - 1ull;
                const int p = c1 ? r1 + __popcll(b1 & below) : r0 + __popcll(b0 & below);
                if (v) { posof[j] = (unsigned short)p; perm[p] = (unsigned short)j; }
                r1 += __popcll(b1); r0 += __popcll(b0);
            }
            for (int j = nt + lane; j < ntp; j += 64) perm[j] = (unsigned short)j;
            if (lane == 0) misc[2] = n1;
            if (bad) misc[1] = 1;
        }
        __syncthreads();
        const float* ftg = kp.feat_t + (size_t)b * kp.nt_max * RP_FEAT;
        float* fto = sc.ft_sc + (size_t)b * kp.nt_max * RP_FEAT;
        float mx = 0.f;
        bool bad = false;
        for (int idx0 = 0; idx0 < ntp * (RP_FEAT / 4); idx0 += NT) {       // (uniform trip count: the lanes of a target reduce through DPP)
            const int idx = idx0 + tid;
            const int j = idx >> 3, c4 = idx & 7;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int row = min(j, ntp - 1);
            const bool ok = j < nt;
            if (ok) { v = rp_ldg4(ftg + (size_t)j * RP_FEAT + 4 * c4); row = posof[j]; }
            v = rp_div100(v);
            if (ok && blockIdx.x == 0) rp_stg4(fto + (size_t)j * RP_FEAT + 4 * c4, v);      // the scaled descriptors, once per pair
            if (j < ntp) *reinterpret_cast<uint2*>(th + (size_t)row * A3_LDB + 8 * c4) = make_uint2(rp_pkrtz(v.x, v.y), rp_pkrtz(v.z, v.w));
            // |t|^2 over the 8 lanes that hold the target's 32 features (a fixed tree; the bound below covers the summation order)
            float a = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            a += __shfl_xor(a, 1, 64);
            a += __shfl_xor(a, 2, 64);
            a += __shfl_xor(a, 4, 64);
            if (c4 == 0 && j < ntp) {
                if (ok && !(a < 1e8f)) bad = true;
                const float g0 = ok ? -0.5f * a : AT2_PADG;
                const _Float16 hi = (_Float16)g0;
                const _Float16 lo = ok ? (_Float16)((g0 - (float)hi) * 1024.0f) : (_Float16)0.0f;
                tpack[row] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
                if (ok) mx = fmaxf(mx, a);
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        if (lane == 0) atomicMax(&misc[0], __float_as_int(mx));
        if (bad) misc[1] = 1;
        __syncthreads();
    }
    const float ntmax = __int_as_float(misc[0]);
    const bool pair_bad = misc[1] != 0;
    const int n1 = misc[2];
    if (nrows <= 0) {
        if (lane == 0) sc.hdr[wv] = make_uint2(0u, 0u);
        return;
    }
    float nsq = 0.f;
    f16x8 bf0, bf1, bfx;
    {
        float fs[RP_FEAT];
#pragma unroll
        for (int c4 = 0; c4 < RP_FEAT / 4; ++c4) {
            const float4 v = rp_div100(fsraw[c4]);
            fs[4 * c4] = v.x; fs[4 * c4 + 1] = v.y; fs[4 * c4 + 2] = v.z; fs[4 * c4 + 3] = v.w;
            if (rowok && h == 0) rp_stg4(sc.fs_sc + si * RP_FEAT + 4 * c4, v);
        }
#pragma unroll
        for (int c = 0; c < RP_FEAT; ++c) nsq += fs[c] * fs[c];
        unsigned P[16], p[8];
#pragma unroll
        for (int q = 0; q < 16; ++q) P[q] = rp_pkrtz(fs[2 * q], fs[2 * q + 1]);
#pragma unroll
        for (int q = 0; q < 4; ++q) { p[q] = h ? P[4 + q] : P[q]; p[4 + q] = h ? P[12 + q] : P[8 + q]; }
        const u32x4 lo = {p[0], p[1], p[2], p[3]}, hi = {p[4], p[5], p[6], p[7]};
        bf0 = __builtin_bit_cast(f16x8, lo);
        bf1 = __builtin_bit_cast(f16x8, hi);
        const u32x4 one = {h ? 0u : rp_pkrtz(1.0f, 0.0009765625f), 0u, 0u, 0u};
        bfx = __builtin_bit_cast(f16x8, one);
    }
    const bool rowone = wsi == 1.0;
    const bool row_bad = pair_bad || !(wsi >= 0.0 && wsi <= 1.0) || !(nsq < 1e8f);
    const float nrd0 = -(float)ac.rden[0], nrd1 = -(float)ac.rden[1];
    const float B1 = rowone ? nrd1 : nrd0, B0 = nrd0;
    const float al1 = -2.0f * B1, be1 = B1 * nsq, al0 = -2.0f * B0, be0 = B0 * nsq;
    const int ntiles = ntp / 32;
    const int tmix = (n1 & 31) ? (n1 >> 5) : -1;

    auto tile_g = [&](int T) {
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const char* row = th + (size_t)(32 * T + n) * A3_LDB + 16 * h;
        const u32x4 k0 = *reinterpret_cast<const u32x4*>(row), k1 = *reinterpret_cast<const u32x4*>(row + 32);
        const u32x4 k2 = {h ? 0u : tpack[32 * T + n], 0u, 0u, 0u};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, k0), bf0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, k1), bf1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, k2), bfx, acc, 0, 0, 0);
        return acc;
    };
    auto pos_of = [&](int T, int r) { return 32 * T + 8 * (r >> 2) + 4 * h + (r & 3); };

    // ---- P1 (as affinity_tile_kernel): row maximum and the KL largest group-of-8 maxima, as exponents
    float te[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) te[k] = -INFINITY;
    for (int T = 0; T < ntiles; ++T) {
        const floatx16 g = tile_g(T);
        float gm[2];
        if (T != tmix) {
            const bool c1 = 32 * T < n1;
            const float al = c1 ? al1 : al0, be = c1 ? be1 : be0;
#pragma unroll
            for (int g8 = 0; g8 < 2; ++g8) {
                const float q0 = rp_max_nc(rp_max_nc(g[8 * g8], g[8 * g8 + 1]), rp_max_nc(g[8 * g8 + 2], g[8 * g8 + 3]));
                const float q1 = rp_max_nc(rp_max_nc(g[8 * g8 + 4], g[8 * g8 + 5]), rp_max_nc(g[8 * g8 + 6], g[8 * g8 + 7]));
                gm[g8] = __builtin_fmaf(al, rp_max_nc(q0, q1), be);
            }
        } else {
#pragma unroll
            for (int g8 = 0; g8 < 2; ++g8) {
                float m8 = -INFINITY;
#pragma unroll
                for (int r = 8 * g8; r < 8 * g8 + 8; ++r) {
                    const bool c1 = pos_of(T, r) < n1;
                    m8 = fmaxf(m8, __builtin_fmaf(c1 ? al1 : al0, g[r], c1 ? be1 : be0));
                }
                gm[g8] = m8;
            }
        }
        rp_list_push<KL>(te, gm[0]);
        rp_list_push<KL>(te, gm[1]);
    }
    {
        float ot[KL];
#pragma unroll
        for (int k = 0; k < KL; ++k) ot[k] = __shfl_xor(te[k], 32, 64);
#pragma unroll
        for (int kk = 0; kk < KL; ++kk) rp_list_push<KL>(te, ot[kk]);
    }
    float kth = te[0];
#pragma unroll
    for (int k = 1; k < KL; ++k) if (k == keff - 1) kth = te[k];
    const float emax_a = te[0];
    const float rdmax = fmaxf(-nrd0, -nrd1);
    const float errd = 2.2e-3f * (nsq + ntmax) + 3.5e-7f * (sqrtf(nsq) + sqrtf(ntmax));
    const float err = errd * rdmax + 4e-6f * (fmaxf(fabsf(kth), fabsf(emax_a)) + rdmax * nsq) + 1e-30f;
    // one threshold: possible winners, the norm window and (materialised wij) the value window are nested sets
    const float thr_imp = fminf(kth, emax_a - (float)RP_AFF_NORM_WINDOW) - 2.0f * err;
    const float thr = WRITE_WIJ ? fminf(thr_imp, emax_a - (float)RP_AFF_WINDOW - 2.0f * err) : thr_imp;
    const bool sel_row = rowok && !row_bad;
    const float tg1 = sel_row ? (thr - be1) / al1 - 1e-6f * fabsf((thr - be1) / al1) : INFINITY;
    const float tg0 = sel_row ? (thr - be0) / al0 - 1e-6f * fabsf((thr - be0) / al0) : INFINITY;

    // ---- P2: the candidate masks of the half row, one word per pair of tiles: bit 31 - (16 u + r) <-> sorted position pos_of(T + u, r)
    unsigned m[TPM];
    int cnt = 0;
#pragma unroll
    for (int tp = 0; tp < TPM; ++tp) {
        m[tp] = 0;
        if (2 * tp < ntiles) {
            const int T = 2 * tp;
            unsigned mm = 0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const floatx16 g = tile_g(T + u);
                if (T + u != tmix) {
                    const bool c1 = 32 * (T + u) < n1;
                    const float tg = c1 ? tg1 : tg0;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const rp_v2f d = (rp_v2f){g[r], g[r + 1]} - (rp_v2f){tg, tg};
                        mm = __builtin_amdgcn_alignbit(mm, __float_as_uint(d.x), 31);
                        mm = __builtin_amdgcn_alignbit(mm, __float_as_uint(d.y), 31);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool c1 = pos_of(T + u, r) < n1;
                        mm = (mm << 1) | (g[r] >= (c1 ? tg1 : tg0) ? 0u : 1u);
                    }
                }
            }
            mm = ~mm;
            if (32 * (T + 2) > nt) {       // padding positions
                unsigned vm = 0;
#pragma unroll
                for (int idx = 0; idx < 32; ++idx) vm = (vm << 1) | (pos_of(T + (idx >> 4), idx & 15) < nt ? 1u : 0u);
                mm &= vm;
            }
            if (!sel_row) mm = 0;
            m[tp] = mm;
            cnt += __popc(mm);
        }
    }
    // ---- one contiguous pool segment per lane: wave total, one reservation, every lane writes its own candidates
    int total;
    const int ex = a3_lane_prefix(cnt, total);
    // (a fixed slice of the pool per screen wave: one global fill counter bumped by 8192 waves from 8 XCDs cost ~100 us of line migrations)
    const unsigned base = (unsigned)wv * sc.cap;
    const bool over = (unsigned)total > sc.cap;                                 // the slice is full: the wave's rows go to the exact redo
    if (lane == 0) sc.hdr[wv] = make_uint2(base, over ? 0u : (unsigned)total);
    sc.seg[(size_t)wv * 64 + lane] = (unsigned short)ex;
    if (rowok && h == 0) corres_j[si * topK] = (row_bad || over) ? RP_AFF_REDO : 0;
    if (over || total == 0) return;
    unsigned pslot = base + (unsigned)ex;
    const unsigned grow = (unsigned)si;
    const unsigned tbase = (unsigned)((size_t)b * kp.nt_max);
#pragma unroll
    for (int tp = 0; tp < TPM; ++tp) {
        unsigned mm = m[tp];
        while (mm) {
            const int bit = __builtin_ctz(mm);
            mm &= mm - 1;
            const int idx = 31 - bit;
            const int pos = pos_of(2 * tp + (idx >> 4), idx & 15);
            const unsigned j = perm[pos];
            const unsigned cls = (rowone && pos < n1) ? 1u : 0u;
            sc.item_row[pslot] = grow;
            sc.item_tgt[pslot] = ((tbase + j) << 1) | cls;
            ++pslot;
        }
    }
}

// one lane per pooled candidate: the numpy-order float32 distance of its (source row, target row) and the exponent
__global__ __launch_bounds__(256) void aff3_exact_kernel(A3Scratch sc, AffConsts ac) {
    // a wave <-> 64 consecutive slots of one screen wave's slice
    const unsigned chunks = sc.cap >> 6;
    const unsigned gw = blockIdx.x * 4u + (threadIdx.x >> 6);
    const unsigned wv = gw / chunks, chunk = gw - wv * chunks;
    if (wv >= sc.nwaves) return;
    const unsigned cnt_w = sc.hdr[wv].y;
    if (chunk * 64u >= cnt_w) return;
    {
        const unsigned slot = chunk * 64u + (threadIdx.x & 63u);
        const unsigned g = wv * sc.cap + slot;
        const bool act = slot < cnt_w;
        const unsigned row = act ? sc.item_row[g] : 0u, tg = act ? sc.item_tgt[g] : 0u;
        const float* ps = sc.fs_sc + (size_t)row * RP_FEAT;
        const float* pt = sc.ft_sc + (size_t)(tg >> 1) * RP_FEAT;
        float r8[8];
#pragma unroll
        for (int c8 = 0; c8 < RP_FEAT / 8; ++c8) {
            const float4 sa = rp_ldg4(ps + 8 * c8), sb = rp_ldg4(ps + 8 * c8 + 4);
            const float4 ta = rp_ldg4(pt + 8 * c8), tb = rp_ldg4(pt + 8 * c8 + 4);
            const float ss[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
            const float tt[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float df = ss[k] - tt[k];
                const float sq = df * df;
                if (c8 == 0) r8[k] = sq; else r8[k] = r8[k] + sq;
            }
        }
        const float d = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
        const bool cls = (tg & 1u) != 0;
        if (act) sc.item_e[g] = rp_exponent(d, cls ? ac.den[1] : ac.den[0], cls ? ac.rden[1] : ac.rden[0]);
    }
}

// one lane per source row: rank the row's candidates (e descending, target index ascending), exp of the K winners, the float64 norm over
// the norm window (winners in rank order, then the other candidates inside it in pool order), the outputs, the row's window values
template <bool WRITE_WIJ, int KL>
__global__ __launch_bounds__(256) void aff3_rank_kernel(RelposeKeypoints kp, int topK, int rows_per_block, int rpw, int nblk, int atw,
                                                         float* __restrict__ wij, int32_t* __restrict__ corres_j, double* __restrict__ corres_w,
                                                         A3Scratch sc) {
    const long long grow = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nrow_all = (long long)kp.B * kp.ns_max;
    const bool inb = grow < nrow_all;
    const int b = inb ? (int)(grow / kp.ns_max) : 0;
    const int r = inb ? (int)(grow - (long long)b * kp.ns_max) : 0;
    const int ns = kp.ns[b], nt = kp.nt[b];
    const int keff = (ns >= 3 && nt >= 3) ? min(topK, nt - 1) : 0;
    bool live = inb && keff > 0 && r < ns;
    if (live && corres_j[(size_t)grow * topK] == RP_AFF_REDO) live = false;
    // the screen wave that owns this row, and the row's two pool segments (its half-row lanes n and n + 32)
    const int blk = r / rows_per_block, rin = r - blk * rows_per_block;
    const int w = rin / rpw, n = rin - w * rpw;
    const size_t wv = ((size_t)b * nblk + blk) * atw + w;
    uint2 hd = make_uint2(0u, 0u);
    unsigned o0 = 0, c0 = 0, o1 = 0, c1 = 0;
    if (live) {
        hd = sc.hdr[wv];
        const unsigned short* sg = sc.seg + wv * 64;
        o0 = sg[n]; c0 = sg[n + 1] - o0;
        o1 = sg[n + 32]; c1 = (n + 32 == 63 ? hd.y : sg[n + 33]) - o1;
        if (hd.y == 0) { c0 = 0; c1 = 0; }
    }
    const unsigned ctot = c0 + c1;
    const unsigned tbase = (unsigned)((size_t)b * kp.nt_max);
    auto slot_of = [&](unsigned c) { return hd.x + (c < c0 ? o0 + c : o1 + (c - c0)); };
    double le[KL];
    int lj[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) { le[k] = -INFINITY; lj[k] = INT_MAX; }
    for (unsigned c = 0; __ballot(c < ctot); ++c) {
        const bool v = c < ctot;
        const unsigned slot = v ? slot_of(c) : 0u;
        double e = v ? sc.item_e[slot] : -INFINITY;
        int jj = v ? (int)((sc.item_tgt[slot] >> 1) - tbase) : INT_MAX;
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            const bool better = (e > le[k]) || (e == le[k] && jj < lj[k]);
            const double te_ = le[k]; const int tj_ = lj[k];
            le[k] = better ? e : te_; lj[k] = better ? jj : tj_;
            e = better ? te_ : e; jj = better ? tj_ : jj;
        }
    }
    double lw[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) lw[k] = exp(le[k]);                            // (exp(-inf) = 0 for an empty entry)
    const double emx = le[0];
    const double need_norm = emx - RP_AFF_NORM_WINDOW;
    double sumsq = 0.0;
#pragma unroll
    for (int k = 0; k < KL; ++k) if (le[k] >= need_norm) sumsq += lw[k] * lw[k];
    {   // candidates inside the norm window that are not among the KL best, in pool order
        const double elast = le[KL - 1]; const int jlast = lj[KL - 1];
        if (__ballot(live && elast >= need_norm && ctot > (unsigned)KL)) {
            for (unsigned c = 0; __ballot(c < ctot); ++c) {
                const bool v = c < ctot;
                const unsigned slot = v ? slot_of(c) : 0u;
                const double e = v ? sc.item_e[slot] : -INFINITY;
                const int jj = v ? (int)((sc.item_tgt[slot] >> 1) - tbase) : INT_MAX;
                const bool want = v && e >= need_norm && ((e < elast) || (e == elast && jj > jlast));
                if (__ballot(want)) { const double wv_ = exp(e); if (want) sumsq += wv_ * wv_; }
            }
        }
    }
    const double nm = sqrt(sumsq);
    if (live) {
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            if (k < keff) {
                const bool ok = lj[k] >= 0 && lj[k] < nt;
                corres_j[(size_t)grow * topK + k] = ok ? lj[k] : 0;
                corres_w[(size_t)grow * topK + k] = (ok && nm != 0.0) ? lw[k] / nm : 0.0;
            }
        }
    }
    if (!WRITE_WIJ) return;
    // window values: wij = exp(e - e_max) * (exp(e_max) / norm), exact zeros below the window (the zeros the screen wrote stay)
    const float rsf = (!live || !(nm > 0.0)) ? 0.f : (float)(lw[0] / nm);
    float* wrow = wij + (size_t)grow * kp.nt_max;
    for (unsigned c = 0; __ballot(c < ctot); ++c) {
        const bool v = live && c < ctot;
        const unsigned slot = v ? slot_of(c) : 0u;
        const double e = v ? sc.item_e[slot] : -INFINITY;
        const int jj = v ? (int)((sc.item_tgt[slot] >> 1) - tbase) : 0;
        const double x = e - emx;
        const double tl = x * 1.4426950408889634;
        const double nr = __builtin_rint(tl);
        const float p2 = __builtin_amdgcn_exp2f((float)(tl - nr));
        float val = ldexpf(p2 * rsf, (int)fmax(nr, -300.0));
        if (!(x >= -RP_AFF_WINDOW && x <= 0.0)) val = 0.f;
        if (v && val != 0.f) rp_stg(wrow + jj, val);
    }
}

template <int TP>
int launch_affinity_rows(const RelposeKeypoints& kp, const AffConsts& ac, int topK, float* wij, int32_t* cj, double* cw, int32_t* keff, hipStream_t s) {
    // ~2 waves per SIMD over the whole chip, between 2 and 32 rows per wave (the per-wave target staging costs ~2 rows' worth)
    const long long rows = (long long)kp.B * kp.ns_max;
    int rpw = (int)((rows + 2047) / 2048);
    rpw = rpw < 2 ? 2 : (rpw > 32 ? 32 : rpw);
    dim3 grid((kp.ns_max + 4 * rpw - 1) / (4 * rpw), kp.B);
    if (wij) hipLaunchKernelGGL((affinity_rows_kernel<TP, true>), grid, dim3(256), 0, s, kp, ac, topK, rpw, wij, cj, cw, keff);
    else hipLaunchKernelGGL((affinity_rows_kernel<TP, false>), grid, dim3(256), 0, s, kp, ac, topK, rpw, wij, cj, cw, keff);
    RP_CHECK_LAUNCH();
    return 0;
}

// scratch of the pool variant: one grow-only block per stream (calls on one stream are ordered; calls on different streams get different blocks)
struct A3Block { void* ptr = nullptr; size_t bytes = 0; };
static std::mutex g_a3_mu;
static std::map<std::pair<int, hipStream_t>, A3Block> g_a3_blocks;

static int a3_scratch(size_t bytes, hipStream_t s, void** out) {
    int dev = 0;
    RP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_a3_mu);
    A3Block& blk = g_a3_blocks[std::make_pair(dev, s)];
    if (blk.bytes < bytes) {
        if (blk.ptr) { RP_HIP(hipStreamSynchronize(s)); RP_HIP(hipFree(blk.ptr)); blk.ptr = nullptr; blk.bytes = 0; }
        RP_HIP(hipMalloc(&blk.ptr, bytes));
        blk.bytes = bytes;
    }
    *out = blk.ptr;
    return 0;
}

template <int TP>
static int a3_launch_fixup(const RelposeKeypoints& kp, const AffConsts& ac, int topK, float* wij, int32_t* cj, double* cw, int32_t* keff, hipStream_t s) {
    const int rpw2 = 32;
    dim3 grid2((kp.ns_max + 4 * rpw2 - 1) / (4 * rpw2), kp.B);
    if (wij) hipLaunchKernelGGL((affinity_rows_kernel<TP, true, true>), grid2, dim3(256), 0, s, kp, ac, topK, rpw2, wij, cj, cw, keff);
    else hipLaunchKernelGGL((affinity_rows_kernel<TP, false, true>), grid2, dim3(256), 0, s, kp, ac, topK, rpw2, wij, cj, cw, keff);
    RP_CHECK_LAUNCH();
    return 0;
}

// screen -> exact -> rank (+ the exact redo of marked rows): see aff3_screen_kernel
static int launch_affinity_pool(const RelposeParams& p, const RelposeKeypoints& kp, const AffConsts& ac, float* wij, int32_t* cj, double* cw, int32_t* keff,
                                hipStream_t s) {
    const long long tiles32 = (long long)kp.B * ((kp.ns_max + 31) / 32);
    const int ntp = (kp.nt_max + 63) & ~63;
    const int atw = tiles32 >= 4096 ? AT_WAVES : (tiles32 >= 1024 ? 4 : 2);
    const int nblk = (kp.ns_max + atw * 32 - 1) / (atw * 32);
    const int rb = (kp.ns_max + nblk - 1) / nblk, rpw = (rb + atw - 1) / atw;
    const size_t rows = (size_t)kp.B * kp.ns_max, tgts = (size_t)kp.B * kp.nt_max;
    if (rows >= (1ull << 31) || tgts >= (1ull << 30)) return RELPOSE_EINVAL;
    const size_t nwaves = (size_t)kp.B * nblk * atw;
    // pool slots per screen wave: 32 candidates per row on average (the bench distributions have ~11; a dense window overflows into the exact redo)
    const size_t capw = (((size_t)rpw * std::min(kp.nt_max, 32)) + 63) & ~(size_t)63;
    const size_t cap = nwaves * capw;
    if (cap >= (1ull << 32)) return RELPOSE_EINVAL;
    size_t off = 0;
    const size_t o_ctl = off; off += 256;
    const size_t o_hdr = off; off += rp_align(nwaves * sizeof(uint2));
    const size_t o_seg = off; off += rp_align(nwaves * 64 * sizeof(unsigned short));
    const size_t o_fs = off; off += rp_align(rows * RP_FEAT * sizeof(float));
    const size_t o_ft = off; off += rp_align(tgts * RP_FEAT * sizeof(float));
    const size_t o_ir = off; off += rp_align(cap * sizeof(unsigned));
    const size_t o_it = off; off += rp_align(cap * sizeof(unsigned));
    const size_t o_ie = off; off += rp_align(cap * sizeof(double));
    void* blk = nullptr;
    const int rc = a3_scratch(off, s, &blk);
    if (rc) return rc;
    char* base = (char*)blk;
    A3Scratch sc;
    sc.ctl = (unsigned*)(base + o_ctl); sc.hdr = (uint2*)(base + o_hdr); sc.seg = (unsigned short*)(base + o_seg);
    sc.fs_sc = (float*)(base + o_fs); sc.ft_sc = (float*)(base + o_ft);
    sc.item_row = (unsigned*)(base + o_ir); sc.item_tgt = (unsigned*)(base + o_it); sc.item_e = (double*)(base + o_ie);
    sc.cap = (unsigned)capw; sc.nwaves = (unsigned)nwaves;
    const size_t lds = (size_t)ntp * (A3_LDB + 4 + 4) + 16;
    dim3 grid(nblk, kp.B);
#define RP_A3_SCREEN(W_, KL_, TPM_) \
    hipLaunchKernelGGL((aff3_screen_kernel<W_, KL_, TPM_>), grid, dim3(atw * 64), lds, s, kp, ac, p.topK, ntp, rb, rpw, wij, cj, keff, sc)
#define RP_A3_SCREEN_T(W_, KL_) { if (ntp <= 256) RP_A3_SCREEN(W_, KL_, 4); else RP_A3_SCREEN(W_, KL_, 8); }
    if (wij) { if (p.topK <= 5) RP_A3_SCREEN_T(true, 5) else RP_A3_SCREEN_T(true, RP_MAXK) }
    else { if (p.topK <= 5) RP_A3_SCREEN_T(false, 5) else RP_A3_SCREEN_T(false, RP_MAXK) }
#undef RP_A3_SCREEN_T
#undef RP_A3_SCREEN
    RP_CHECK_LAUNCH();
    const unsigned eg = (unsigned)((nwaves * (capw >> 6) + 3) / 4);
    hipLaunchKernelGGL(aff3_exact_kernel, dim3(eg), dim3(256), 0, s, sc, ac);
    RP_CHECK_LAUNCH();
    const dim3 rg((unsigned)((rows + 255) / 256));
#define RP_A3_RANK(W_, KL_) hipLaunchKernelGGL((aff3_rank_kernel<W_, KL_>), rg, dim3(256), 0, s, kp, p.topK, rb, rpw, nblk, atw, wij, cj, cw, sc)
    if (wij) { if (p.topK <= 5) RP_A3_RANK(true, 5); else RP_A3_RANK(true, RP_MAXK); }
    else { if (p.topK <= 5) RP_A3_RANK(false, 5); else RP_A3_RANK(false, RP_MAXK); }
#undef RP_A3_RANK
    RP_CHECK_LAUNCH();
    switch ((kp.nt_max + 127) / 128) {
        case 1: return a3_launch_fixup<1>(kp, ac, p.topK, wij, cj, cw, keff, s);
        case 2: return a3_launch_fixup<2>(kp, ac, p.topK, wij, cj, cw, keff, s);
        case 3: return a3_launch_fixup<3>(kp, ac, p.topK, wij, cj, cw, keff, s);
        default: return a3_launch_fixup<4>(kp, ac, p.topK, wij, cj, cw, keff, s);
    }
}

}  // namespace

int rp_launch_affinity(const RelposeParams& p, const RelposeKeypoints& kp, float* wij, int32_t* cj, double* cw, int32_t* keff, hipStream_t s, int sel_call) {
    const int sel = sel_call > 0 ? sel_call : g_rp_tune[RELPOSE_TUNE_AFFINITY_KERNEL];      // 0: by size, 1: row kernel, 2: tile kernel, 3: LDS kernel, 4: pool variant
    const RpPairConsts kc = rp_make_consts(p);
    AffConsts ac;
    ac.den[0] = kc.den_other; ac.den[1] = kc.den_both;
    ac.exact_div = 1;
    for (int q = 0; q < 2; ++q) {
        ac.rden[q] = 1.0 / ac.den[q];
        uint64_t bits; memcpy(&bits, &ac.den[q], 8);
        // Markstein's theorem needs RN(1/den) and excludes an all-ones significand; the LDS kernel divides in hardware otherwise
        if (!(ac.den[q] > 1e-290 && ac.den[q] < 1e290) || (bits & 0xfffffffffffffull) == 0xfffffffffffffull) ac.exact_div = 0;
    }
    if (kp.nt_max <= 512 && ac.exact_div && sel != 3) {
        const int tp = (kp.nt_max + 127) / 128;
        // small batches: the row kernel (one wave per few rows) has the lower latency; the tile kernel pays from ~256 row tiles on
        const long long tiles32 = (long long)kp.B * ((kp.ns_max + 31) / 32);
        const bool use_tile = sel == 2 || (sel == 0 && tiles32 >= RP_AFF_TILE_MIN_TILES);
        // the pool variant: forced, or by itself where it measured faster than the tile kernel -- the fused form (no wij copy) beyond 256 targets
        // (B = 1024, N = 400: 254 vs 342 us; at N = 200 the tile kernel wins 95 vs 117 us, and with a materialised wij everywhere: profiles/r05_affinity_pool.txt)
        if (sel == 4 || (sel == 0 && !wij && kp.nt_max > 256 && tiles32 >= RP_AFF_TILE_MIN_TILES)) return launch_affinity_pool(p, kp, ac, wij, cj, cw, keff, s);
        if (use_tile) {
            // second-generation tile kernel + (normally idle) exact redo of the rows it marked
            const int ntp = (kp.nt_max + 63) & ~63;
            const int atw = tiles32 >= 4096 ? AT_WAVES : (tiles32 >= 1024 ? 4 : 2);
            const int istk_cap = ntp <= 256 ? AT_ISTK : 24;
            // rows: blocks of <= 32 atw rows, equally loaded; inside a block every wave takes rpw consecutive rows
            const int nblk = (kp.ns_max + atw * 32 - 1) / (atw * 32);
            const int rb = (kp.ns_max + nblk - 1) / nblk, rpw = (rb + atw - 1) / atw;
            const size_t lds = (size_t)ntp * (AT_LDT * 4 + 8) + 16 + (size_t)atw * (AT2_CAPW * 6 + istk_cap * 128);
            dim3 grid(nblk, kp.B);
#define RP_TILE2_LAUNCH(W_, KL_)                                                                                                      \
            {                                                                                                                          \
                RP_HIP(hipFuncSetAttribute((const void*)affinity_tile_kernel<W_, KL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                hipLaunchKernelGGL((affinity_tile_kernel<W_, KL_>), grid, dim3(atw * 64), lds, s, kp, ac, p.topK, ntp, rb, rpw, wij, cj, cw, keff, istk_cap); \
            }
            if (wij) { if (p.topK <= 5) RP_TILE2_LAUNCH(true, 5) else RP_TILE2_LAUNCH(true, RP_MAXK) }
            else { if (p.topK <= 5) RP_TILE2_LAUNCH(false, 5) else RP_TILE2_LAUNCH(false, RP_MAXK) }
#undef RP_TILE2_LAUNCH
            RP_CHECK_LAUNCH();
            const int rpw2 = 32;
            dim3 grid2((kp.ns_max + 4 * rpw2 - 1) / (4 * rpw2), kp.B);
#define RP_FIXUP_LAUNCH(TP_)                                                                                                             \
            if (wij) hipLaunchKernelGGL((affinity_rows_kernel<TP_, true, true>), grid2, dim3(256), 0, s, kp, ac, p.topK, rpw2, wij, cj, cw, keff);   \
            else hipLaunchKernelGGL((affinity_rows_kernel<TP_, false, true>), grid2, dim3(256), 0, s, kp, ac, p.topK, rpw2, wij, cj, cw, keff);
            switch (tp) {
                case 1: RP_FIXUP_LAUNCH(1) break;
                case 2: RP_FIXUP_LAUNCH(2) break;
                case 3: RP_FIXUP_LAUNCH(3) break;
                default: RP_FIXUP_LAUNCH(4) break;
            }
#undef RP_FIXUP_LAUNCH
            RP_CHECK_LAUNCH();
            return 0;
        }
        switch (tp) {
            case 1: return launch_affinity_rows<1>(kp, ac, p.topK, wij, cj, cw, keff, s);
            case 2: return launch_affinity_rows<2>(kp, ac, p.topK, wij, cj, cw, keff, s);
            case 3: return launch_affinity_rows<3>(kp, ac, p.topK, wij, cj, cw, keff, s);
            default: return launch_affinity_rows<4>(kp, ac, p.topK, wij, cj, cw, keff, s);
        }
    }
    const int ntp = (kp.nt_max + 63) & ~63;
    const size_t lds = (size_t)ntp * 8 + (size_t)RP_FEAT * (ntp + 1) * 4;
    if (lds > 160 * 1024 || kp.nt_max > RELPOSE_MAX_TARGETS) return RELPOSE_EINVAL;      // (RELPOSE_MAX_TARGETS = what fits: 1152 x 136 B + 128 B < 160 KB)
    const int rows = 8;
    dim3 grid((kp.ns_max + rows - 1) / rows, kp.B);
    if (wij) {
        RP_HIP(hipFuncSetAttribute((const void*)affinity_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(affinity_lds_kernel<true>, grid, dim3(256), lds, s, kp, kc, p.topK, rows, wij, cj, cw, keff);
    } else {
        RP_HIP(hipFuncSetAttribute((const void*)affinity_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(affinity_lds_kernel<false>, grid, dim3(256), lds, s, kp, kc, p.topK, rows, wij, cj, cw, keff);
    }
    RP_CHECK_LAUNCH();
    return 0;
}
