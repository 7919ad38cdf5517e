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
//   affinity_lds_kernel    nt_max > 512: targets transposed in LDS (any size up to 4096)
// The bound is HBM by SURVEY 8(d)'s definition: (Ns + Nt) * 33 * 4 + Ns * Nt * 4 algorithmic bytes per pair-step.
//
// Compiled with -ffp-contract=off: the float32 distance must round like numpy.
#include "matcher_internal.h"
#include <limits.h>
#include <string.h>

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
// materialised, within RP_AFF_WINDOW of it (everything else is an exact 0 in the float32 wij).  A wave owns 32 source rows
// and finds those entries from APPROXIMATE exponents:
//   e~ = -rd (|s|^2 + |t|^2 - 2 s.t), s.t from v_mfma_f32_32x32x16_f16 with the TARGETS as M and the 32 source rows as N: in
//   the C layout lane (n, h) holds entries of ITS OWN row n (16 targets per 32-target tile, lanes n and n + 32 share a row),
//   so row-wise selection is in-lane work (no cross-lane traffic) and the approximate values never leave registers.
// With err bounding |e~ - e| (fp16 operand rounding, derived below) every true winner has e~ >= kth~ - 2 err, where kth~ is the
// K-th largest of the row's group-of-8 maxima (a lower bound of the K-th largest e~), and every entry inside a window W of
// the true maximum has e~ >= max~ - W - 2 err.  Three sweeps over the targets (the MFMAs are recomputed, 4 per 64 targets):
//   P1   max~ and kth~ of every row (one v_max3 chain per 8 entries + one sorted-list insertion per group);
//   P2a  the IMPORTANT entries (possible winners + norm window) onto a per-lane stack, then per lane the exact treatment of
//        affinity_rows_kernel -- numpy-order distance, Markstein division, float64 exp, (e, smaller j) ordering, norm --
//        and the K outputs of the row;
//   P2b  (materialised wij only) the window entries go through a wave-wide ring (ballot compaction) and are evaluated 64 at a
//        time, one per lane whatever row they belong to (source row gathered from its owner lane with ds_bpermute): exact
//        distance, exp(e - max~) on the float32 exp2 unit, scaled by the row's exp(max~)/norm, stored over the zero.
//   The zeros themselves -- the 164 MB of a 1024-pair batch -- are one contiguous span per wave (its 32 rows), written with
//   coalesced 16-byte stores between the P1 tiles, so that they drain while the matrix and vector pipes work (issued in the
//   P2b sweep, when every wave of the chip is in its store phase at once, they cost 47 us instead of 15).
// A row whose stack overflows, or whose data the bound does not cover (weights outside [0, 1], |descriptor| >= 1e4, NaN), is
// marked RP_AFF_REDO and redone by affinity_rows_kernel<FIXUP> (launched right behind, normally a no-op).
// Results: indices and float64 weights identical to affinity_rows_kernel's (same exact arithmetic on a superset of the
// entries that matter; the norm leaves out terms below e^-48 of the largest); float32 wij within 3e-7 relative.
#define AT_WAVES 8             // waves per workgroup = tiles of 32 source rows (small batches launch 2 or 4)
#define AT_LDT 36              // LDS row stride of the target descriptors (floats): conflict-free b128 reads
#define AT_ISTK 16             // important-entry stack per lane (half a row), uint16 target indices; 24 beyond 256 targets
#define AT_QCAP 128            // window-entry ring per wave (flushed 64 at a time)
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
__device__ __forceinline__ void rp_wave_lds_sync() {       // LDS traffic between the lanes of ONE wave (in order in hardware; keeps the compiler from reordering)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <int KL>
__device__ __forceinline__ void rp_list_push(float (&te)[KL], float x) {       // sorted (descending) list of the KL largest values
    float prev = te[0];
    te[0] = fmaxf(prev, x);
#pragma unroll
    for (int k = 1; k < KL; ++k) { const float cur = te[k]; te[k] = __builtin_amdgcn_fmed3f(prev, cur, x); prev = cur; }
}
// numpy-order float32 squared distance of two 32-vectors: 8 strided partial sums + fixed tree (s in registers, t in LDS)
__device__ __forceinline__ float rp_exact_dist(const float (&s)[RP_FEAT], const float* t) {
    float r8[8];
#pragma unroll
    for (int c4 = 0; c4 < RP_FEAT / 4; ++c4) {
        const float4 tv = *reinterpret_cast<const float4*>(t + 4 * c4);
        const float tt[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * c4 + k;
            const float df = s[c] - tt[k];
            const float sq = df * df;
            if (c < 8) r8[c] = sq; else r8[c & 7] = r8[c & 7] + sq;
        }
    }
    return ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
}
// RN(d / den) by Markstein's correction of d * RN(1/den), negated: the exponent of one entry
__device__ __forceinline__ double rp_exponent(float d, double den, double rd) {
    const double x = (double)d;
    double q = x * rd;
    const double rem = __builtin_fma(-q, den, x);
    q = __builtin_fma(rem, rd, q);
    return -q;
}

template <bool WRITE_WIJ, int KL>       // KL = length of the per-lane winner lists (>= topK): 5 or RP_MAXK
__global__ __launch_bounds__(AT_WAVES * 64, 4) void affinity_tile_kernel(RelposeKeypoints kp, AffConsts ac, int topK, int ntp,
                                                                          float* __restrict__ wij, int32_t* __restrict__ corres_j,
                                                                          double* __restrict__ corres_w, int32_t* __restrict__ keff_out, int istk_cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NT = blockDim.x, rows_per_block = (NT >> 6) * 32;      // 2, 4 or 8 waves: small batches use smaller workgroups
    const int b = blockIdx.y;
    const int ns = kp.ns[b], nt = kp.nt[b];
    const int keff = (ns >= 3 && nt >= 3) ? min(topK, nt - 1) : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) keff_out[b] = keff;
    if (keff == 0) return;
    if (blockIdx.x * rows_per_block >= ns) return;
    float* ftT = (float*)smem;                                   // [ntp][AT_LDT] scaled target descriptors (float32, exact)
    float* tab = ftT + (size_t)ntp * AT_LDT;                     // {A_j, B_j} per target: e~ = B_j (|s|^2 - 2 g) + A_j, for rows with weight 1 ...
    const int tab_other = 2 * ntp + 16;                          // ... and for the other rows (16 floats further: other banks)
    int* misc = (int*)(tab + 4 * ntp + 16);                      // [0] max |t|^2 (float bits), [1] pair not covered by the bound
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, n = lane & 31;
    unsigned short* queue = (unsigned short*)(misc + 4) + (size_t)wave * (AT_QCAP + istk_cap * 64);
    unsigned short* istk = queue + AT_QCAP;                      // [istk_cap][64]
    const float nrd0 = -(float)ac.rden[0], nrd1 = -(float)ac.rden[1];
    // this lane's source row (both halves of a wave share it): its loads are issued before the staging so that their latency
    // overlaps the target loads'
    const int i0w = blockIdx.x * rows_per_block + wave * 32;          // first source row of this wave
    const int i = i0w + n;
    const bool rowok = i < ns;
    const size_t si = (size_t)b * kp.ns_max + (rowok ? i : 0);
    float4 fsraw[RP_FEAT / 4];
#pragma unroll
    for (int c4 = 0; c4 < RP_FEAT / 4; ++c4) fsraw[c4] = rowok ? rp_ldg4(kp.feat_s + si * RP_FEAT + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const double wsi = rowok ? rp_ldg(kp.weight_s + si) : 0.0;
    {   // ---- stage the pair's targets: descriptors / 100 (float32 division like numpy), |t|^2, the exponent tables
        const float* ftg = kp.feat_t + (size_t)b * kp.nt_max * RP_FEAT;
        for (int idx = tid; idx < ntp * (RP_FEAT / 4); idx += NT) {
            const int j = idx >> 3, c4 = idx & 7;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < nt) v = rp_ldg4(ftg + (size_t)j * RP_FEAT + 4 * c4);
            *reinterpret_cast<float4*>(&ftT[j * AT_LDT + 4 * c4]) = rp_div100(v);
        }
        if (tid == 0) { misc[0] = 0; misc[1] = 0; }
        const double wt_first = (tid < nt) ? rp_ldg(kp.weight_t + (size_t)b * kp.nt_max + tid) : 0.0;       // (ntp <= NT: one target per thread)
        __syncthreads();
        float mx = 0.f;
        bool bad = false;
        for (int j = tid; j < ntp; j += NT) {
            float a = 0.f;
            for (int c = 0; c < RP_FEAT; ++c) a += ftT[j * AT_LDT + c] * ftT[j * AT_LDT + c];
            const bool ok = j < nt;
            const double wtj = ok ? (j == tid ? wt_first : rp_ldg(kp.weight_t + (size_t)b * kp.nt_max + j)) : 0.0;
            if (ok && (!(wtj >= 0.0 && wtj <= 1.0) || !(a < 1e8f))) bad = true;
            const float b1 = (wtj == 1.0) ? nrd1 : nrd0;
            tab[2 * j] = ok ? b1 * a : -INFINITY;
            tab[2 * j + 1] = ok ? b1 : 0.f;
            tab[tab_other + 2 * j] = ok ? nrd0 * a : -INFINITY;
            tab[tab_other + 2 * j + 1] = ok ? nrd0 : 0.f;
            if (ok) mx = fmaxf(mx, a);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        if (lane == 0) atomicMax(&misc[0], __float_as_int(mx));          // non-negative floats order like ints
        if (bad) misc[1] = 1;
        __syncthreads();
    }
    const float ntmax = __int_as_float(misc[0]);
    const bool pair_bad = misc[1] != 0;
    if (i0w >= ns) return;
    // ---- the source row: all 32 scaled features (exact distances, ds_bpermute source of the flush) + the fp16 half this lane feeds to the MFMA
    float fs[RP_FEAT];
#pragma unroll
    for (int c4 = 0; c4 < RP_FEAT / 4; ++c4) {
        const float4 v = rp_div100(fsraw[c4]);
        fs[4 * c4] = v.x; fs[4 * c4 + 1] = v.y; fs[4 * c4 + 2] = v.z; fs[4 * c4 + 3] = v.w;
    }
    float nsq = 0.f;
#pragma unroll
    for (int c = 0; c < RP_FEAT; ++c) nsq += fs[c] * fs[c];
    f16x8 bf0, bf1;                                                   // B operand: features 8h .. 8h+7 and 16+8h .. 16+8h+7 of row n
    {
        unsigned p[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            p[q] = rp_pkrtz(h ? fs[8 + 2 * q] : fs[2 * q], h ? fs[9 + 2 * q] : fs[1 + 2 * q]);
            p[4 + q] = rp_pkrtz(h ? fs[24 + 2 * q] : fs[16 + 2 * q], h ? fs[25 + 2 * q] : fs[17 + 2 * q]);
        }
        const u32x4 lo = {p[0], p[1], p[2], p[3]}, hi = {p[4], p[5], p[6], p[7]};
        bf0 = __builtin_bit_cast(f16x8, lo);
        bf1 = __builtin_bit_cast(f16x8, hi);
    }
    const bool rowone = wsi == 1.0;
    const bool row_bad = pair_bad || !(wsi >= 0.0 && wsi <= 1.0) || !(nsq < 1e8f);
    const float* tsel = tab + (rowone ? 0 : tab_other);
    const int ntiles = ntp / 32;                                                // even: ntp is a multiple of 64

    // approximate exponents of TWO 32-target tiles for this lane's row: et[u][r] belongs to target j0 + 32 u + 8 (r >> 2) + 4 h + (r & 3)
    auto tile_exponents = [&](int j0, float (&et)[2][16]) {
        floatx16 acc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
            const float* row = &ftT[(j0 + 32 * u + n) * AT_LDT + 8 * h];
            const float4 a0 = *reinterpret_cast<const float4*>(row), a1 = *reinterpret_cast<const float4*>(row + 4);
            const float4 a2 = *reinterpret_cast<const float4*>(row + 16), a3 = *reinterpret_cast<const float4*>(row + 20);
            const u32x4 k0 = {rp_pkrtz(a0.x, a0.y), rp_pkrtz(a0.z, a0.w), rp_pkrtz(a1.x, a1.y), rp_pkrtz(a1.z, a1.w)};
            const u32x4 k1 = {rp_pkrtz(a2.x, a2.y), rp_pkrtz(a2.z, a2.w), rp_pkrtz(a3.x, a3.y), rp_pkrtz(a3.z, a3.w)};
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, k0), bf0, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, k1), bf1, acc[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* tp = tsel + 2 * (j0 + 32 * u + 8 * q + 4 * h);
                const float4 t0 = *reinterpret_cast<const float4*>(tp), t1 = *reinterpret_cast<const float4*>(tp + 4);
                const float A[4] = {t0.x, t0.z, t1.x, t1.z}, B[4] = {t0.y, t0.w, t1.y, t1.w};
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    et[u][4 * q + rr] = __builtin_fmaf(B[rr], __builtin_fmaf(-2.0f, acc[u][4 * q + rr], nsq), A[rr]);
            }
    };
    // target index of bit `bit` of a candidate mask built by m = (m << 1) | pred over (u, r) in order
    auto mask_target = [&](int t, int bit) { const int idx = 31 - bit, u = idx >> 4, r = idx & 15; return (t + u) * 32 + 8 * (r >> 2) + 4 * h + (r & 3); };

    // zero-fill of the wave's rows of wij (ONE contiguous span: rows i0w .. i0w + 31), part `part` of `nparts`: coalesced 16-byte
    // stores issued early (between the P1 tiles) so that the 164 MB of zeros drain while the matrix / vector pipes work
    float* wbase = WRITE_WIJ ? wij + ((size_t)b * kp.ns_max + i0w) * kp.nt_max : nullptr;
    const int span = min(32, ns - i0w) * kp.nt_max;                           // floats
    const bool vec4 = (kp.nt_max & 3) == 0;                                   // span start 16-byte aligned (the buffer itself is)
    auto zero_span = [&](int part, int nparts) {
        if (vec4) {
            const int n4 = span >> 2, per = (n4 + nparts - 1) / nparts, lo = part * per, hi = min(n4, lo + per);
            for (int q = lo + lane; q < hi; q += 64) rp_stg4(wbase + 4 * q, make_float4(0.f, 0.f, 0.f, 0.f));
        } else {
            const int per = (span + nparts - 1) / nparts, lo = part * per, hi = min(span, lo + per);
            for (int q = lo + lane; q < hi; q += 64) rp_stg(wbase + q, 0.f);
        }
    };

    // ---- P1: the row maximum and the KL largest group-of-8 maxima of the half row (sorted, te[0] = max)
    float te[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) te[k] = -INFINITY;
    for (int t = 0; t < ntiles; t += 2) {
        float et[2][16];
        if (WRITE_WIJ) zero_span(t >> 1, ntiles >> 1);
        tile_exponents(t * 32, et);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g8 = 0; g8 < 2; ++g8) {
                const float* e8 = &et[u][8 * g8];
                const float gm = fmaxf(fmaxf(fmaxf(e8[0], e8[1]), fmaxf(e8[2], e8[3])), fmaxf(fmaxf(e8[4], e8[5]), fmaxf(e8[6], e8[7])));
                rp_list_push<KL>(te, gm);
            }
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
    // |e~ - e|: operands rounded toward zero to fp16 (|dx| <= 2^-10 |x| + 2^-24), products and sums in float32 on the matrix pipe:
    // |g~ - g| <= 2^-9 (1 + 2^-10) |s||t| + 2^-24 sqrt(32) (|s| + |t|) (1 + 2^-10) + 32 2^-24 |s||t|, the distance doubles that and adds the
    // float32 roundings of |s|^2, |t|^2 and of the two FMAs (< 6e-6 (|s|^2 + |t|^2)); 2 |s||t| <= |s|^2 + |t|^2.
    const float rdmax = fmaxf(-nrd0, -nrd1);
    const float errd = 2.1e-3f * (nsq + ntmax) + 3.5e-7f * (sqrtf(nsq) + sqrtf(ntmax));
    const float err = errd * rdmax + 1e-6f * fmaxf(fabsf(kth), fabsf(emax_a)) + 1e-30f;
    const float thr_imp = fminf(kth, emax_a - (float)RP_AFF_NORM_WINDOW) - 2.0f * err;
    const float thr_win = emax_a - (float)RP_AFF_WINDOW - 2.0f * err;
    const double need_norm = (double)emax_a - RP_AFF_NORM_WINDOW - 3.0 * (double)err;

    // ---- P2a: important entries onto the lane's stack
    int cnt = 0;
    for (int t = 0; t < ntiles; t += 2) {
        float et[2][16];
        tile_exponents(t * 32, et);
        unsigned m = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = (m << 1) | (et[u][r] >= thr_imp ? 1u : 0u);
        while (m) {
            const int bit = __builtin_ctz(m);
            m &= m - 1;
            const int jc = mask_target(t, bit);
            if (jc < nt) {
                if (cnt < istk_cap) istk[cnt * 64 + lane] = (unsigned short)jc;
                ++cnt;
            }
        }
    }
    const bool over = cnt > istk_cap;
    const int ncand = min(cnt, istk_cap);

    // ---- exact treatment of the important entries: sorted list of the KL best by (e descending, j ascending), norm
    double le[KL];
    int lj[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) { le[k] = -INFINITY; lj[k] = INT_MAX; }
    double sumsq = 0.0;
    {
        constexpr int IL = 1;                                 // entries per trip (2 gives the scheduler independent chains but spills at 128 VGPRs)
        for (int c = 0; __ballot(c < ncand); c += IL) {
            double ev[IL];
            int jv[IL];
#pragma unroll
            for (int q = 0; q < IL; ++q) {
                const bool v = c + q < ncand;
                const int j = v ? istk[(c + q) * 64 + lane] : 0;
                const float d = rp_exact_dist(fs, &ftT[j * AT_LDT]);
                const bool cls = rowone && (tab[2 * j + 1] == nrd1);
                const double e = rp_exponent(d, cls ? ac.den[1] : ac.den[0], cls ? ac.rden[1] : ac.rden[0]);
                ev[q] = v ? e : -INFINITY;
                jv[q] = v ? j : INT_MAX;
            }
#pragma unroll
            for (int q = 0; q < IL; ++q) {
                double e = ev[q];
                int jj = jv[q];
                if (__ballot(e >= need_norm)) { const double w = exp(e); if (e >= need_norm) sumsq += w * w; }
#pragma unroll
                for (int k = 0; k < KL; ++k) {    // insert (e, jj) into the sorted list (an empty entry, (-inf, INT_MAX), never displaces anything)
                    const bool better = (e > le[k]) || (e == le[k] && jj < lj[k]);
                    const double te_ = le[k]; const int tj_ = lj[k];
                    le[k] = better ? e : te_; lj[k] = better ? jj : tj_;
                    e = better ? te_ : e; jj = better ? tj_ : jj;
                }
            }
        }
    }
    {   // merge the two halves of the row: the partner's list, the partner's share of the norm, the partner's overflow flag
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
    // fixed order: (half 0) + (half 1)
    const double s_other = rp_shfl_xor_d(sumsq, 32);
    const double nm = sqrt(h == 0 ? sumsq + s_other : s_other + sumsq);
    const int partner_over = __shfl_xor((int)over, 32, 64);       // unconditionally: a short-circuited shuffle would read inactive lanes
    const bool redo = over || partner_over != 0 || row_bad;
    if (rowok && h == 0) {
        if (redo) corres_j[si * topK] = RP_AFF_REDO;
        else {
#pragma unroll
            for (int k = 0; k < KL; ++k) {
                if (k < keff) {
                    const bool ok = lj[k] >= 0 && lj[k] < nt;
                    corres_j[si * topK + k] = ok ? lj[k] : 0;
                    corres_w[si * topK + k] = (ok && nm != 0.0) ? exp(le[k]) / nm : 0.0;
                }
            }
        }
    }
    if (!WRITE_WIJ) return;

    // ---- the window entries of the wave's 32 rows, evaluated 64 at a time, one per lane whatever row they belong to
    const double rsd = (nm > 0.0) ? exp((double)emax_a) / nm : 0.0;          // wij = exp(e - max~) * exp(max~) / norm
    const float rsf = redo ? 0.f : (float)rsd;                                 // (rows to be redone get zeros here, the fix-up kernel rewrites them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the zero-fill stores of this wave have landed: same-address order
    auto flush = [&](int base, int nitems, int mask) {
        rp_wave_lds_sync();
        const bool act = lane < nitems;
        const int item = act ? queue[(base + lane) & mask] : 0;
        const int rown = item >> 9, j = item & 511, pidx = rown << 2;
        const float em_i = rp_bperm_f(pidx, emax_a), rs_i = rp_bperm_f(pidx, rsf);
        const int one_i = __builtin_amdgcn_ds_bpermute(pidx, rowone ? 1 : 0);
        float d;
        {   // numpy-order distance, the source row gathered from its owner lane 8 features at a time
            const float* tj = &ftT[j * AT_LDT];
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
            }
            d = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
        }
        const bool cls = one_i && (tab[2 * j + 1] == nrd1);
        const double x = rp_exponent(d, cls ? ac.den[1] : ac.den[0], cls ? ac.rden[1] : ac.rden[0]) - (double)em_i;
        // exp(x) * rs in float32: 2^(x log2 e) = 2^nr * 2^fr with |fr| <= 1/2 on the float32 exp2 unit (x <= 2 err, >= -(window + 4 err))
        const double tl = x * 1.4426950408889634;
        const double nr = __builtin_rint(tl);
        const float p2 = __builtin_amdgcn_exp2f((float)(tl - nr));
        float val = ldexpf(p2 * rs_i, (int)fmax(nr, -300.0));
        if (!(x >= -200.0)) val = 0.f;
        if (act) rp_stg(wbase + (size_t)rown * kp.nt_max + j, val);
    };
    // ---- P2b: third sweep: window entries through the wave's ring (ballot compaction), flushed 64 at a time
    int qhead = 0, qn = 0;                                                     // wave-uniform ring state
    for (int t = 0; t < ntiles; t += 2) {
        float et[2][16];
        tile_exponents(t * 32, et);
        unsigned mw = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) mw = (mw << 1) | (et[u][r] >= thr_win ? 1u : 0u);
        if (!rowok || redo) mw = 0;
        while (__ballot(mw != 0)) {
            int jc = INT_MAX;
            if (mw) { const int bit = __builtin_ctz(mw); mw &= mw - 1; jc = mask_target(t, bit); }
            const bool push = jc < nt;
            const unsigned long long bm = __ballot(push);
            const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
            if (push) queue[(qhead + qn + pos) & 127] = (unsigned short)((n << 9) | jc);
            qn += __popcll(bm);
            if (qn >= 64) { flush(qhead, 64, 127); qhead = (qhead + 64) & 127; qn -= 64; }
        }
    }
    if (qn) flush(qhead, qn, 127);
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

}  // namespace

int rp_launch_affinity(const RelposeParams& p, const RelposeKeypoints& kp, float* wij, int32_t* cj, double* cw, int32_t* keff, hipStream_t s) {
    const int sel = g_rp_tune[RELPOSE_TUNE_AFFINITY_KERNEL];      // 0: by size, 1: row kernel, 2: tile kernel, 3: LDS kernel
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
        if (use_tile) {
            // tile kernel + (normally idle) exact redo of the rows it marked
            const int ntp = (kp.nt_max + 63) & ~63;
            // waves (32-row tiles) per workgroup: 8 when the batch fills the chip anyway, fewer for small batches (the per-workgroup
            // target staging is then paid more often, but more CUs work)
            const int atw = tiles32 >= 4096 ? AT_WAVES : (tiles32 >= 1024 ? 4 : 2);
            const int istk_cap = ntp <= 256 ? AT_ISTK : 24;          // (more than 256 targets: one workgroup per CU anyway)
            const size_t lds = (size_t)ntp * AT_LDT * 4 + ((size_t)4 * ntp + 16) * 4 + 16 + (size_t)atw * (AT_QCAP + istk_cap * 64) * 2;
            dim3 grid((kp.ns_max + atw * 32 - 1) / (atw * 32), kp.B);
#define RP_TILE_LAUNCH(W_, KL_)                                                                                                       \
            {                                                                                                                          \
                RP_HIP(hipFuncSetAttribute((const void*)affinity_tile_kernel<W_, KL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                hipLaunchKernelGGL((affinity_tile_kernel<W_, KL_>), grid, dim3(atw * 64), lds, s, kp, ac, p.topK, ntp, wij, cj, cw, keff, istk_cap); \
            }
            if (wij) { if (p.topK <= 5) RP_TILE_LAUNCH(true, 5) else RP_TILE_LAUNCH(true, RP_MAXK) }
            else { if (p.topK <= 5) RP_TILE_LAUNCH(false, 5) else RP_TILE_LAUNCH(false, RP_MAXK) }
#undef RP_TILE_LAUNCH
            RP_CHECK_LAUNCH();
            const int rpw = 32;
            dim3 grid2((kp.ns_max + 4 * rpw - 1) / (4 * rpw), kp.B);
#define RP_FIXUP_LAUNCH(TP_)                                                                                                             \
            if (wij) hipLaunchKernelGGL((affinity_rows_kernel<TP_, true, true>), grid2, dim3(256), 0, s, kp, ac, p.topK, rpw, wij, cj, cw, keff);   \
            else hipLaunchKernelGGL((affinity_rows_kernel<TP_, false, true>), grid2, dim3(256), 0, s, kp, ac, p.topK, rpw, wij, cj, cw, keff);
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
    if (lds > 160 * 1024) return RELPOSE_EINVAL;
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
