// SCNet (completion + feature encoder-decoder) on gfx950.
// Replaces the reference module model/mymodel.py:141-380 (skipLayer=1, batchnorm=1, 'rgbdnsf').
//
// Design (DESIGN.md "SCNet"):
//  * activations live in HBM as NHWC float32, RAW conv outputs (pre-BatchNorm); every buffer has a
//    companion per-(group,channel) {scale,shift} table: float64 statistics written by the producing
//    kernel's epilogue + a fixed-order finalize (BatchNorm uses batch statistics over each group of
//    2 images, mymodel.py:19,32);
//  * one implicit-GEMM kernel for every conv / transposed conv: M = output pixels, N = Cout,
//    K = taps x Cin, k-tiles of 32.  The A-tile loader gathers the im2col slice from up to two NHWC
//    sources (skip concatenations are never materialised), applies scale/shift + LeakyReLU(0.1)
//    on the fly and zero-pads; B = weights pre-packed [Cout][K].  Tiles are staged through LDS
//    (row stride 36 floats: conflict-free ds_read_b128) and contracted with
//    v_mfma_f32_32x32x2_f32 (exact fp32, 64 lanes) -- or, opt-in, with three bf16 MFMA products per
//    fp32 product (relpose_scnet_set_precision);
//  * stride-2 transposed convs are decomposed into their sub-pixel phases (4 members of a 2x2-tap
//    conv, launched phase-interleaved so one XCD's L2 serves all four) so no multiply-by-zero work
//    is issued; shared-weight encoder streams (self / warped view) are channel blocks of one
//    concatenated buffer; conv1 (K = 18/36) and the five 1x1 heads have their own direct kernels;
//  * bilinear resize kernels (align_corners=False) in and out.
#include "common.h"
#include <map>
#include <string>
#include <vector>
#include <algorithm>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>
#include <stdlib.h>
#include <stddef.h>

#ifndef RP_EXPERIMENTS
// (the two-part forward exists in the experiments build only, include/relpose.h; the product build rejects these flag bits)
enum { RELPOSE_FWD_PART_FRONT = 8, RELPOSE_FWD_PART_BACK = 16 };
#endif

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int SPLIT> struct SplitT { typedef bf16x8 v8; typedef bf16x4 v4; };
template <> struct SplitT<2> { typedef f16x8 v8; typedef f16x4 v4; };
template <> struct SplitT<3> { typedef f16x8 v8; typedef f16x4 v4; };     // plain fp16 products (hi halves only), fp32 accumulate
// SPLIT 4 / 5 (primary template: bf16): EXACT-product emulation of the fp32 contraction on the bf16 matrix pipe.  Every fp32 operand is
// cut into three bfloat16 pieces a = a1 + a2 + a3 (8 + 8 + 8 = 24 significand bits: the sum is exact, bf16 has the fp32 exponent range),
// every partial product ai * bj is exact in the fp32 accumulator (16 significand bits), and a * b = sum of the 9 partial products.
// SPLIT 4 ("bf16x9") issues all nine v_mfma_f32_32x32x16_bf16 terms, smallest first; SPLIT 5 ("bf16x6") drops a2 b3, a3 b2, a3 b3
// (the two larger ones <= 2^-24 |a b| each: together at most 2^-23 |a b|, rms 2^-27.4 -- the size of one fp32 rounding, of which the accumulation that follows makes one per product).
typedef float rp_f4v __attribute__((ext_vector_type(4)));

#ifndef RP_ABLATE
#define RP_ABLATE 0
#endif
#ifndef RP_C1_ABLATE
#define RP_C1_ABLATE 0
#endif
#ifndef RP_BK
#define RP_BK 32
#endif
#ifndef RP_AGPR
#define RP_AGPR 0               // 1 = fp32 MFMA accumulators in AccVGPRs through inline asm (experiment)
#endif
#ifndef RP_TILE_ABLATE
#define RP_TILE_ABLATE 0
#endif
#ifndef RP_TILE_TIMING
#define RP_TILE_TIMING 0
#endif
#if RP_TILE_TIMING
__device__ unsigned long long g_tile_timing[8];
extern "C" int relpose_debug_tile_timing(unsigned long long* out8_host, int reset) {
    if (out8_host && hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_tile_timing), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_timing), z, 64) != hipSuccess) return -1; }
    return 0;
}
#endif
#ifndef RP_STAGGER
#define RP_STAGGER 0            // 0 = off; n = workgroups (blockIdx.x / n) % 3 get a start offset (see the k-loop prologue)
#endif
#ifndef RP_STAGGER_SLEEP
#define RP_STAGGER_SLEEP 64     // s_sleep units of 64 cycles
#endif
constexpr int BK = RP_BK;       // K-tile (floats): 16 (double-buffered LDS) or 32 (whole 128-B lines per row, single LDS buffer)
constexpr int LDK = BK + 4;     // LDS row stride in floats (80 / 144 B: 16-B aligned, conflict-free b128 reads)
constexpr int KQ = BK / 4;      // float4 slots per tile row
constexpr int LDK3 = 52;        // LDS row stride (floats) of the three-piece bf16 rows [32 hi | 32 mid | 32 lo | pad]: 208 B, an odd multiple of 16 B
template <int SPLIT> struct RowLd { static constexpr int v = SPLIT >= 4 ? LDK3 : LDK; };
constexpr bool rp_split3(int split) { return split >= 4; }
constexpr int NBUF = (BK == 16) ? 2 : 1;
constexpr float LRELU = 0.1f;
constexpr double BN_EPS = 1e-5;
constexpr int RS = 224;         // internal resolution (mymodel.py:261)

struct Src {
    const float* x;        // NHWC base (+ channel offset)
    const float2* ss;      // [G][sstride] {scale, shift} (+ channel offset); identity table for the raw net input
    int cstride;           // floats per pixel
    int C;                 // channels read from this source
    int sstride;           // float2 per group
    float slope;           // LeakyReLU slope applied after scale/shift (1.0 = no activation)
};

struct ConvDesc {
    Src src[2];
    int nsrc, Cin;
    int Nimg, Hin, Win;
    int Hp, Wp, sy, sx;    // output grid of this launch and input step
    int ntaps;
    signed char offy[16], offx[16];
    const float* w;        // [CoutPad][ntaps*Cin]
    int Cout;
    float* y;              // NHWC output
    int Hout, Wout, ycstride, ychoff, osy, osx, py, px;
    const float* bias;     // heads only
    int tanh_out;
    int M, K;
    int ksplit, kt_per;    // split-K: slices along K and k-tiles per slice (ksplit==1: direct store)
    int tap_inner;         // K order of the main loop: 1 = channel chunk outer / tap inner, 0 = tap outer
    float wscale;          // f16x3 mode: the packed weights carry a power-of-two factor 1/wscale (keeps their lo halves normal); 1 otherwise
    int ntiles_n;          // N tiles (grid.y = ntiles_n * ksplit)
    float* partial;        // [ksplit][M][CoutPad] partial sums when ksplit > 1
    double* stat_part;     // [mtiles][2 group slots][CoutPad][2] per-tile BatchNorm partial sums (or null)
    int cout_pad;
    int stat_bm;           // rows per tile of the kernel that writes stat_part (bn_finalize_fused_kernel walks the records with it)
    int shared_slices;     // RELPOSE_FWD_ZERO_WARP, conv_s2_strip_kernel: bit ks = K slice ks is the same for every image pair: computed for
                           // images 0, 1 only, and the split-K reduce reads row (img & 1, pixel) of it for every image
    int skip_slices;       // self-stream cache (relpose_scnet_forward4), conv_s2_strip_kernel: bit ks = K slice ks is NOT computed -- its partial
                           // sums are still in `partial` from the forward that filled the cache (conv4's partials have a region of their own)
    // self-stream cache, deconv_tile_kernel (fp32 products): the skip source (src[1]: the self-view block of A3 / A2, level-invariant) is
    // accumulated FIRST in every plan; snap_mode 1 stores the accumulators after those chunks to `snap` (thread-private layout, coalesced),
    // snap_mode 2 starts from that snapshot and runs only the chunks of src[0] -- bitwise the accumulator chain of the full forward
    int snap_mode;
    float* snap;
};

__device__ __forceinline__ float lrelu(float v, float slope) { return fmaxf(v, slope * v); }   // slope in (0,1]

// buffer_load_dwordx4 v, voff, s[rsrc], soff offen: 128-bit SGPR descriptor, 32-bit lane byte offset, SGPR byte offset
typedef float rp_f32x4g __attribute__((__vector_size__(16)));
__device__ __forceinline__ float4 rp_bufld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const rp_f32x4g f = (rp_f32x4g)__builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(f[0], f[1], f[2], f[3]);
}
typedef float rp_f32x2g __attribute__((__vector_size__(8)));
__device__ __forceinline__ float2 rp_bufld2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const rp_f32x2g f = (rp_f32x2g)__builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return make_float2(f[0], f[1]);
}

// ---- the split-operand contraction of one staged 32-channel chunk (one tap) for MI x NI accumulators --------------------------------------
// Staged rows hold NP pieces of 32 16-bit values each ([hi | lo] or [hi | mid | lo], 64 bytes per piece); a 16-channel step reads one 16-byte
// fragment per piece and row and issues the mode's partial products as v_mfma_f32_32x32x16_{f16,bf16}:
//   SPLIT 1 / 2 (x3): lo hi, hi lo, hi hi       SPLIT 3 (plain f16): hi hi       SPLIT 5 (bf16x6): lo hi, hi lo, mid mid, mid hi, hi mid, hi hi
//   SPLIT 4 (bf16x9): lo lo, lo mid, mid lo, then the six of SPLIT 5
// Terms in ascending magnitude; where a wave holds several accumulators (MI NI > 1) the loop is TERM-major, so consecutive MFMAs go to different
// accumulators (a dependent v_mfma_f32_32x32x16 issues ~40 cycles after its predecessor, an independent one after 32: profiles/r06_mfma_bf16_chain.txt).
// That order is free; giving the one-accumulator tiles a second accumulator chain was measured and LOST (16 more registers: spills in
// deconv_tile_kernel; profiles/r06_split_experiments.txt 4) -- the dependent-accumulator latency is not what holds these kernels.
// The summation order is fixed per instantiation, so the results stay deterministic and plan-invariant (the self-stream cache and the
// level-0 plan run the same calls in the same order as the full forward).
template <int SPLIT> struct SplitTerms;
template <> struct SplitTerms<1> { static constexpr int NP = 2, NT = 3; static constexpr int ta[3] = {1, 0, 0}, tb[3] = {0, 1, 0}; };
template <> struct SplitTerms<2> { static constexpr int NP = 2, NT = 3; static constexpr int ta[3] = {1, 0, 0}, tb[3] = {0, 1, 0}; };
template <> struct SplitTerms<3> { static constexpr int NP = 1, NT = 1; static constexpr int ta[1] = {0}, tb[1] = {0}; };
template <> struct SplitTerms<4> { static constexpr int NP = 3, NT = 9;
                                   static constexpr int ta[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, tb[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}; };
template <> struct SplitTerms<5> { static constexpr int NP = 3, NT = 6;
                                   static constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0}; };
template <int SPLIT>
__device__ __forceinline__ floatx16 rp_mfma16(const typename SplitT<SPLIT>::v8& a, const typename SplitT<SPLIT>::v8& b, const floatx16& c) {
    if constexpr (SPLIT == 2 || SPLIT == 3) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int SPLIT, int MI, int NI>
__device__ __forceinline__ void rp_split_mma(floatx16 (&acc)[MI][NI], const float* At, const int (&a_rows)[MI], const float* Bt, const int (&b_rows)[NI]) {
    typedef typename SplitT<SPLIT>::v8 h8;
    typedef SplitTerms<SPLIT> TT;
    constexpr int NP = TT::NP, NT = TT::NT;
    {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            h8 a[NP][MI], b[NP][NI];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[pc][i] = *reinterpret_cast<const h8*>(&At[a_rows[i] + pc * 16 + st * 8]);
#pragma unroll
                for (int j = 0; j < NI; ++j) b[pc][j] = *reinterpret_cast<const h8*>(&Bt[b_rows[j] + pc * 16 + st * 8]);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = rp_mfma16<SPLIT>(a[TT::ta[t]][i], b[TT::tb[t]][j], acc[i][j]);
        }
    }
}

// Tile: WM x WN waves (WM*WN = 4), each wave MI x NI MFMA 32x32 blocks.
// Both tiles are register-staged: the global loads of tile kt+1 (A values, their BatchNorm
// scale/shift, B weights) are issued BEFORE the MFMAs of tile kt and consumed AFTER them, so their
// latency hides under 64 MFMAs; the loader is branch-free (clamped addresses + selects) so hipcc
// keeps the loads in flight across the MFMA block.
template <int WM, int WN, int MI, int NI, bool SSLDS, bool UNI = false, int SPLIT = 0>
__global__ __launch_bounds__(WM * WN * 64, (NI == 4 || SPLIT >= 4) ? 2 : ((WM * WN == 8 || MI == 1) ? 4 : (SSLDS ? 3 : 2))) void conv_igemm_kernel(const ConvDesc* __restrict__ descs, int ninner, int mt_max) {
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    constexpr int LD = RowLd<SPLIT>::v;                 // LDS row stride (floats); three-piece bf16 rows: [32 hi | 32 mid | 32 lo | pad]
    constexpr bool S3 = SPLIT >= 4;
    // Block order.  ninner == 1: member-major (each member's weights stay L2-resident while it runs).
    // ninner == 4 (the sub-pixel phases of one transposed conv, which gather from the SAME input tile):
    // the 4 phases of a spatial tile run back to back on the SAME XCD (blocks are dealt round-robin to the
    // 8 XCDs), so the tile is fetched from HBM once and re-used from that XCD's L2 by the other phases.
    int zmem, mtile;
    if (ninner == 1) { zmem = blockIdx.x / mt_max; mtile = blockIdx.x - zmem * mt_max; }
    else {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, t8 = (mt_max + 7) >> 3;
        const int inner = slot % ninner, rest = slot / ninner;
        zmem = (rest / t8) * ninner + inner;
        mtile = (rest % t8) * 8 + xcd;
    }
    const ConvDesc d = descs[zmem];                   // block-uniform: scalar loads
    if (mtile * BM >= d.M) return;                    // groups share a grid; shorter members exit
    constexpr int NT = WM * WN * 64;                    // threads per block (4 or 8 waves)
    constexpr int RPI = NT / KQ;                        // tile rows covered per loader iteration
    constexpr int A_IT = BM / RPI;                      // float4 slots per thread for the A tile
    constexpr int B_IT = (BN + RPI - 1) / RPI;
    constexpr int SS_CAP = (WN == 2) ? 2048 : 512;      // float2 entries of the LDS scale/shift table (16 / 4 KB: still 3 blocks per CU)
    __shared__ __attribute__((aligned(16))) float As[NBUF][BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[NBUF][BN * LD];
    __shared__ __attribute__((aligned(16))) float sstab[SSLDS ? 2 * SS_CAP : 8];   // per 4 channels: 4 scales, then 4 shifts
    __shared__ __attribute__((aligned(16))) int rowpix[BM];
    __shared__ signed char rowslot[BM];        // BatchNorm group of the row relative to the tile's first group
    __shared__ int tapdelta[16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ks = blockIdx.y / d.ntiles_n;
    const int m0 = mtile * BM, n0 = (blockIdx.y - ks * d.ntiles_n) * BN;
    const int hw = d.Hp * d.Wp;
    const int g0 = (m0 / hw) >> 1;                      // first BatchNorm group touched by this tile

    // Per tile row (computed once, by one thread per row): pixel index of the (tap 0,0) origin, a 16-bit mask
    // of the taps that fall inside the image (everything else is zero padding) packed with the BatchNorm
    // group, the output pixel and the group slot.  The loader table lives in the (not yet used) A buffer.
    const int kq = tid % KQ, lrow = tid / KQ;
    int* rtab = reinterpret_cast<int*>(&As[0][0]);      // [BM][2] {base, mask | grp << 16}
    for (int row_ = tid; row_ < BM; row_ += NT) {
        const int m = m0 + row_;
        int pix = -1, base = 0, mg = 0, slot = -1;
        if (m < d.M) {
            const int img = m / hw, rem = m - img * hw;
            const int yp = rem / d.Wp, xp = rem - yp * d.Wp;
            pix = (img * d.Hout + yp * d.osy + d.py) * d.Wout + xp * d.osx + d.px;
            slot = min((img >> 1) - g0, 1);
            const int y0 = yp * d.sy, x0 = xp * d.sx;
            base = (img * d.Hin + y0) * d.Win + x0;
            int msk = 0;
            for (int t = 0; t < d.ntaps; ++t) {
                const int iy = y0 + descs[zmem].offy[t], ix = x0 + descs[zmem].offx[t];
                msk |= ((iy >= 0) & (iy < d.Hin) & (ix >= 0) & (ix < d.Win)) ? (1 << t) : 0;
            }
            mg = msk | ((SSLDS ? ((img >> 1) - g0) * d.Cin : (img >> 1)) << 16);
        }
        rowpix[row_] = pix; rowslot[row_] = (signed char)slot;
        rtab[2 * row_] = base; rtab[2 * row_ + 1] = mg;
    }
    if (tid < 16) tapdelta[tid] = (int)descs[zmem].offy[tid] * d.Win + (int)descs[zmem].offx[tid];
    if (SSLDS) {
        // scale/shift of every (group, input channel) this tile can touch; both sources concatenated
        const int ng = min(((min(m0 + BM, d.M) - 1) / hw >> 1) - g0 + 1, SS_CAP / d.Cin);
        for (int idx = tid; idx < ng * d.Cin; idx += NT) {
            const int g = g0 + idx / d.Cin, c = idx % d.Cin;
            const bool s1 = c >= d.src[0].C;
            const float2 e = rp_ldg2(reinterpret_cast<const float*>(s1 ? d.src[1].ss + (size_t)g * d.src[1].sstride + (c - d.src[0].C)
                                                                       : d.src[0].ss + (size_t)g * d.src[0].sstride + c));
            sstab[(idx & ~3) * 2 + (idx & 3)] = e.x; sstab[(idx & ~3) * 2 + 4 + (idx & 3)] = e.y;
        }
    }

    floatx16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#if RP_STAGGER
    // Phase-stagger the workgroups that share a CU (experiment): identical blocks launched together run their k loops in
    // lockstep, so their MFMA-free phases (transform + LDS store + barriers) coincide on every SIMD and the matrix pipe
    // idles; a one-time offset of 1/3 and 2/3 of a k-tile period de-phases the three resident blocks.
    {
        const int ph = (blockIdx.x / RP_STAGGER) % 3;
        if (ph >= 1) __builtin_amdgcn_s_sleep(RP_STAGGER_SLEEP);
        if (ph >= 2) __builtin_amdgcn_s_sleep(RP_STAGGER_SLEEP);
    }
#endif

    // B rows of this thread (BN < 64: rows are clamped, the extra threads re-load row BN-1 and never store)
    const float* b_src[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int row = min(lrow + it * RPI, BN - 1);
        b_src[it] = d.w + (size_t)(n0 + row) * d.K * (S3 ? 3 : 2) / 2 + kq * 4;      // (S3: 6 bytes per weight; the row's lo piece at + 32 - 2 kq)
    }

    int r_base[A_IT], r_mg[UNI ? (A_IT + 1) / 2 : A_IT];     // UNI: no group field, two 16-bit tap masks per register
    float4 ra[A_IT], q0[SSLDS ? 1 : A_IT], q1[SSLDS ? 1 : A_IT];
    float4 rb[B_IT];
    float2 rbl[S3 ? B_IT : 1];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) rb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    static_assert(B_IT <= 8 && (B_IT <= 2 || BN % RPI == 0), "B loader layout");
    int okm = 0, cst = 0;
    const float slope = d.src[0].slope;                 // both sources of a skip concatenation use LeakyReLU(0.1)
    const int kt_begin = ks * d.kt_per;
    const int nkt = min(d.K / BK, kt_begin + d.kt_per);
    // K order of the loop: channel chunk outer, TAP INNER -- the 2x2 / 4x4 taps of one 32-channel chunk touch the same
    // 128-byte lines (neighbouring pixels) in consecutive k-tiles, so they hit in L2; tap-major order re-read every
    // line Cin/32 k-tiles later, after the other workgroups of the XCD had flushed the 4 MB L2 (16x HBM re-fetch on
    // deconv2, PMC).  The weight rows stay [tap][Cin].
    const int cpt = d.Cin / BK;                         // channel chunks per tap
    int tap = d.tap_inner ? kt_begin % d.ntaps : kt_begin / cpt, c0 = (d.tap_inner ? kt_begin / d.ntaps : kt_begin % cpt) * BK;
    __syncthreads();       // tapdelta / rowpix / sstab / rtab visible
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int2 e = *reinterpret_cast<const int2*>(&rtab[2 * (lrow + it * RPI)]);
        r_base[it] = e.x;
        if (!UNI) r_mg[UNI ? 0 : it] = e.y;
        else if (it & 1) r_mg[it >> 1] |= e.y << 16;
        else r_mg[it >> 1] = e.y & 0xffff;
    }

#define RP_ISSUE_LOADS(KT)                                                                                        \
    {                                                                                                             \
        const bool s1_ = (d.nsrc > 1) && (c0 >= d.src[0].C);                                                      \
        const float* sx_ = s1_ ? d.src[1].x : d.src[0].x;                                                         \
        const int scs_ = s1_ ? d.src[1].cstride : d.src[0].cstride;                                               \
        const int cc_ = c0 - (s1_ ? d.src[0].C : 0) + kq * 4;                                                     \
        const int td_ = tapdelta[tap];                                                                            \
        okm = 0; cst = c0 + kq * 4;                                                                               \
        _Pragma("unroll") for (int it = 0; it < A_IT; ++it) {                                                     \
            const bool ok_ = UNI ? ((r_mg[it >> 1] >> (tap + 16 * (it & 1))) & 1) : ((r_mg[UNI ? 0 : it] >> tap) & 1);                                                             \
            okm |= ok_ ? (1 << it) : 0;                                                                           \
            const int pix_ = (RP_ABLATE == 6) ? (it * 8) : (RP_ABLATE == 7) ? ((r_base[it] + td_) & 0xfff) : ok_ ? r_base[it] + td_ : 0;   \
            ra[it] = rp_ldg4(sx_ + (size_t)pix_ * scs_ + cc_);                                                    \
            if (!SSLDS && !UNI) {                                                                                 \
                const float2* sss_ = s1_ ? d.src[1].ss : d.src[0].ss;                                             \
                const int sst_ = s1_ ? d.src[1].sstride : d.src[0].sstride;                                       \
                const float* q = reinterpret_cast<const float*>(sss_ + (size_t)(r_mg[UNI ? 0 : it] >> 16) * sst_ + cc_);      \
                q0[SSLDS ? 0 : it] = rp_ldg4(q); q1[SSLDS ? 0 : it] = rp_ldg4(q + 4);                             \
            }                                                                                                     \
        }                                                                                                         \
        if (!SSLDS && UNI) {                                                                                      \
            const float2* sss_ = s1_ ? d.src[1].ss : d.src[0].ss;                                                 \
            const int sst_ = s1_ ? d.src[1].sstride : d.src[0].sstride;                                           \
            const float* q = reinterpret_cast<const float*>(sss_ + (size_t)g0 * sst_ + cc_);                      \
            qu0 = rp_ldg4(q); qu1 = rp_ldg4(q + 4);                                                               \
        }                                                                                                         \
        const int kof_ = (tap * d.Cin + c0) * (S3 ? 3 : 2) / 2;                                                   \
        _Pragma("unroll") for (int it = 0; it < B_IT; ++it) {                                                     \
            rb[it] = rp_ldg4(b_src[it] + kof_);                                                                   \
            if (S3) { const float2 l_ = rp_ldg2(b_src[it] + kof_ + 32 - kq * 2); rbl[S3 ? it : 0] = l_; }          \
        }                                                                                                         \
        if (d.tap_inner) { ++tap; if (tap == d.ntaps) { tap = 0; c0 += BK; } }                                    \
        else { c0 += BK; if (c0 == d.Cin) { c0 = 0; ++tap; } }                                                    \
    }

#define RP_STORE_TILE(BUF)                                                                                        \
    {                                                                                                             \
        /* BatchNorm scale/shift + LeakyReLU + zero padding as packed fp32 ops (v_pk_fma / v_pk_mul): u0_ = the   \
           4 scales, u1_ = the 4 shifts of this thread's channels */                                             \
        float4 u0_ = make_float4(qu0.x, qu0.z, qu1.x, qu1.z), u1_ = make_float4(qu0.y, qu0.w, qu1.y, qu1.w);      \
        if (SSLDS && UNI) {                                                                                       \
            const float4* q = reinterpret_cast<const float4*>(&sstab[2 * cst]);                                   \
            u0_ = q[0]; u1_ = q[1];                                                                               \
        }                                                                                                         \
        const rp_v2f sl2_ = {slope, slope};                                                                       \
        _Pragma("unroll") for (int it = 0; it < A_IT; ++it) {                                                     \
            float4 s0_, s1v_;                                                                                     \
            if (UNI) { s0_ = u0_; s1v_ = u1_; }                                                                   \
            else if (SSLDS) {                                                                                     \
                const float4* q = reinterpret_cast<const float4*>(&sstab[2 * ((r_mg[UNI ? 0 : it] >> 16) + cst)]);          \
                s0_ = q[0]; s1v_ = q[1];                                                                          \
            } else {                                                                                              \
                const float4 a_ = q0[SSLDS ? 0 : it], b_ = q1[SSLDS ? 0 : it];                                    \
                s0_ = make_float4(a_.x, a_.z, b_.x, b_.z); s1v_ = make_float4(a_.y, a_.w, b_.y, b_.w);            \
            }                                                                                                     \
            rp_v2f v01 = {ra[it].x, ra[it].y}, v23 = {ra[it].z, ra[it].w};                                        \
            v01 = v01 * (rp_v2f){s0_.x, s0_.y} + (rp_v2f){s1v_.x, s1v_.y};                                        \
            v23 = v23 * (rp_v2f){s0_.z, s0_.w} + (rp_v2f){s1v_.z, s1v_.w};                                        \
            const rp_v2f t01 = v01 * sl2_, t23 = v23 * sl2_;                                                      \
            const float okf_ = ((okm >> it) & 1) ? 1.f : 0.f;                                                     \
            const rp_v2f mk_ = {okf_, okf_};                                                                      \
            v01 = (rp_v2f){fmaxf(v01.x, t01.x), fmaxf(v01.y, t01.y)} * mk_;                                       \
            v23 = (rp_v2f){fmaxf(v23.x, t23.x), fmaxf(v23.y, t23.y)} * mk_;                                       \
            if (!SPLIT) *reinterpret_cast<float4*>(&As[BUF][(lrow + it * RPI) * LD + kq * 4]) = make_float4(v01.x, v01.y, v23.x, v23.y); \
            else {   /* row = [32 x 16-bit hi | 32 x 16-bit lo]: v = hi + lo to 2^-16 (bf16) / ~2^-22 (f16); S3: [hi | mid | lo], exact */ \
                typedef typename SplitT<SPLIT>::v4 h4_;                                                           \
                const rp_f4v vf_ = {v01.x, v01.y, v23.x, v23.y};                                                  \
                const h4_ hi_ = __builtin_convertvector(vf_, h4_);                                                \
                const rp_f4v r1_ = vf_ - __builtin_convertvector(hi_, rp_f4v);                                    \
                const h4_ lo_ = __builtin_convertvector(r1_, h4_);                                                \
                h4_* ar_ = reinterpret_cast<h4_*>(&As[BUF][(lrow + it * RPI) * LD]);                              \
                ar_[kq] = hi_;                                                                                    \
                if (SPLIT != 3) ar_[8 + kq] = lo_;                                                                \
                if (S3) ar_[16 + kq] = __builtin_convertvector(r1_ - __builtin_convertvector(lo_, rp_f4v), h4_);  \
            }                                                                                                     \
        }                                                                                                         \
        if (BN >= RPI || lrow < BN) {                                                                             \
            *reinterpret_cast<float4*>(&Bs[BUF][lrow * LD + kq * 4]) = rb[0];                                     \
            if (S3) *reinterpret_cast<float2*>(&Bs[BUF][lrow * LD + 32 + kq * 2]) = rbl[0];                       \
        }                                                                                                         \
        _Pragma("unroll") for (int it = 1; it < B_IT; ++it) {                                                     \
            *reinterpret_cast<float4*>(&Bs[BUF][(lrow + it * RPI) * LD + kq * 4]) = rb[it];                       \
            if (S3) *reinterpret_cast<float2*>(&Bs[BUF][(lrow + it * RPI) * LD + 32 + kq * 2]) = rbl[S3 ? it : 0]; \
        }                                                                                                         \
    }

    // UNI (chosen by the host when the rows of a BatchNorm group are a multiple of BM, i.e. no tile of the
    // launch straddles two groups): the scale/shift of a k-tile is read ONCE per thread instead of per A slot.
    float4 qu0 = make_float4(0.f, 0.f, 0.f, 0.f), qu1 = qu0;
    RP_ISSUE_LOADS(kt_begin)
    __syncthreads();       // every thread has its rtab rows in registers: the A buffer may be overwritten
    RP_STORE_TILE(0)
    __syncthreads();
    const int arow = (wm * MI * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
    const int brow = (wn * NI * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
    for (int kt = kt_begin; kt < nkt; ++kt) {
        const int buf = (NBUF == 2) ? ((kt - kt_begin) & 1) : 0;
#if RP_ABLATE != 2 && RP_ABLATE != 5
        if (kt + 1 < nkt) RP_ISSUE_LOADS(kt + 1)
#endif
        if constexpr (SPLIT != 0) {
            // split-operand modes: 3 / 1 / 9 / 6 16-bit MFMA terms per product and 16-channel step (rp_split_mma: no two consecutive MFMAs share an accumulator)
            int ar_[MI], br_[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) ar_[i] = arow + i * 32 * LD;
#pragma unroll
            for (int j = 0; j < NI; ++j) br_[j] = brow + j * 32 * LD;
            rp_split_mma<SPLIT, MI, NI>(acc, &As[buf][0], ar_, &Bs[buf][0], br_);
        } else
#if RP_AGPR
        {   // experiment: accumulators pinned to AccVGPRs (the compiler picks the ArchVGPR form of the MFMA); fragments of step
            // kc + 1 are read from LDS before the MFMAs of step kc are issued; k-major MFMA order (4 accumulators in rotation)
            float4 a[2][MI], b[2][NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[0][i] = *reinterpret_cast<const float4*>(&As[buf][arow + i * 32 * LDK]);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[0][j] = *reinterpret_cast<const float4*>(&Bs[buf][brow + j * 32 * LDK]);
#pragma unroll
            for (int kc = 0; kc < BK / 8; ++kc) {
                const int cur = kc & 1, nxt = cur ^ 1;
                if (kc + 1 < BK / 8) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) a[nxt][i] = *reinterpret_cast<const float4*>(&As[buf][arow + i * 32 * LDK + (kc + 1) * 8]);
#pragma unroll
                    for (int j = 0; j < NI; ++j) b[nxt][j] = *reinterpret_cast<const float4*>(&Bs[buf][brow + j * 32 * LDK + (kc + 1) * 8]);
                }
#define RP_MFMA_A(C_) \
                _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) \
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[cur][i].C_), "v"(b[cur][j].C_));
                RP_MFMA_A(x) RP_MFMA_A(y) RP_MFMA_A(z) RP_MFMA_A(w)
#undef RP_MFMA_A
            }
        }
#else
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            float4 a[MI], b[NI];
#if RP_ABLATE == 3 || RP_ABLATE == 5
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = make_float4(1.f + kt, 2.f, 3.f, 4.f + lane);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = make_float4(0.5f, 0.25f + kt, 0.125f, 1.f);
#else
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const float4*>(&As[buf][arow + i * 32 * LDK + kc * 8]);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const float4*>(&Bs[buf][brow + j * 32 * LDK + kc * 8]);
#endif
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
#if RP_ABLATE == 1
                    acc[i][j][0] += a[i].x * b[j].x + a[i].y * b[j].y + a[i].z * b[j].z + a[i].w * b[j].w;
#else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
#endif
                }
        }
#endif
#if RP_ABLATE != 5
        if (NBUF == 1) __syncthreads();                 // single buffer: every wave is done reading tile kt
        if (kt + 1 < nkt) RP_STORE_TILE(NBUF == 2 ? (buf ^ 1) : 0)
#endif
#if RP_ABLATE != 4 && RP_ABLATE != 5
        __syncthreads();
#endif
    }
#undef RP_ISSUE_LOADS
#undef RP_STORE_TILE
    if constexpr (SPLIT == 2 || SPLIT == 3) {          // undo the power-of-two weight pre-scale (exact)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= d.wscale;
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (d.ksplit > 1) {                                // partial sums, reduced in fixed order by splitk_reduce_kernel
        float* po = d.partial + (size_t)ks * d.M * d.cout_pad;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * MI * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= d.M) continue;
#pragma unroll
                for (int j = 0; j < NI; ++j) rp_stg(po + (size_t)m * d.cout_pad + n0 + wn * NI * 32 + j * 32 + (lane & 31), acc[i][j][r]);
            }
        return;
    }
    if (d.stat_part) {
        // Fused BatchNorm statistics: float64 sum / sum of squares of this tile's raw outputs per column and
        // per group slot (a tile spans at most two groups here), reduced in a fixed order:
        // lane rows -> lane pair (xor 32) -> the WM waves of a column (LDS, in order) -> one record per tile.
        double* red = reinterpret_cast<double*>(&As[0][0]);       // main loop is over (it ended with a barrier)
        double s0[NI], q0s[NI], s1[NI], q1s[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) { s0[j] = 0; q0s[j] = 0; s1[j] = 0; q1s[j] = 0; }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // UNI: every row of the tile is in slot 0 (rows past M hold exact zeros: their A rows are masked)
                const int sl = UNI ? 0 : rowslot[wm * MI * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const double v = (double)acc[i][j][r];
                    if (UNI) { s0[j] += v; q0s[j] += v * v; }
                    else {                                  // branch-free: the other slot adds +0
                        const double v0 = sl == 0 ? v : 0.0, v1 = sl == 1 ? v : 0.0;
                        s0[j] += v0; q0s[j] += v0 * v0; s1[j] += v1; q1s[j] += v1 * v1;
                    }
                }
            }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            s0[j] += rp_shfl_xor_d(s0[j], 32); q0s[j] += rp_shfl_xor_d(q0s[j], 32);
            if (!UNI) { s1[j] += rp_shfl_xor_d(s1[j], 32); q1s[j] += rp_shfl_xor_d(q1s[j], 32); }
            if (lane < 32) {
                const int cl = wn * NI * 32 + j * 32 + lane;
                red[((0 * WM + wm) * BN + cl) * 2 + 0] = s0[j]; red[((0 * WM + wm) * BN + cl) * 2 + 1] = q0s[j];
                red[((1 * WM + wm) * BN + cl) * 2 + 0] = s1[j]; red[((1 * WM + wm) * BN + cl) * 2 + 1] = q1s[j];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * BN; idx += NT) {
            const int sl = idx / BN, cl = idx - sl * BN;
            double a = 0, b = 0;
#pragma unroll
            for (int w = 0; w < WM; ++w) { a += red[((sl * WM + w) * BN + cl) * 2]; b += red[((sl * WM + w) * BN + cl) * 2 + 1]; }
            double* o = d.stat_part + (((size_t)mtile * 2 + sl) * d.cout_pad + n0 + cl) * 2;
            rp_stg(o, a); rp_stg(o + 1, b);
        }
    }
    // output rows of this lane come in runs of 4 (r & 3): one 16-byte LDS read of their pixel indices per run
    const int col0 = n0 + wn * NI * 32 + (lane & 31);
    if (!d.bias && !d.tanh_out) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int4 px = *reinterpret_cast<const int4*>(&rowpix[wm * MI * 32 + i * 32 + 8 * r4 + 4 * (lane >> 5)]);
                const int pxs[4] = {px.x, px.y, px.z, px.w};
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    if (pxs[rr] < 0) continue;
                    float* yo = d.y + (size_t)pxs[rr] * d.ycstride + d.ychoff + col0;
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        if (col0 + j * 32 < d.Cout) rp_stg(yo + j * 32, acc[i][j][r4 * 4 + rr]);
                }
            }
        return;
    }
    float bias_v[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) bias_v[j] = (d.bias && col0 + j * 32 < d.Cout) ? rp_ldg(d.bias + col0 + j * 32) : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pix = rowpix[wm * MI * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
            if (pix < 0) continue;
            float* yo = d.y + (size_t)pix * d.ycstride + d.ychoff + col0;
#pragma unroll
            for (int j = 0; j < NI; ++j)
                if (col0 + j * 32 < d.Cout) {
                    float v = acc[i][j][r] + bias_v[j];
                    if (d.tanh_out) v = tanhf(v);
                    rp_stg(yo + j * 32, v);
                }
        }
}

// ---- 16-bit operand modes of the tile kernels (SPLIT 2 = f16x3, 3 = f16; the split modes of conv_igemm_kernel) -------------------------
// A staged LDS row keeps the 144-byte stride of the fp32 tiles: [32 x fp16 hi | 32 x fp16 lo | pad] -- the geometry (row indices, tap
// offsets, conflict-free b128 fragment reads) is unchanged; the values are converted ONCE, when the tile is staged, and every
// (tap, 16-channel step) of a chunk is one v_mfma_f32_32x32x16_f16 per product term (3 terms, or 1 for plain f16) with the fragment
// read straight from LDS.  The packed weights of these modes are already in the same [hi | lo] k-tile format (finalize).
template <int SPLIT>
__device__ __forceinline__ void rp_tile_store_a(float* row, int kq, rp_v2f v01, rp_v2f v23) {
    if constexpr (SPLIT == 0) *reinterpret_cast<float4*>(row + kq * 4) = make_float4(v01.x, v01.y, v23.x, v23.y);
    else {
        typedef typename SplitT<SPLIT>::v4 h4_;
        const rp_f4v vf_ = {v01.x, v01.y, v23.x, v23.y};
        const h4_ hi_ = __builtin_convertvector(vf_, h4_);
        h4_* ar_ = reinterpret_cast<h4_*>(row);
        ar_[kq] = hi_;
        if constexpr (SPLIT >= 4) {       // three bf16 pieces: both remainders are exact fp32 differences, the third piece is exact
            const rp_f4v r1_ = vf_ - __builtin_convertvector(hi_, rp_f4v);
            const h4_ mid_ = __builtin_convertvector(r1_, h4_);
            ar_[8 + kq] = mid_;
            ar_[16 + kq] = __builtin_convertvector(r1_ - __builtin_convertvector(mid_, rp_f4v), h4_);
        } else if (SPLIT != 3) ar_[8 + kq] = __builtin_convertvector(vf_ - __builtin_convertvector(hi_, rp_f4v), h4_);
    }
}
// one 32-channel chunk of one tap for MI x NI accumulators: a_rows[i] / b_rows[j] are LDS float indices of the lane's fragment rows
template <int SPLIT, int MI, int NI>
__device__ __forceinline__ void rp_tile_mma(floatx16 (&acc)[MI][NI], const float* At, const int (&a_rows)[MI], const float* Bt, const int (&b_rows)[NI]) {
    if constexpr (SPLIT == 0) {
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            float4 a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const float4*>(&At[a_rows[i] + kc * 8]);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const float4*>(&Bt[b_rows[j] + kc * 8]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    } else rp_split_mma<SPLIT, MI, NI>(acc, At, a_rows, Bt, b_rows);
}

// ---- stride-2 4x4 transposed conv, the four sub-pixel phases of a spatial patch in ONE workgroup -------------------------
// conv_igemm_kernel treats every (phase, tap) of a transposed conv as its own k-tile: a 32-channel chunk of the input is
// gathered from global memory, BatchNorm-transformed and stored to LDS 16 times per output patch (4 phases x 2x2 taps), which
// is what holds deconv2 at 96-102 TFLOP/s (the MFMA-only ablation of those launches runs 134).  All 16 (phase, tap)
// products of a patch read the same 3x3 neighbourhood, so here the (PR + 2) x 18 halo tile of a chunk is fetched and
// transformed ONCE, out-of-image pixels stored as zeros, and every tap of every phase is an LDS offset into it:
//   patch = PR x 16 input pixels of one image (PR = 8 MI), wave w owns patch rows 2 MI w .. 2 MI (w + 1) - 1, an MFMA tile
//   of 32 rows = 2 patch rows x 16 columns; accumulators acc[phase][MI][NI] (128 VGPRs);
//   per 32-channel chunk: stage the halo tile, then per phase: stage the phase's 4 weight tiles [tap][32 NI cols][32 k] and
//   run 4 taps x 16 k-steps x MI x NI MFMAs between barriers (conv_igemm_kernel: MI x NI x 16 per barrier pair).
// Global A loads and transforms per output: 16 x 256 rows -> (PR + 2) x 18 rows per chunk (12.6x / 11.4x fewer).
// descs: 4 consecutive phase descriptors per head (blockIdx.y), as built for conv_igemm_kernel; the BatchNorm records go to
// each phase's stat_part with one record per patch (stat_bm = PR * 16, slot 0: a patch lies in one image).
// NW waves per workgroup; an MFMA tile of 32 rows = (32 / TC) patch rows x TC columns.  TC == 16: the NW x MI tiles are stacked
// vertically (patch 2 MI NW x 16); TC == 8: side by side (patch 4 x 8 NW, MI == 1: a full-width strip of a 56-wide grid with NW = 7).
// blockIdx.z = N tile of NI * 32 output columns (Cout 64 as two tiles of 32: three workgroups per CU instead of two).
// PAIR (TC == 8, 4 waves): the workgroup takes TWO 8 x 8 patches (consecutive in patch order, possibly in the two images of one
// BatchNorm group) with separate 10 x 10 halo tiles -- 56-wide grids tile into 8 x 8 but not into 8 x 16 (deconv3).
#ifndef RP_DT_SPLIT_OCC
#define RP_DT_SPLIT_OCC 2
#endif
// (16-bit modes of the paired 8 x 8 variant -- deconv3 -- spill 44 registers at the 168 of 3 workgroups per CU: 2 per CU there, -9 %)
#define RP_DT_OCC(MI_, NI_, NW_, SP_, PAIR_) ((SP_) >= 4 ? 2 : ((MI_) * (NI_) == 1 ? ((NW_) == 4 ? (((SP_) && (PAIR_)) ? RP_DT_SPLIT_OCC : 3) : 2) : 2))   /* three-piece rows: 68 KB of LDS */
template <int MI, int NI, int NW, int TC, bool PAIR = false, int SPLIT = 0>
__global__ __launch_bounds__(NW * 64, RP_DT_OCC(MI, NI, NW, SPLIT, PAIR)) void deconv_tile_kernel(const ConvDesc* __restrict__ descs) {
    constexpr int NT = NW * 64, TR = 32 / TC;
    constexpr int LD = RowLd<SPLIT>::v;                               // LDS row stride (floats)
    constexpr bool S3 = SPLIT >= 4;                                   // three-piece bf16 operands: 6 bytes per weight, rows [hi | mid | lo]
    constexpr int WB = S3 ? 6 : 4;
    constexpr int PR = PAIR ? 8 : (TC == 16 ? TR * MI * NW : TR), PW = PAIR ? 8 : (TC == 16 ? 16 : TC * NW), HW2 = PW + 2;
    constexpr int SUBPIX = (PR + 2) * HW2, NPIX = (PAIR ? 2 : 1) * SUBPIX;
    static_assert(TC == 16 || MI == 1, "strip layout: one tile per wave");
    static_assert(!PAIR || (TC == 8 && NW == 4 && MI == 1), "paired 8 x 8 patches: 4 waves of one 4 x 8 tile");
    constexpr int PSTEP = NT / KQ;                                    // halo pixels covered per slot iteration
    constexpr int A_SLOTS = (NPIX + PSTEP - 1) / PSTEP;               // float4 slots per thread for the halo tile
    constexpr int B_ROWS = 4 * NI * 32;                               // rows [tap][col] of a phase's weight tiles
    constexpr int B_SLOTS = (B_ROWS + PSTEP - 1) / PSTEP;
    static_assert(BK == 32, "one 128-byte line per pixel and chunk");
    __shared__ __attribute__((aligned(16))) float At[NPIX * LD];
    __shared__ __attribute__((aligned(16))) float Bt[4 * NI * 32 * LD];
    __shared__ __attribute__((aligned(16))) float sstab[2 * 512];   // per 4 channels: 4 scales, then 4 shifts
    const ConvDesc* dh = descs + blockIdx.y * 4;
    const ConvDesc d = dh[0];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int ppx = d.Win / PW, ppi = (d.Hin / PR) * ppx;
    int imgs[2], y0s[2], x0s[2];                                      // (PAIR: per sub-patch; block-uniform)
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
        const int q = PAIR ? 2 * blockIdx.x + sp : blockIdx.x;
        imgs[sp] = q / ppi;
        const int prem = q - imgs[sp] * ppi;
        y0s[sp] = (prem / ppx) * PR; x0s[sp] = (prem % ppx) * PW;
    }
    const int g = imgs[0] >> 1;
    const int img_base = PAIR ? 2 * g : imgs[0];                      // image the buffer descriptors start at
    const int n0 = blockIdx.z * NI * 32;
    for (int c = tid; c < d.Cin; c += NT) {
        const bool s1 = c >= d.src[0].C;
        const float2 e = rp_ldg2(reinterpret_cast<const float*>(s1 ? d.src[1].ss + (size_t)g * d.src[1].sstride + (c - d.src[0].C)
                                                                   : d.src[0].ss + (size_t)g * d.src[0].sstride + c));
        sstab[(c & ~3) * 2 + (c & 3)] = e.x; sstab[(c & ~3) * 2 + 4 + (c & 3)] = e.y;
    }
    // halo slots of this thread: pixel offset in the image (or -1: zero padding) and LDS position
    // slot it of this thread = halo pixel tid / 8 + PSTEP it, chunk column kqa (NT % KQ == 0); LDS position = a_lds0 + it * PSTEP * LD
    const int kqa = tid % KQ;
    const int a_lds0 = (tid / KQ) * LD + kqa * 4;
    // Buffer loads (SGPR descriptor + 32-bit lane offset + SGPR chunk offset): one index register per slot instead of a 64-bit
    // address pair -- with 128 accumulator and 44 + 16 prefetch registers the flat-address version spilled.
    int a_px[A_SLOTS];                                                // pixel index inside the image, or -1 (zero padding)
#pragma unroll
    for (int it = 0; it < A_SLOTS; ++it) {
        const int pix = tid / KQ + it * PSTEP;
        const int sp = (PAIR && pix >= SUBPIX) ? 1 : 0, lp = pix - sp * SUBPIX;
        const int hy = lp / HW2, hx = lp - hy * HW2;
        const int iy = (sp ? y0s[1] : y0s[0]) - 1 + hy, ix = (sp ? x0s[1] : x0s[0]) - 1 + hx;
        const bool ok = (pix < NPIX) && (iy >= 0) && (iy < d.Hin) && (ix >= 0) && (ix < d.Win);
        a_px[it] = ok ? (((sp ? imgs[1] : imgs[0]) - img_base) * d.Hin + iy) * d.Win + ix : -1;
    }
    const size_t img_px = (size_t)img_base * d.Hin * d.Win;
    const int a_bytes = (PAIR ? 2 : 1) * d.Hin * d.Win * 4;
    const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)(d.src[0].x + img_px * d.src[0].cstride), 0,
                                                                          a_bytes * d.src[0].cstride, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a1 = d.nsrc > 1 ? __builtin_amdgcn_make_buffer_rsrc((void*)(d.src[1].x + img_px * d.src[1].cstride), 0,
                                                                                       a_bytes * d.src[1].cstride, 0x00020000) : rs_a0;
    __amdgpu_buffer_rsrc_t rs_b[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) rs_b[p] = __builtin_amdgcn_make_buffer_rsrc((void*)dh[p].w, 0, d.cout_pad * d.K * WB, 0x00020000);
    // weight slot it: row r = tid / 8 + PSTEP it of [tap][col] (r = tap * NI * 32 + col), chunk column kqa; LDS position r * LD + kqa * 4
    int b_off[B_SLOTS];                                               // byte offset of the row in the phase's weights (chunk 0), or -1
#pragma unroll
    for (int it = 0; it < B_SLOTS; ++it) {
        const int r = tid / KQ + it * PSTEP;
        const int tap = r / (NI * 32), col = r - tap * (NI * 32);
        b_off[it] = r < B_ROWS ? ((n0 + col) * d.K + tap * d.Cin) * WB + kqa * 16 : -1;     // (S3: the row's lo piece at + 128 - 8 kqa)
    }
    const int b_lds0 = (tid / KQ) * LD + kqa * 4;
    floatx16 acc[4][MI][NI];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][i][j][r] = 0.f;
    // A fragment base of MFMA tile i of this wave: patch row / column of MFMA row l31, halo origin (+1, +1)
    const int try0 = PAIR ? TR * (wave & 1) : (TC == 16 ? TR * MI * wave : 0), tcx0 = (PAIR || TC == 16) ? 0 : TC * wave;   // first patch row / column of the wave's tiles
    const int wsp = PAIR ? (wave >> 1) : 0;                           // sub-patch of this wave
    int arow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) arow[i] = (wsp * SUBPIX + (try0 + TR * i + l31 / TC + 1) * HW2 + tcx0 + (l31 % TC) + 1) * LD + h * 4;
    const int brow = l31 * LD + h * 4;
    const float slope = d.src[0].slope;
    int aoffs[4][4];                                                  // block-uniform: LDS offset of (phase, tap) relative to the output pixel
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int t = 0; t < 4; ++t) aoffs[p][t] = ((int)dh[p].offy[t] * HW2 + (int)dh[p].offx[t]) * LD;
    float4 ra[A_SLOTS], rb[B_SLOTS];
    float2 rbl[S3 ? B_SLOTS : 1];                                     // (S3) the lo pieces of the weight rows
    const int nchunk = d.Cin / BK;
    // chunk order: the skip source's channels (src[1]) first, then src[0]'s -- the k-th chunk processed starts at channel c0_of(k)
    const int nch0 = d.src[0].C / BK, nch1 = nchunk - nch0;
    auto c0_of = [&](int k) { return (k < nch1 ? nch0 + k : k - nch1) * BK; };
    // (the <1, 2> variant serves the heads without a skip source -- deconv2 s / f --: no snapshot code there, it cost 2 spilled registers)
    constexpr bool SNAP = !(MI == 1 && NI == 2);
    const int snap_mode = SNAP ? d.snap_mode : 0;
    const int k_first = (snap_mode == 2) ? nch1 : 0;
    // accumulator snapshot (ConvDesc::snap, one region per head): float4 q of accumulator (p, i, j) of thread tid of this workgroup at
    // float4 index ((((p MI + i) NI + j) 4 + q) NT + tid: buffer accesses with ONE lane offset register and the rest as the scalar offset
    // (64-bit addresses per access cost the tile kernels up to 54 spilled registers)
    const __amdgpu_buffer_rsrc_t rs_snap = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(snap_mode ? d.snap + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * (size_t)(4 * MI * NI * 16) * NT : nullptr), 0,
        snap_mode ? 4 * MI * NI * 16 * NT * 4 : 0, 0x00020000);
    auto snap_load = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = rp_bufld4(rs_snap, tid * 16, ((((p * MI + i) * NI + j) * 4 + q) * NT) * 16);
                        acc[p][i][j][4 * q] = v.x; acc[p][i][j][4 * q + 1] = v.y; acc[p][i][j][4 * q + 2] = v.z; acc[p][i][j][4 * q + 3] = v.w;
                    }
    };
    auto snap_store = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const rp_f32x4g v = {acc[p][i][j][4 * q], acc[p][i][j][4 * q + 1], acc[p][i][j][4 * q + 2], acc[p][i][j][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128((__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned)v, rs_snap, tid * 16,
                                                               ((((p * MI + i) * NI + j) * 4 + q) * NT) * 16, 0);
                    }
    };

#define RP_DT_LOAD_A(C0)                                                                                       \
    {                                                                                                         \
        const bool s1_ = (d.nsrc > 1) && ((C0) >= d.src[0].C);                                                \
        const int scs4_ = (s1_ ? d.src[1].cstride : d.src[0].cstride) * 4;                                   \
        const int soff_ = ((C0) - (s1_ ? d.src[0].C : 0)) * 4;                                                \
        _Pragma("unroll") for (int it = 0; it < A_SLOTS; ++it) {                                              \
            int px_ = a_px[it];                                                                               \
            asm volatile("" : "+v"(px_));      /* keep the offset arithmetic here: hoisted, it costs 22 registers */ \
            const int voff_ = max(px_, 0) * scs4_ + kqa * 16;                                                 \
            ra[it] = s1_ ? rp_bufld4(rs_a1, voff_, soff_) : rp_bufld4(rs_a0, voff_, soff_);                    \
        }                                                                                                     \
    }
#define RP_DT_STORE_A(C0)                                                                                      \
    {                                                                                                         \
        const float4* q = reinterpret_cast<const float4*>(&sstab[2 * ((C0) + kqa * 4)]);                      \
        const float4 s0_ = q[0], s1v_ = q[1];                                                                 \
        const rp_v2f sl2_ = {slope, slope};                                                                   \
        _Pragma("unroll") for (int it = 0; it < A_SLOTS; ++it) {                                              \
            rp_v2f v01 = {ra[it].x, ra[it].y}, v23 = {ra[it].z, ra[it].w};                                    \
            v01 = v01 * (rp_v2f){s0_.x, s0_.y} + (rp_v2f){s1v_.x, s1v_.y};                                    \
            v23 = v23 * (rp_v2f){s0_.z, s0_.w} + (rp_v2f){s1v_.z, s1v_.w};                                    \
            const rp_v2f t01 = v01 * sl2_, t23 = v23 * sl2_;                                                  \
            const float okf_ = a_px[it] >= 0 ? 1.f : 0.f;                                                     \
            const rp_v2f mk_ = {okf_, okf_};                                                                  \
            v01 = (rp_v2f){fmaxf(v01.x, t01.x), fmaxf(v01.y, t01.y)} * mk_;                                   \
            v23 = (rp_v2f){fmaxf(v23.x, t23.x), fmaxf(v23.y, t23.y)} * mk_;                                   \
            if ((it + 1) * PSTEP <= NPIX || tid / KQ + it * PSTEP < NPIX)                                     \
                rp_tile_store_a<SPLIT>(&At[a_lds0 - kqa * 4 + it * PSTEP * LD], kqa, v01, v23);              \
        }                                                                                                     \
    }
#define RP_DT_LOAD_B(P, C0) RP_DT_LOAD_B2(rb, rbl, P, C0)
#define RP_DT_STORE_B() RP_DT_STORE_B2(rb, rbl)

#define RP_DT_LOAD_B2(RB, RBL, P, C0)                                                                          \
    {                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < B_SLOTS; ++it) {                                              \
            RB[it] = rp_bufld4(rs_b[P], max(b_off[it], 0), (C0) * WB);                                         \
            if (S3) RBL[S3 ? it : 0] = rp_bufld2(rs_b[P], max(b_off[it], 0) + 128 - kqa * 8, (C0) * WB);       \
        }                                                                                                     \
    }
#define RP_DT_STORE_B2(RB, RBL)                                                                                \
    {                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < B_SLOTS; ++it)                                                \
            if ((it + 1) * PSTEP <= B_ROWS || b_off[it] >= 0) {                                               \
                *reinterpret_cast<float4*>(&Bt[b_lds0 + it * PSTEP * LD]) = RB[it];                           \
                if (S3) *reinterpret_cast<float2*>(&Bt[b_lds0 + 32 - kqa * 2 + it * PSTEP * LD]) = RBL[S3 ? it : 0]; \
            }                                                                                                 \
    }
    if constexpr (SPLIT != 0 && SPLIT < 4 && PAIR) {       // (three-piece modes: a phase is 48-72 MFMAs, the plain one-phase-ahead prefetch below is deep enough and needs one weight register set)
                                              // (deconv3's variant, which has the registers for it at 2 workgroups per CU; <1, 1> 8 x 16: no gain,
                                              // <1, 2>: two weight sets + 128 accumulators spill, 896 -> 1415 us)
        // 16-bit modes: a phase is 24 MFMAs of 32 cycles (fp32: 64 of 64), shorter than a trip to L2, so the register prefetch runs
        // deeper: the weights of phase q + 2 are requested at the top of phase q (two register sets, alternating by phase parity) and
        // the next chunk's halo tile at the top of phase 1 (fp32: one phase ahead for both)
        float4 rb2[B_SLOTS];
        float2 rbl2[S3 ? B_SLOTS : 1];
        if (snap_mode == 2) snap_load();
        RP_DT_LOAD_A(c0_of(k_first))
        RP_DT_LOAD_B2(rb, rbl, 0, c0_of(k_first))
        RP_DT_LOAD_B2(rb2, rbl2, 1, c0_of(k_first))
        __syncthreads();                                              // sstab
        RP_DT_STORE_A(c0_of(k_first))
        RP_DT_STORE_B2(rb, rbl)
        __syncthreads();
        for (int ch = k_first; ch < nchunk; ++ch) {
            const int c0 = c0_of(ch), c0n = c0_of(ch + 1 < nchunk ? ch + 1 : ch);
            const bool last = (ch + 1 == nchunk);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // the set that was stored to LDS for THIS phase is free: phase p + 2's weights go there
                if (p < 2) { if (p == 0) RP_DT_LOAD_B2(rb, rbl, 2, c0) else RP_DT_LOAD_B2(rb2, rbl2, 3, c0) }
                else if (!last) { if (p == 2) RP_DT_LOAD_B2(rb, rbl, 0, c0n) else RP_DT_LOAD_B2(rb2, rbl2, 1, c0n) }
                if (p == 1 && !last) RP_DT_LOAD_A(c0n)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int aoff = aoffs[p][t];
                    int ar_[MI], br_[NI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) ar_[i] = arow[i] + aoff;
#pragma unroll
                    for (int j = 0; j < NI; ++j) br_[j] = brow + (t * NI + j) * 32 * LD;
                    rp_tile_mma<SPLIT, MI, NI>(acc[p], At, ar_, Bt, br_);
                }
                __syncthreads();                                      // every wave is done with this phase's weights (and, p == 3, the halo tile)
                if (p < 3 || !last) { if (p & 1) RP_DT_STORE_B2(rb, rbl) else RP_DT_STORE_B2(rb2, rbl2) }      // phase p + 1's weights
                if (p == 3 && !last) RP_DT_STORE_A(c0n)
                __syncthreads();
            }
            if (snap_mode == 1 && ch + 1 == nch1) snap_store();     // the skip source is done: the accumulators for the self-cached forwards
        }
    } else {
#if RP_TILE_TIMING
        // experiment (tools/build_variant.py tt -DRP_TILE_TIMING=1): s_memtime stamps around the segments of the main loop, summed over the waves of
        // the launch in g_tile_timing: [0] load issue, [1] MFMAs + fragment reads, [2] first barrier, [3] LDS stores (+ A transform), [4] second barrier,
        // [5] prologue, [6] whole kernel, [7] waves
        unsigned long long tt_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned long long tt_start_ = __builtin_readcyclecounter();
        unsigned long long tt_prev_ = tt_start_;
#define RP_TT(i_) { const unsigned long long n_ = __builtin_readcyclecounter(); tt_[i_] += n_ - tt_prev_; tt_prev_ = n_; }
#else
#define RP_TT(i_)
#endif
        if (snap_mode == 2) snap_load();        // the accumulators as the full forward left them after the skip source's chunks
        RP_DT_LOAD_A(c0_of(k_first))
        RP_DT_LOAD_B(0, c0_of(k_first))
        __syncthreads();                                                  // sstab
        RP_DT_STORE_A(c0_of(k_first))
        RP_DT_STORE_B()
        __syncthreads();
        RP_TT(5)
        for (int ch = k_first; ch < nchunk; ++ch) {
            const int c0 = c0_of(ch), c0n = c0_of(ch + 1 < nchunk ? ch + 1 : ch);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // prefetch into registers: the next phase's weights, and (during the last phase) the next chunk's halo tile
                const bool last = (ch + 1 == nchunk);
                if (p < 3) RP_DT_LOAD_B(p + 1, c0)
                else if (!last) { RP_DT_LOAD_B(0, c0n) RP_DT_LOAD_A(c0n) }
                RP_TT(0)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int aoff = aoffs[p][t];
                    int ar_[MI], br_[NI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) ar_[i] = arow[i] + aoff;
#pragma unroll
                    for (int j = 0; j < NI; ++j) br_[j] = brow + (t * NI + j) * 32 * LD;
                    rp_tile_mma<SPLIT, MI, NI>(acc[p], At, ar_, Bt, br_);
                }
                RP_TT(1)
                __syncthreads();                                          // every wave is done with this phase's weights (and, p == 3, the halo tile)
                RP_TT(2)
                if (p < 3 || !last) RP_DT_STORE_B()
                if (p == 3 && !last) RP_DT_STORE_A(c0n)
                RP_TT(3)
                __syncthreads();
                RP_TT(4)
            }
            if (snap_mode == 1 && ch + 1 == nch1) snap_store();         // the skip source is done: the accumulators for the self-cached forwards
        }
#if RP_TILE_TIMING
        tt_[6] = __builtin_readcyclecounter() - tt_start_; tt_[7] = 1;
        if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_tile_timing[i], tt_[i]);
#endif
#undef RP_TT
    }
#undef RP_DT_LOAD_B2
#undef RP_DT_STORE_B2
#undef RP_DT_LOAD_A
#undef RP_DT_STORE_A
#undef RP_DT_LOAD_B
#undef RP_DT_STORE_B

    // epilogue per phase: BatchNorm record of the patch, then the strided NHWC stores
    double* red = reinterpret_cast<double*>(&At[0]);                  // [NW waves][NI * 32][2]
    const int patch = blockIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if constexpr (SPLIT == 2 || SPLIT == 3) {          // undo the power-of-two weight pre-scale of the phase (exact)
            const float wsc = dh[p].wscale;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[p][i][j][r] *= wsc;
        }
        if (dh[p].stat_part) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                double sm = 0.0, sq = 0.0;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const double v = (double)acc[p][i][j][r]; sm += v; sq += v * v; }
                sm += rp_shfl_xor_d(sm, 32); sq += rp_shfl_xor_d(sq, 32);
                if (h == 0) { red[((wave * NI + j) * 32 + l31) * 2] = sm; red[((wave * NI + j) * 32 + l31) * 2 + 1] = sq; }
            }
            __syncthreads();
            if (tid < NI * 32) {
                double a = 0, b = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) { a += red[((w * NI) * 32 + tid) * 2]; b += red[((w * NI) * 32 + tid) * 2 + 1]; }
                double* o = dh[p].stat_part + (((size_t)patch * 2) * d.cout_pad + n0 + tid) * 2;
                rp_stg(o, a); rp_stg(o + 1, b);
            }
            __syncthreads();
        }
        const int opy = dh[p].py, opx = dh[p].px;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;         // row of the 32-row MFMA tile
                const int ry = try0 + TR * i + rl / TC, cx = tcx0 + rl % TC;
                const size_t pix = ((size_t)(wsp ? imgs[1] : imgs[0]) * d.Hout + 2 * ((wsp ? y0s[1] : y0s[0]) + ry) + opy) * d.Wout + 2 * ((wsp ? x0s[1] : x0s[0]) + cx) + opx;
                float* yo = d.y + pix * d.ycstride + d.ychoff + n0 + l31;
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    if (n0 + j * 32 + l31 < d.Cout) rp_stg(yo + j * 32, acc[p][i][j][r]);
            }
    }
}

// ---- stride-2 4x4 convs (conv2, conv3) on parity planes ---------------------------------------------------------------------
// out(y, x) = sum_{ky, kx} in(2y - 1 + ky, 2x - 1 + kx) w(ky, kx): the four taps with ky = p ? {0, 2} : {1, 3} and kx = q ? {0, 2} : {1, 3}
// read the parity plane (p, q) of the input, in(2a + p, 2b + q), at a in {y - p, y - p + 1}: a 2x2 stride-1 conv per plane.  Per
// 32-channel chunk and plane the (PR + 1) x (PW + 1) plane pixels of an output patch are gathered, BatchNorm-transformed and
// stored to LDS ONCE and the plane's 4 taps are LDS offsets {0, 1} x {0, 1} into that tile (conv_igemm_kernel gathers the PR x PW rows
// of every one of the 16 taps: 3.3-3.5x the global loads and transforms).  The weight tile of one tap is staged per k-step as before.
// Geometry as in deconv_tile_kernel: TC == 16: patch 2 MI NW x 16 outputs, tiles stacked; PAIR: two 8 x 8 patches (56-wide grids).
template <int MI, int NI, int TC, bool PAIR, int SPLIT = 0>
__global__ __launch_bounds__(256, SPLIT >= 4 ? 2 : (MI * NI <= 2 ? 4 : 3)) void conv_s2_tile_kernel(const ConvDesc* __restrict__ descs) {
    constexpr int NW = 4, NT = 256, TR = 32 / TC;
    constexpr int LD = RowLd<SPLIT>::v;                               // LDS row stride (floats)
    constexpr bool S3 = SPLIT >= 4;                                   // three-piece bf16 operands: 6 bytes per weight, rows [hi | mid | lo]
    constexpr int WB = S3 ? 6 : 4;
    constexpr int PR = PAIR ? 8 : TR * MI * NW, PW = PAIR ? 8 : 16, HW1 = PW + 1;
    constexpr int SUBPIX = (PR + 1) * HW1, NPIX = (PAIR ? 2 : 1) * SUBPIX;
    static_assert(!PAIR || (TC == 8 && MI == 1), "paired 8 x 8 patches: 4 waves of one 4 x 8 tile");
    constexpr int PSTEP = NT / KQ;
    constexpr int A_SLOTS = (NPIX + PSTEP - 1) / PSTEP;
    constexpr int B_ROWS = NI * 32;
    constexpr int B_SLOTS = (B_ROWS + PSTEP - 1) / PSTEP;
    static_assert(BK == 32 && B_ROWS % PSTEP == 0, "weight tile layout");
    __shared__ __attribute__((aligned(16))) float At[NPIX * LD];
    __shared__ __attribute__((aligned(16))) float Bt[B_ROWS * LD];
    __shared__ __attribute__((aligned(16))) float sstab[2 * 128];
    const ConvDesc d = descs[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int ppx = d.Wp / PW, ppi = (d.Hp / PR) * ppx;
    int imgs[2], y0s[2], x0s[2];
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
        const int q = PAIR ? 2 * blockIdx.x + sp : blockIdx.x;
        imgs[sp] = q / ppi;
        const int prem = q - imgs[sp] * ppi;
        y0s[sp] = (prem / ppx) * PR; x0s[sp] = (prem % ppx) * PW;
    }
    const int g = imgs[0] >> 1;
    const int img_base = PAIR ? 2 * g : imgs[0];
    for (int c = tid; c < d.Cin; c += NT) {
        const float2 e = rp_ldg2(reinterpret_cast<const float*>(d.src[0].ss + (size_t)g * d.src[0].sstride + c));
        sstab[(c & ~3) * 2 + (c & 3)] = e.x; sstab[(c & ~3) * 2 + 4 + (c & 3)] = e.y;
    }
    const int kqa = tid % KQ;
    const int a_lds0 = (tid / KQ) * LD + kqa * 4;
    // halo slot it = plane-tile pixel tid / 8 + PSTEP it: input pixel (2 (Y0 + r) - p, 2 (X0 + c) - q) of plane (p, q); a_in[it] is the
    // (p, q) = (0, 0) pixel index relative to image img_base, a_edge[it] flags the slots that leave the image for p = 1 / p = 0 / q = 1 / q = 0
    int a_in[A_SLOTS], a_edge = 0;
#pragma unroll
    for (int it = 0; it < A_SLOTS; ++it) {
        const int pix = tid / KQ + it * PSTEP;
        const int sp = (PAIR && pix >= SUBPIX) ? 1 : 0, lp = pix - sp * SUBPIX;
        const int r = lp / HW1, c = lp - r * HW1;
        const int iy = 2 * ((sp ? y0s[1] : y0s[0]) + r), ix = 2 * ((sp ? x0s[1] : x0s[0]) + c);
        a_in[it] = (((sp ? imgs[1] : imgs[0]) - img_base) * d.Hin + iy) * d.Win + ix;
        // p = 1 needs iy - 1 >= 0; p = 0 needs iy < Hin; same for columns; slots past the tile are always invalid
        const int e = (pix >= NPIX) ? 15 : ((iy - 1 < 0 ? 1 : 0) | (iy >= d.Hin ? 2 : 0) | (ix - 1 < 0 ? 4 : 0) | (ix >= d.Win ? 8 : 0));
        if (it < 8) a_edge |= e << (4 * it);                          // (slots 8.. : second word below)
    }
    int a_edge2 = 0;
#pragma unroll
    for (int it = 8; it < A_SLOTS; ++it) {
        const int pix = tid / KQ + it * PSTEP;
        const int sp = (PAIR && pix >= SUBPIX) ? 1 : 0, lp = pix - sp * SUBPIX;
        const int r = lp / HW1, c = lp - r * HW1;
        const int iy = 2 * ((sp ? y0s[1] : y0s[0]) + r), ix = 2 * ((sp ? x0s[1] : x0s[0]) + c);
        const int e = (pix >= NPIX) ? 15 : ((iy - 1 < 0 ? 1 : 0) | (iy >= d.Hin ? 2 : 0) | (ix - 1 < 0 ? 4 : 0) | (ix >= d.Win ? 8 : 0));
        a_edge2 |= e << (4 * (it - 8));
    }
    const size_t img_px = (size_t)img_base * d.Hin * d.Win;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(d.src[0].x + img_px * d.src[0].cstride), 0,
                                                                         (PAIR ? 2 : 1) * d.Hin * d.Win * 4 * d.src[0].cstride, 0x00020000);
    const int n0 = blockIdx.z * NI * 32;
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, d.cout_pad * d.K * WB, 0x00020000);
    int b_off[B_SLOTS];
#pragma unroll
    for (int it = 0; it < B_SLOTS; ++it) b_off[it] = (n0 + tid / KQ + it * PSTEP) * d.K * WB + kqa * 16;
    const int b_lds0 = (tid / KQ) * LD + kqa * 4;
    floatx16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int try0 = PAIR ? TR * (wave & 1) : TR * MI * wave;
    const int wsp = PAIR ? (wave >> 1) : 0;
    int arow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) arow[i] = (wsp * SUBPIX + (try0 + TR * i + l31 / TC) * HW1 + (l31 % TC)) * LD + h * 4;
    const int brow = l31 * LD + h * 4;
    const float slope = d.src[0].slope;
    const int scs4 = d.src[0].cstride * 4;
    float4 ra[A_SLOTS], rb[B_SLOTS];
    float2 rbl[S3 ? B_SLOTS : 1];                                     // (S3) the lo pieces of the weight rows
    const int nchunk = d.Cin / BK;
    const int nstep = nchunk * 16;                                    // (chunk, plane, tap) steps: s = (chunk * 4 + plane) * 4 + tap

    // plane pl = 2 p + q; step tap tt = 2 ty + tx -> kernel tap (ky, kx) = (p ? 2 ty : 1 + 2 ty, q ? 2 tx : 1 + 2 tx), K offset (ky * 4 + kx) * Cin
#define RP_S2_LOAD_A(CH, PL)                                                                                   \
    {                                                                                                         \
        const int p_ = (PL) >> 1, q_ = (PL) & 1;                                                              \
        const int bad_ = (p_ ? 1 : 2) | (q_ ? 4 : 8);                                                         \
        const int dlt_ = p_ * d.Win + q_;                                                                     \
        _Pragma("unroll") for (int it = 0; it < A_SLOTS; ++it) {                                              \
            const int e_ = ((it < 8 ? a_edge : a_edge2) >> (4 * (it & 7))) & bad_;                            \
            int px_ = a_in[it];                                                                               \
            asm volatile("" : "+v"(px_));                                                                     \
            const int voff_ = (e_ ? 0 : px_ - dlt_) * scs4 + kqa * 16;                                        \
            ra[it] = rp_bufld4(rs_a, voff_, (CH) * BK * 4);                                                   \
        }                                                                                                     \
    }
#define RP_S2_STORE_A(CH, PL)                                                                                  \
    {                                                                                                         \
        const int p_ = (PL) >> 1, q_ = (PL) & 1;                                                              \
        const int bad_ = (p_ ? 1 : 2) | (q_ ? 4 : 8);                                                         \
        const float4* qq = reinterpret_cast<const float4*>(&sstab[2 * ((CH) * BK + kqa * 4)]);                \
        const float4 s0_ = qq[0], s1v_ = qq[1];                                                               \
        const rp_v2f sl2_ = {slope, slope};                                                                   \
        _Pragma("unroll") for (int it = 0; it < A_SLOTS; ++it) {                                              \
            rp_v2f v01 = {ra[it].x, ra[it].y}, v23 = {ra[it].z, ra[it].w};                                    \
            v01 = v01 * (rp_v2f){s0_.x, s0_.y} + (rp_v2f){s1v_.x, s1v_.y};                                    \
            v23 = v23 * (rp_v2f){s0_.z, s0_.w} + (rp_v2f){s1v_.z, s1v_.w};                                    \
            const rp_v2f t01 = v01 * sl2_, t23 = v23 * sl2_;                                                  \
            const float okf_ = ((((it < 8 ? a_edge : a_edge2) >> (4 * (it & 7))) & bad_) == 0) ? 1.f : 0.f;   \
            const rp_v2f mk_ = {okf_, okf_};                                                                  \
            v01 = (rp_v2f){fmaxf(v01.x, t01.x), fmaxf(v01.y, t01.y)} * mk_;                                   \
            v23 = (rp_v2f){fmaxf(v23.x, t23.x), fmaxf(v23.y, t23.y)} * mk_;                                   \
            if ((it + 1) * PSTEP <= NPIX || tid / KQ + it * PSTEP < NPIX)                                     \
                rp_tile_store_a<SPLIT>(&At[a_lds0 - kqa * 4 + it * PSTEP * LD], kqa, v01, v23);              \
        }                                                                                                     \
    }
#define RP_S2_LOAD_B(S)                                                                                        \
    {                                                                                                         \
        const int ch_ = (S) >> 4, pl_ = ((S) >> 2) & 3, tt_ = (S) & 3;                                        \
        const int ky_ = (pl_ >> 1) ? 2 * (tt_ >> 1) : 1 + 2 * (tt_ >> 1), kx_ = (pl_ & 1) ? 2 * (tt_ & 1) : 1 + 2 * (tt_ & 1); \
        const int so_ = ((ky_ * 4 + kx_) * d.Cin + ch_ * BK) * WB;                                            \
        _Pragma("unroll") for (int it = 0; it < B_SLOTS; ++it) {                                              \
            rb[it] = rp_bufld4(rs_b, b_off[it], so_);                                                         \
            if (S3) rbl[S3 ? it : 0] = rp_bufld2(rs_b, b_off[it] + 128 - kqa * 8, so_);                        \
        }                                                                                                     \
    }
#define RP_S2_STORE_B()                                                                                        \
    {                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < B_SLOTS; ++it) {                                              \
            *reinterpret_cast<float4*>(&Bt[b_lds0 + it * PSTEP * LD]) = rb[it];                               \
            if (S3) *reinterpret_cast<float2*>(&Bt[b_lds0 + 32 - kqa * 2 + it * PSTEP * LD]) = rbl[S3 ? it : 0]; \
        }                                                                                                     \
    }

    RP_S2_LOAD_A(0, 0)
    RP_S2_LOAD_B(0)
    __syncthreads();                                                  // sstab
    RP_S2_STORE_A(0, 0)
    RP_S2_STORE_B()
    __syncthreads();
    for (int cp = 0; cp < nchunk * 4; ++cp) {                         // (chunk, plane)
        const int ch = cp >> 2, pl = cp & 3;
        const bool lastp = (cp + 1 == nchunk * 4);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int sidx = cp * 4 + tt;
            // (RP_TILE_ABLATE: timing decomposition of this loop, experiments only -- 1 = no MFMAs / fragment reads, 2 = no staging (global loads,
            // transforms, LDS stores), 3 = no barriers, 4 = weights staged but the A tile never re-staged; results are wrong in every one of them)
            if (RP_TILE_ABLATE != 2) { if (sidx + 1 < nstep) RP_S2_LOAD_B(sidx + 1) }
            if (RP_TILE_ABLATE != 2 && RP_TILE_ABLATE != 4) { if (tt == 0 && !lastp) RP_S2_LOAD_A((cp + 1) >> 2, (cp + 1) & 3) }
            const int aoff = ((tt >> 1) * HW1 + (tt & 1)) * LD;
            if (RP_TILE_ABLATE != 1) {
                int ar_[MI], br_[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) ar_[i] = arow[i] + aoff;
#pragma unroll
                for (int j = 0; j < NI; ++j) br_[j] = brow + j * 32 * LD;
                rp_tile_mma<SPLIT, MI, NI>(acc, At, ar_, Bt, br_);
            }
            if (RP_TILE_ABLATE != 3) __syncthreads();
            if (RP_TILE_ABLATE != 2) { if (sidx + 1 < nstep) RP_S2_STORE_B() }
            if (RP_TILE_ABLATE != 2 && RP_TILE_ABLATE != 4) { if (tt == 3 && !lastp) RP_S2_STORE_A((cp + 1) >> 2, (cp + 1) & 3) }
            if (RP_TILE_ABLATE != 3) __syncthreads();
        }
        (void)ch; (void)pl;
    }
#undef RP_S2_LOAD_A
#undef RP_S2_STORE_A
#undef RP_S2_LOAD_B
#undef RP_S2_STORE_B

    if constexpr (SPLIT == 2 || SPLIT == 3) {          // undo the power-of-two weight pre-scale (exact)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= d.wscale;
    }
    // epilogue: one BatchNorm record per workgroup (stat_bm = rows per workgroup, slot 0), then the NHWC stores
    if (d.stat_part) {
        double* red = reinterpret_cast<double*>(&At[0]);              // [NW][NI * 32][2]
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            double sm = 0.0, sq = 0.0;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const double v = (double)acc[i][j][r]; sm += v; sq += v * v; }
            sm += rp_shfl_xor_d(sm, 32); sq += rp_shfl_xor_d(sq, 32);
            if (h == 0) { red[((wave * NI + j) * 32 + l31) * 2] = sm; red[((wave * NI + j) * 32 + l31) * 2 + 1] = sq; }
        }
        __syncthreads();
        if (tid < NI * 32) {
            double a = 0, b = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { a += red[((w * NI) * 32 + tid) * 2]; b += red[((w * NI) * 32 + tid) * 2 + 1]; }
            double* o = d.stat_part + (((size_t)blockIdx.x * 2) * d.cout_pad + n0 + tid) * 2;
            rp_stg(o, a); rp_stg(o + 1, b);
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int ry = try0 + TR * i + rl / TC, cx = rl % TC;
            const size_t pix = ((size_t)(wsp ? imgs[1] : imgs[0]) * d.Hout + (wsp ? y0s[1] : y0s[0]) + ry) * d.Wout + (wsp ? x0s[1] : x0s[0]) + cx;
            float* yo = d.y + pix * d.ycstride + d.ychoff + n0 + l31;
#pragma unroll
            for (int j = 0; j < NI; ++j)
                if (n0 + j * 32 + l31 < d.Cout) rp_stg(yo + j * 32, acc[i][j][r]);
        }
}

// ---- stride-2 4x4 convs on parity planes, LINEAR tiles (conv4, conv5: 28 x 28 / 14 x 14 grids that do not tile into patches) ----------
// Same decomposition as conv_s2_tile_kernel, but the M tile is 128 consecutive output pixels (like conv_igemm_kernel) and the plane
// pixels are staged as a STRIP: in the padded plane of (Hp + 1) x (Wp + 1) positions per image (local (r, c) = input pixel
// (2r - p, 2c - q); row Hp / column Wp and the p = 1 / q = 1 first row / column are zero padding) output (img, y, x) sits at position
// P = (img (Hp + 1) + y)(Wp + 1) + x and its four taps at P + {0, 1, Wp + 1, Wp + 2}.  A tile's taps therefore lie in ONE contiguous
// position range of <= BM + BM / Wp + 2 (Wp + 1) + 4 (+ Wp + 1 when the tile crosses an image) <= 224 positions, staged once per
// chunk and plane (conv_igemm_kernel: 4 x 128 rows).  Any Wp; a tile may span two images / two BatchNorm groups (scale / shift of both
// groups in registers, selected per staged position).  Used for split-K layers only: the epilogue writes partial sums
// [ks][M][cout_pad], reduced (and BatchNorm statistics taken) by the existing kernels.
template <int NI, int SPLIT = 0>
__global__ __launch_bounds__(256, SPLIT >= 4 ? 2 : 3) void conv_s2_strip_kernel(const ConvDesc* __restrict__ descs) {
    constexpr int LD = RowLd<SPLIT>::v;                               // LDS row stride (floats)
    constexpr bool S3 = SPLIT >= 4;                                   // three-piece bf16 operands: 6 bytes per weight, rows [hi | mid | lo]
    constexpr int WB = S3 ? 6 : 4;
    constexpr int BM = 128, SMAX = 224, PSTEP = 256 / KQ, A_SLOTS = SMAX / PSTEP, B_ROWS = NI * 32, B_SLOTS = B_ROWS / PSTEP;
    static_assert(BK == 32 && SMAX % PSTEP == 0 && B_ROWS % PSTEP == 0, "slot layout");
    __shared__ __attribute__((aligned(16))) float At[SMAX * LD];
    __shared__ __attribute__((aligned(16))) float Bt[B_ROWS * LD];
    const ConvDesc d = descs[0];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int ks = blockIdx.y / d.ntiles_n, n0 = (blockIdx.y - ks * d.ntiles_n) * NI * 32;
    const int m0 = blockIdx.x * BM, hw = d.Hp * d.Wp, W1 = d.Wp + 1, H1 = d.Hp + 1, img_pos = H1 * W1;
    if (((d.shared_slices >> ks) & 1) && m0 >= 2 * hw) return;              // a shared slice: only the first image pair's rows are ever read
    if ((d.skip_slices >> ks) & 1) return;                                  // a cached slice: its partial sums are already there
    auto pos_of = [&](int m) { const int img = m / hw, rem = m - img * hw; const int y = rem / d.Wp; return (img * H1 + y) * W1 + (rem - y * d.Wp); };
    const int pmin = pos_of(m0);
    const int nvalid = pos_of(min(m0 + BM, d.M) - 1) + W1 + 2 - pmin;          // staged positions (host guarantees <= SMAX)
    const int g0 = (m0 / hw) >> 1;                                             // first BatchNorm group of the tile; rsrc starts at image 2 g0
    const int kqa = tid % KQ;
    const int a_lds0 = (tid / KQ) * LD + kqa * 4;
    // slot it = position pmin + tid / 8 + PSTEP it: input pixel index of plane (0, 0) relative to image 2 g0, edge bits (as in
    // conv_s2_tile_kernel) and the BatchNorm group (0 / 1 relative to g0), packed: a_in[it] = index, bits in a_edge (4 per slot) / a_grp
    int a_in[A_SLOTS], a_edge = 0, a_grp = 0;
#pragma unroll
    for (int it = 0; it < A_SLOTS; ++it) {
        const int sl = tid / KQ + it * PSTEP;
        const int pos = pmin + sl;
        const int img = pos / img_pos, rem = pos - img * img_pos;
        const int r = rem / W1, c = rem - r * W1;
        const int iy = 2 * r, ix = 2 * c;
        a_in[it] = ((img - 2 * g0) * d.Hin + iy) * d.Win + ix;
        const int e = (sl >= nvalid || img >= d.Nimg) ? 15 : ((iy - 1 < 0 ? 1 : 0) | (iy >= d.Hin ? 2 : 0) | (ix - 1 < 0 ? 4 : 0) | (ix >= d.Win ? 8 : 0));
        a_edge |= e << (4 * it);
        a_grp |= (((img >> 1) - g0) & 1) << it;
    }
    static_assert(A_SLOTS <= 8, "edge bits in one register");
    const size_t base_px = (size_t)(2 * g0) * d.Hin * d.Win;
    const int nimg_left = min(4, d.Nimg - 2 * g0);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(d.src[0].x + base_px * d.src[0].cstride), 0,
                                                                         nimg_left * d.Hin * d.Win * 4 * d.src[0].cstride, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, d.cout_pad * d.K * WB, 0x00020000);
    // scale / shift of the two groups the tile can touch (clamped to the last group)
    const int ng = d.Nimg / 2;
    const float* ss0 = reinterpret_cast<const float*>(d.src[0].ss + (size_t)g0 * d.src[0].sstride) + kqa * 8;
    const float* ss1 = reinterpret_cast<const float*>(d.src[0].ss + (size_t)min(g0 + 1, ng - 1) * d.src[0].sstride) + kqa * 8;
    int b_off[B_SLOTS];
#pragma unroll
    for (int it = 0; it < B_SLOTS; ++it) b_off[it] = (n0 + tid / KQ + it * PSTEP) * d.K * WB + kqa * 16;
    const int b_lds0 = (tid / KQ) * LD + kqa * 4;
    floatx16 acc[1][NI];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    const int mrow = min(m0 + 32 * wave + l31, d.M - 1);                        // this lane's MFMA row (rows past M are never stored)
    const int arow = (pos_of(mrow) - pmin) * LD + h * 4;
    const int brow = l31 * LD + h * 4;
    const float slope = d.src[0].slope;
    const int scs4 = d.src[0].cstride * 4;
    const int nchunk = d.Cin / BK;
    const int cpk = (nchunk + d.ksplit - 1) / d.ksplit;
    const int ch_begin = ks * cpk, ch_end = min(nchunk, ch_begin + cpk);
    float4 ra[A_SLOTS], rb[B_SLOTS];
    float2 rbl[S3 ? B_SLOTS : 1];                                              // (S3) the lo pieces of the weight rows
    float4 sc0a, sc0b, sc1a, sc1b;                                             // {scale, shift} pairs of the thread's 4 channels, groups g0 / g0 + 1

#define RP_ST_LOAD_SS(CH)                                                                                       \
    { sc0a = rp_ldg4(ss0 + (CH) * BK * 2); sc0b = rp_ldg4(ss0 + (CH) * BK * 2 + 4);                               \
      sc1a = rp_ldg4(ss1 + (CH) * BK * 2); sc1b = rp_ldg4(ss1 + (CH) * BK * 2 + 4); }
#define RP_ST_LOAD_A(CH, PL)                                                                                   \
    {                                                                                                         \
        const int p_ = (PL) >> 1, q_ = (PL) & 1;                                                              \
        const int bad_ = (p_ ? 1 : 2) | (q_ ? 4 : 8);                                                         \
        const int dlt_ = p_ * d.Win + q_;                                                                     \
        _Pragma("unroll") for (int it = 0; it < A_SLOTS; ++it) {                                              \
            const int e_ = (a_edge >> (4 * it)) & bad_;                                                       \
            int px_ = a_in[it];                                                                               \
            asm volatile("" : "+v"(px_));                                                                     \
            const int voff_ = (e_ ? 0 : px_ - dlt_) * scs4 + kqa * 16;                                        \
            ra[it] = rp_bufld4(rs_a, voff_, (CH) * BK * 4);                                                   \
        }                                                                                                     \
    }
#define RP_ST_STORE_A(PL)                                                                                      \
    {                                                                                                         \
        const int p_ = (PL) >> 1, q_ = (PL) & 1;                                                              \
        const int bad_ = (p_ ? 1 : 2) | (q_ ? 4 : 8);                                                         \
        const rp_v2f sl2_ = {slope, slope};                                                                   \
        _Pragma("unroll") for (int it = 0; it < A_SLOTS; ++it) {                                              \
            const bool g1_ = (a_grp >> it) & 1;                                                               \
            const float4 qa_ = g1_ ? sc1a : sc0a, qb_ = g1_ ? sc1b : sc0b;     /* {s0, h0, s1, h1}, {s2, h2, s3, h3} */ \
            rp_v2f v01 = {ra[it].x, ra[it].y}, v23 = {ra[it].z, ra[it].w};                                    \
            v01 = v01 * (rp_v2f){qa_.x, qa_.z} + (rp_v2f){qa_.y, qa_.w};                                      \
            v23 = v23 * (rp_v2f){qb_.x, qb_.z} + (rp_v2f){qb_.y, qb_.w};                                      \
            const rp_v2f t01 = v01 * sl2_, t23 = v23 * sl2_;                                                  \
            const float okf_ = (((a_edge >> (4 * it)) & bad_) == 0) ? 1.f : 0.f;                              \
            const rp_v2f mk_ = {okf_, okf_};                                                                  \
            v01 = (rp_v2f){fmaxf(v01.x, t01.x), fmaxf(v01.y, t01.y)} * mk_;                                   \
            v23 = (rp_v2f){fmaxf(v23.x, t23.x), fmaxf(v23.y, t23.y)} * mk_;                                   \
            rp_tile_store_a<SPLIT>(&At[a_lds0 - kqa * 4 + it * PSTEP * LD], kqa, v01, v23);                  \
        }                                                                                                     \
    }
#define RP_ST_LOAD_B(S)                                                                                        \
    {                                                                                                         \
        const int ch_ = (S) >> 4, pl_ = ((S) >> 2) & 3, tt_ = (S) & 3;                                        \
        const int ky_ = (pl_ >> 1) ? 2 * (tt_ >> 1) : 1 + 2 * (tt_ >> 1), kx_ = (pl_ & 1) ? 2 * (tt_ & 1) : 1 + 2 * (tt_ & 1); \
        const int so_ = ((ky_ * 4 + kx_) * d.Cin + ch_ * BK) * WB;                                            \
        _Pragma("unroll") for (int it = 0; it < B_SLOTS; ++it) {                                              \
            rb[it] = rp_bufld4(rs_b, b_off[it], so_);                                                         \
            if (S3) rbl[S3 ? it : 0] = rp_bufld2(rs_b, b_off[it] + 128 - kqa * 8, so_);                        \
        }                                                                                                     \
    }
#define RP_ST_STORE_B()                                                                                        \
    {                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < B_SLOTS; ++it) {                                              \
            *reinterpret_cast<float4*>(&Bt[b_lds0 + it * PSTEP * LD]) = rb[it];                               \
            if (S3) *reinterpret_cast<float2*>(&Bt[b_lds0 + 32 - kqa * 2 + it * PSTEP * LD]) = rbl[S3 ? it : 0]; \
        }                                                                                                     \
    }

    if (ch_begin < ch_end) {
        const int s_begin = ch_begin * 16, s_end = ch_end * 16;                // (chunk, plane, tap) steps: s = (chunk * 4 + plane) * 4 + tap
        RP_ST_LOAD_SS(ch_begin)
        RP_ST_LOAD_A(ch_begin, 0)
        RP_ST_LOAD_B(s_begin)
        RP_ST_STORE_A(0)
        RP_ST_STORE_B()
        __syncthreads();
        for (int cp = ch_begin * 4; cp < ch_end * 4; ++cp) {                   // (chunk, plane)
            const bool lastp = (cp + 1 == ch_end * 4);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int sidx = cp * 4 + tt;
                if (sidx + 1 < s_end) RP_ST_LOAD_B(sidx + 1)
                if (tt == 0 && !lastp) {
                    RP_ST_LOAD_A((cp + 1) >> 2, (cp + 1) & 3)
                    // last plane of a chunk: all four planes of this chunk are stored already, so the next chunk's scale / shift
                    // can be fetched straight into the registers (consumed when its plane 0 is stored, 4 taps from now)
                    if (((cp + 1) & 3) == 0) RP_ST_LOAD_SS((cp + 1) >> 2)
                }
                const int aoff = ((tt >> 1) * W1 + (tt & 1)) * LD;
                {
                    const int ar_[1] = {arow + aoff};
                    int br_[NI];
#pragma unroll
                    for (int j = 0; j < NI; ++j) br_[j] = brow + j * 32 * LD;
                    rp_tile_mma<SPLIT, 1, NI>(acc, At, ar_, Bt, br_);
                }
                __syncthreads();
                if (sidx + 1 < s_end) RP_ST_STORE_B()
                if (tt == 3 && !lastp) {
                    RP_ST_STORE_A((cp + 1) & 3)
                }
                __syncthreads();
            }
        }
    }
#undef RP_ST_LOAD_SS
#undef RP_ST_LOAD_A
#undef RP_ST_STORE_A
#undef RP_ST_LOAD_B
#undef RP_ST_STORE_B
    // partial sums of this K slice (an empty slice writes zeros): reduced in fixed order by splitk_reduce_kernel
    const float wsc = (SPLIT == 2 || SPLIT == 3) ? d.wscale : 1.f;             // (16-bit modes: the power-of-two weight pre-scale, exact)
    float* po = d.partial + (size_t)ks * d.M * d.cout_pad;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= d.M) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) rp_stg(po + (size_t)m * d.cout_pad + n0 + j * 32 + l31, acc[0][j][r] * wsc);
    }
}

// y[pix(m)][col] = sum over K slices (fixed order) of the partial tiles
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvDesc* __restrict__ descs) {
    const ConvDesc d = descs[blockIdx.z];
    const int q4 = d.cout_pad >> 2;
    const size_t total = (size_t)d.M * q4;
    const int hw = d.Hp * d.Wp;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / q4), c4 = (int)(idx - (size_t)m * q4) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const int img = m / hw, rem = m - img * hw;
        const int msh = (img & 1) * hw + rem;                     // the row of a shared slice (ConvDesc::shared_slices)
        for (int ks = 0; ks < d.ksplit; ++ks) {
            const float4 v = rp_ldg4(d.partial + ((size_t)ks * d.M + (((d.shared_slices >> ks) & 1) ? msh : m)) * d.cout_pad + c4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        const int yp = rem / d.Wp, xp = rem - yp * d.Wp;
        const size_t pix = ((size_t)img * d.Hout + yp * d.osy + d.py) * d.Wout + xp * d.osx + d.px;
        float* yo = d.y + pix * d.ycstride + d.ychoff + c4;
        if (c4 + 3 < d.Cout) rp_stg4(yo, a);
        else { if (c4 < d.Cout) rp_stg(yo, a.x); if (c4 + 1 < d.Cout) rp_stg(yo + 1, a.y); if (c4 + 2 < d.Cout) rp_stg(yo + 2, a.z); }
    }
}

// ---- conv1{rgb,n,d} x {self, warped}: direct 3x3 convolution (mymodel.py:151,155,159 / :266-286) ------------
// Six tiny convs (Cin 4/4/2 -> 32) of the resized 16-channel input.  K is 36 or 18, far too small for an
// implicit GEMM; one wave handles 64 pixels of one (modality, stream) block so the weights are wave-uniform
// (scalar loads, SGPR operands) and every lane accumulates its 32 outputs in registers.
// w1: [6][9 taps][4 ch][32 out] (zero rows for the 2-channel depth block).
// BatchNorm statistics of A1 are fused: each pass (256 pixels of ONE image) leaves a float64 {sum, sum of squares}
// record per output channel in stat[group][pass in group][192][2], consumed by bn_finalize_kernel in pass order.
constexpr int C1_PASSES_PER_GROUP = 2 * RS * RS / 256;      // 392
__global__ __launch_bounds__(256, 2) void conv1_direct_kernel(const float* __restrict__ x0, const float* __restrict__ w1,
                                                            float* __restrict__ a1, double* __restrict__ stat, int n) {
    const size_t total = (size_t)n * RS * RS;      // multiple of 256 (224*224 = 196*256): a wave's pass never straddles the end
    __shared__ __attribute__((aligned(16))) float wl[6 * 9 * 4 * 32];  // all six blocks' weights, read as LDS broadcasts
    __shared__ __attribute__((aligned(16))) float tile[4 * 64 * 36];
    for (int i = threadIdx.x; i < 6 * 9 * 4 * 32; i += 256) wl[i] = w1[i];
    __syncthreads();
    // The six (modality, stream) blocks q = 2*m + s of a pixel run are computed back to back by the SAME wave: the
    // 16-channel input lines are fetched from HBM once (6 separate sweeps re-read X0 ~18x: 4.2 GB vs 0.2 GB, PMC)
    // and a pixel's 768-byte output row is completed within one pass.
    // A wave takes 256 consecutive pixels per pass, lane l the 4 horizontal neighbours base + 4l .. 4l+3 (224 is a
    // multiple of 4: they share a row).  Per kernel row the lane fetches its 6 input columns ONCE (the 3x3 windows
    // of the 4 pixels overlap) and only the channel quads this block needs; every weight quad read from LDS (a
    // broadcast, but still 1 KB of register writes) feeds 4 pixels; the 32 outputs of a pixel are 16 pairs
    // accumulated with packed FMAs (v_pk_fma_f32: two fp32 FMAs per lane per issue).
    constexpr int PX = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tw = tile + wave * 64 * 36;
    for (size_t base = ((size_t)blockIdx.x * 4 + wave) * (64 * PX); base < total; base += (size_t)gridDim.x * 4 * (64 * PX)) {
        const size_t pix = base + lane * PX;             // first of the lane's 4 pixels
        const int x = (int)(pix % RS), y = (int)((pix / RS) % RS);
        rp_v2f acc[PX][16];
        float nx[PX + 2][4];                             // the 4 input channels of the 6 columns of the NEXT (q, ty) step
        // Software pipeline over the 18 (block q, kernel row ty) steps: the loads of step it+1 are in flight while
        // step it runs its 768 packed FMAs (two waves per SIMD cannot hide an HBM round trip per step otherwise).
        // Only the channels the block needs are loaded: {rgb|n}: 3 + mask, depth: 1 + mask (mymodel.py:264-286).
#define RP_C1_LOAD(IT)                                                                                           \
        {                                                                                                        \
            const int q_ = (IT) / 3, ty_ = (IT) - q_ * 3, m_ = q_ >> 1, sft_ = (q_ & 1) * 8;                     \
            const int iy_ = y + ty_ - 1;                                                                         \
            const bool oky_ = (iy_ >= 0) & (iy_ < RS);                                                           \
            _Pragma("unroll") for (int cx = 0; cx < PX + 2; ++cx) {                                              \
                const int ix_ = x + cx - 1;                                                                      \
                const bool ok_ = oky_ & (ix_ >= 0) & (ix_ < RS);                                                 \
                const float* p_ = x0 + (pix + (ok_ ? (ptrdiff_t)(ty_ - 1) * RS + (cx - 1) : 0)) * 16 + sft_;     \
                if (m_ == 0) { nx[cx][0] = p_[0]; nx[cx][1] = p_[1]; nx[cx][2] = p_[2]; nx[cx][3] = p_[7]; }     \
                else if (m_ == 1) { nx[cx][0] = p_[3]; nx[cx][1] = p_[4]; nx[cx][2] = p_[5]; nx[cx][3] = p_[7]; } \
                else { nx[cx][0] = p_[6]; nx[cx][1] = p_[7]; nx[cx][2] = 0.f; nx[cx][3] = 0.f; }                 \
            }                                                                                                    \
        }
        RP_C1_LOAD(0)
#pragma unroll 1
        for (int it = 0; it < 18; ++it) {
            const int q = it / 3, ty = it - q * 3;           // block q = 2*modality + stream
            const int iy = y + ty - 1;
            const bool oky = (iy >= 0) & (iy < RS);
            float in[PX + 2][4];
#pragma unroll
            for (int cx = 0; cx < PX + 2; ++cx) {
                const int ix = x + cx - 1;
                const bool ok = oky & (ix >= 0) & (ix < RS);
#pragma unroll
                for (int c = 0; c < 4; ++c) in[cx][c] = ok ? nx[cx][c] : 0.f;       // zero padding
            }
            if (it + 1 < 18) RP_C1_LOAD(it + 1)
            if (ty == 0) {
#pragma unroll
                for (int k = 0; k < PX; ++k)
#pragma unroll
                    for (int o = 0; o < 16; ++o) acc[k][o] = (rp_v2f){0.f, 0.f};
            }
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    rp_v2f vv[PX];
#pragma unroll
                    for (int k = 0; k < PX; ++k) vv[k] = (rp_v2f){in[k + tx][c], in[k + tx][c]};
                    const float* wrow = &wl[((it * 3 + tx) * 4 + c) * 32];      // [q][ty][tx][c][32 out]
#pragma unroll
                    for (int o4 = 0; o4 < 8; ++o4) {
                        const float4 wv = *reinterpret_cast<const float4*>(wrow + o4 * 4);   // wave-uniform address
                        const rp_v2f w01 = {wv.x, wv.y}, w23 = {wv.z, wv.w};
#pragma unroll
                        for (int k = 0; k < PX; ++k) {
                            acc[k][o4 * 2 + 0] = vv[k] * w01 + acc[k][o4 * 2 + 0];
                            acc[k][o4 * 2 + 1] = vv[k] * w23 + acc[k][o4 * 2 + 1];
                        }
                    }
                }
            if (ty != 2) continue;
            // block q done.  Transpose through LDS: lane l first holds pixel 4l+k (32 channels); it then writes chunk
            // (l&7) of the pixels 4*((l>>3)+8j)+k, so every store instruction covers 8 full 128-B lines
            double ssum = 0.0, ssq = 0.0;                   // channel lane&31, pixel half lane>>5 of every 64-pixel run
#pragma unroll
            for (int k = 0; k < PX; ++k) {
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    *reinterpret_cast<float4*>(tw + lane * 36 + o * 4) = make_float4(acc[k][2 * o].x, acc[k][2 * o].y, acc[k][2 * o + 1].x, acc[k][2 * o + 1].y);
#pragma unroll 8
                for (int i = 0; i < 32; ++i) {
                    const double v = (double)tw[((lane >> 5) * 32 + i) * 36 + (lane & 31)];
                    ssum += v; ssq += v * v;
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int pl = (lane >> 3) + 8 * jj;
                    const float4 v = *reinterpret_cast<const float4*>(tw + pl * 36 + (lane & 7) * 4);
                    *reinterpret_cast<float4*>(a1 + (base + (size_t)pl * PX + k) * 192 + q * 32 + (lane & 7) * 4) = v;
                }
            }
            ssum += rp_shfl_xor_d(ssum, 32); ssq += rp_shfl_xor_d(ssq, 32);
            if (lane < 32) {
                const size_t pass = base / (64 * PX);        // = group * C1_PASSES_PER_GROUP + pass in group
                double* o = stat + (pass * 192 + q * 32 + lane) * 2;
                o[0] = ssum; o[1] = ssq;
            }
        }
#undef RP_C1_LOAD
    }
}

// ---- conv1 on the matrix pipe ---------------------------------------------------------------------------------
// Same six 3x3 convs as conv1_direct_kernel, as per-block GEMMs M = pixels, N = 32, K = 9 taps x 4 (2 for depth) channels
// on v_mfma_f32_32x32x2_f32.  One workgroup = an 8 x 32 pixel tile of one image: the 10 x 34 halo tile of the 16-channel
// input is staged in LDS once (pixel stride 17 floats: the A fragments of 32 neighbouring pixels hit 32 different banks), the
// weights of all six blocks as well; wave w owns tile rows 2w, 2w+1 (two 32-pixel M tiles sharing every B fragment).
// k = tap * 4 + c4 (depth: tap * 2 + c2): at MFMA step kk lanes 0-31 supply k = 2kk, lanes 32-63 k = 2kk + 1.
// C layout: lane holds output channel lane & 31 of 16 pixels -> every store instruction writes two full 128-byte lines of
// A1 and the BatchNorm sums need no transposition (the direct kernel went through LDS for both).  The VALU kernel ran
// 34 TFLOP/s on 44 GFLOP (1.08 ms); here the MFMA time is ~0.25 ms and the 2.47 GB of A1 stores are the bound.
// One {sum, sum of squares} record per tile and channel: stat[group][(img & 1) * 196 + tile][192][2] (C1_PASSES_PER_GROUP).
constexpr int C1T_PS = 17;                          // LDS pixel stride (floats)
constexpr int C1T_XIN = 10 * 34 * C1T_PS;
__global__ __launch_bounds__(256, 3) void conv1_mfma_kernel(const float* __restrict__ x0, const float* __restrict__ w1,
                                                          float* __restrict__ a1, double* __restrict__ stat, int n, int mode) {
    __shared__ __attribute__((aligned(16))) float wl[6 * 9 * 4 * 32];
    __shared__ __attribute__((aligned(16))) float xin[C1T_XIN];      // later reused for the per-wave statistics [4][192][2] f64
    static_assert(C1T_XIN * 4 >= 4 * 192 * 2 * 8, "statistics scratch aliases the input tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.x % 196, img = blockIdx.x / 196;
    const int ty0 = (tile / 7) * 8, tx0 = (tile % 7) * 32;
    for (int i = tid; i < 6 * 9 * 4 * 32 / 4; i += 256) reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(w1)[i];
    for (int i = tid; i < 340 * 4; i += 256) {
        const int p = i >> 2, q4 = i & 3;
        const int py = p / 34, px = p - py * 34;
        const int iy = ty0 + py - 1, ix = tx0 + px - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                  // zero padding
        if (iy >= 0 && iy < RS && ix >= 0 && ix < RS) v = *reinterpret_cast<const float4*>(x0 + (((size_t)img * RS + iy) * RS + ix) * 16 + q4 * 4);
        float* d = &xin[p * C1T_PS + q4 * 4];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int pb = ((2 * wave) * 34 + l31) * C1T_PS;                 // tile row 2w, column l31, tap (0, 0), channel 0
    double st_s[6], st_q[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int m = q >> 1, sft = (q & 1) * 8;
        // RELPOSE_FWD_ZERO_WARP: the warped-view blocks (odd q) are exact zeros in every image and only the first image pair's are read
        // (conv2 of those streams runs for one BatchNorm group): no MFMAs, no stores, zero statistics records for the other images
        // mode bit 1 (self-stream cache): the self-view blocks (even q) of A1 are still there from the forward that filled the cache
        if (((mode & 1) && img >= 2 && (q & 1)) || ((mode & 2) && !(q & 1))) { st_s[q] = 0.0; st_q[q] = 0.0; continue; }
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        const int nk = (m == 2) ? 9 : 18;
#pragma unroll
        for (int kk = 0; kk < nk; ++kk) {
            // (tap, in-block channel) of k = 2kk (lanes 0-31) and k = 2kk + 1 (lanes 32-63); input channel: rgb {0,1,2,7}, n {3,4,5,7}, d {6,7}
            const int tap = (m == 2) ? kk : (kk >> 1);
            const int c40 = (m == 2) ? 0 : 2 * (kk & 1), c41 = c40 + 1;
            const int ch0 = (m == 2) ? 6 : (c40 == 3 ? 7 : m * 3 + c40);
            const int ch1 = (m == 2) ? 7 : (c41 == 3 ? 7 : m * 3 + c41);
            const int tyy = tap / 3, txx = tap - tyy * 3;
            const int off = (tyy * 34 + txx) * C1T_PS + sft;
            const int idx = pb + off + (h ? ch1 : ch0);
            const float a0 = xin[idx];
            const float a1v = xin[idx + 34 * C1T_PS];
            const float bv = wl[((q * 9 + tap) * 4 + (h ? c41 : c40)) * 32 + l31];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, bv, acc1, 0, 0, 0);
        }
        // store: register r = pixel column (r & 3) + 8 (r >> 2) + 4 h of the M tile, output channel l31
        double ssum = 0.0, ssq = 0.0;
        float* o0 = a1 + ((((size_t)img * RS + ty0 + 2 * wave) * RS + tx0 + 4 * h) * 192 + q * 32 + l31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = (r & 3) + 8 * (r >> 2);
            o0[(size_t)px * 192] = acc0[r];
            o0[((size_t)RS + px) * 192] = acc1[r];
            const double v0 = (double)acc0[r], v1 = (double)acc1[r];
            ssum += v0; ssq += v0 * v0;
            ssum += v1; ssq += v1 * v1;
        }
        st_s[q] = ssum + rp_shfl_xor_d(ssum, 32);
        st_q[q] = ssq + rp_shfl_xor_d(ssq, 32);
    }
    __syncthreads();                                                   // every wave is done with the input tile
    double* wst = reinterpret_cast<double*>(xin);
    if (h == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { wst[(wave * 192 + q * 32 + l31) * 2] = st_s[q]; wst[(wave * 192 + q * 32 + l31) * 2 + 1] = st_q[q]; }
    }
    __syncthreads();
    if (tid < 192) {
        const double ssum = ((wst[tid * 2] + wst[(192 + tid) * 2]) + wst[(384 + tid) * 2]) + wst[(576 + tid) * 2];
        const double ssq = ((wst[tid * 2 + 1] + wst[(192 + tid) * 2 + 1]) + wst[(384 + tid) * 2 + 1]) + wst[(576 + tid) * 2 + 1];
        const size_t pass = (size_t)(img >> 1) * C1_PASSES_PER_GROUP + (size_t)(img & 1) * 196 + tile;
        double* o = stat + (pass * 192 + tid) * 2;
        o[0] = ssum; o[1] = ssq;
    }
}

// ---- the five 1x1 heads deconv1{rgb,n,d,s,f} (mymodel.py:188,196,204,220,228 / :312-376) in one pass ---------
// HBM-bound: every pixel reads its 224 D2 channels (+ the 3x32 skip channels of A1) exactly once, applies
// BatchNorm + LeakyReLU, and feeds the five small matrix-vector products (3+3+1+S+32 outputs, 64 inputs each).
// One lane per pixel; weights are wave-uniform LDS broadcasts (ds_read_b128 of 4 output weights).
// Weight image (pack_heads): per input channel a row of padded output weights --
//   [0,768)      rgb/n/d rows (D2[0:96] then the A1 skip blocks), 4 floats per row
//   [768,2304)   s rows  (D2[96:160]),  24 floats per row
//   [2304,4352)  f rows  (D2[160:224]), 32 floats per row
struct HeadsDesc {
    const float* d2; const float* a1; const float2* ss_d2; const float2* ss_a1;
    const float* w; const float* bias; float* out;
    int n, S, cf, use_tanh;
    // self-stream cache: the skip source of the rgb / n / d heads (the self-view blocks of A1) is level-invariant and is accumulated FIRST
    // in every plan; snap_mode 1 stores the three heads' accumulators after it (12 floats per pixel), snap_mode 2 starts from them and
    // does not read A1 at all (384 of the 1280 bytes a pixel reads)
    int snap_mode; float* snap;
};
constexpr int HEADS_W = 4352;

// Round-5 experiment, NOT kept: the three accumulator sets one after the other + 3 waves per SIMD (238 -> 132 VGPRs, no spills, bitwise the same
// output).  Alone the kernel gets a third wave per SIMD; in the pipeline the headline DROPPED 668 -> 655 pairs/s: at 238 VGPRs a wave of this kernel
// does not fit beside two waves of the fp32 tile kernels (168 VGPRs each), so a workgroup only enters a CU where TWO conv workgroups have left -- it
// fills the drain of the conv launches and otherwise stays out; at 132 VGPRs it takes every single slot a conv workgroup frees and keeps it (the next
// heads workgroup fits where a conv workgroup does not) -- the working hypothesis; the converse experiment, head / tail launches padded with unused
// LDS so that they need more than one freed conv slot (conv1 +6 / +12 KB, heads and resize_out +8 KB), changed nothing at configs[1] (675.0 / 674.2
// base, 673.6, 676.2, 676.3, 673.1) and cost 2.7 % at configs[2]: not kept either (tools/gpu_r5_pads.sh).
template <int S, bool POSE = false>      // POSE (RELPOSE_FWD_POSE_OUTPUTS): the n, d and f heads only; rgb and semantic channels are written as zeros
__global__ __launch_bounds__(256) void heads_kernel(const HeadsDesc hd) {
    __shared__ __attribute__((aligned(16))) float wl[HEADS_W];
    __shared__ float2 ssl[320];                        // scale/shift of this block's BatchNorm group
    constexpr int cf = 7 + S + 32;
    const size_t pix0 = (size_t)blockIdx.x * 256;      // 256 consecutive pixels: one image, one BatchNorm group
    const int g = (int)(pix0 / ((size_t)RS * RS)) >> 1;
    for (int i = threadIdx.x; i < HEADS_W; i += 256) wl[i] = hd.w[i];
    for (int i = threadIdx.x; i < 320; i += 256) {
        // A1 skip blocks: channels [0:32] (rgb self), [64:96] (n self), [128:160] (d self) of the 192-channel buffer
        ssl[i] = i < 224 ? hd.ss_d2[(size_t)g * 224 + i] : hd.ss_a1[(size_t)g * 192 + ((i - 224) >> 5) * 64 + ((i - 224) & 31)];
    }
    __syncthreads();
    const size_t pix = pix0 + threadIdx.x;
    const float* pd = hd.d2 + pix * 224;
    const float* pa = hd.a1 + pix * 192;
    // accumulators as PAIRS: one v_pk_fma_f32 (two independent fused multiply-adds: the same roundings as two v_fma_f32) per weight pair --
    // half the vector instructions of the 4352 multiply-adds a pixel costs (round 6)
    rp_v2f a3[6], as_[12], af[16];                     // rgb 0:3 | n 4:7 | d 8:11 (padded quads), s, f
#pragma unroll
    for (int o = 0; o < 6; ++o) a3[o] = (rp_v2f){0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 12; ++o) as_[o] = (rp_v2f){0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 16; ++o) af[o] = (rp_v2f){0.f, 0.f};
    // One 128-byte line (32 channels) of a pixel at a time: all 8 loads are issued back to back so the line is
    // fetched once (a wave touches 64 lines per load instruction; interleaving compute between the loads of a
    // line let other waves evict it from the 32 KB L1 first).  The NEXT line's loads are issued before this line's arithmetic
    // (two register sets, alternating): at 234 VGPRs only 2 waves share a SIMD and nothing else hides the trip to HBM.
    // Then BN + LeakyReLU and acc[4j+t'] += v * w[row][4j+t'] over the row's NQ4 weight quads.
#define RP_HEAD_LOAD(X8, PTR)                                                                             \
    { _Pragma("unroll") for (int q = 0; q < 8; ++q) X8[q] = reinterpret_cast<const float4*>(PTR)[q]; }
#define RP_HEAD_LINE(X8, NEXT, SSROW, WOFF, WSTRIDE, NQ4, ACC, OBASE)                                     \
    {                                                                                                    \
        NEXT                                                                                             \
        _Pragma("unroll 1") for (int q = 0; q < 8; ++q) {                                                \
            const float4 x4 = X8[0];                                                                     \
            _Pragma("unroll") for (int z = 0; z < 7; ++z) X8[z] = X8[z + 1];                             \
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w};                                                \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                              \
                const float2 sc = ssl[(SSROW) + q * 4 + t];                                              \
                const float v = lrelu(xv[t] * sc.x + sc.y, LRELU);                                       \
                const rp_v2f vv = {v, v};                                                                \
                _Pragma("unroll") for (int j = 0; j < (NQ4); ++j) {                                      \
                    const float4 w4 = *reinterpret_cast<const float4*>(&wl[(WOFF) + (q * 4 + t) * (WSTRIDE) + j * 4]); \
                    ACC[(OBASE) / 2 + j * 2 + 0] = __builtin_elementwise_fma(vv, (rp_v2f){w4.x, w4.y}, ACC[(OBASE) / 2 + j * 2 + 0]); \
                    ACC[(OBASE) / 2 + j * 2 + 1] = __builtin_elementwise_fma(vv, (rp_v2f){w4.z, w4.w}, ACC[(OBASE) / 2 + j * 2 + 1]); \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
    }
    // line order: A1 skip lines of [rgb,] n, d (level-invariant: the snapshot point), then the D2 lines of [rgb,] n, d, [s: 2 lines,] f: 2 lines
    float4 xa[8], xb[8];
    constexpr int M0 = POSE ? 1 : 0;
    float4* snp = reinterpret_cast<float4*>(hd.snap) + pix * 3;
    // (two register sets, strictly alternating: a line's NEXT load goes to the set the line does not read)
#define RP_HEAD_SKIP(X8, NEXT, M_) RP_HEAD_LINE(X8, NEXT, 224 + (M_) * 32, (96 + (M_) * 32) * 4, 4, 1, a3, (M_) * 4)
#define RP_HEAD_D2(X8, NEXT, M_) RP_HEAD_LINE(X8, NEXT, (M_) * 32, ((M_) * 32) * 4, 4, 1, a3, (M_) * 4)
    if (hd.snap_mode == 2) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const float4 v = rp_ldg4(reinterpret_cast<const float*>(snp + m));
            a3[2 * m] = (rp_v2f){v.x, v.y}; a3[2 * m + 1] = (rp_v2f){v.z, v.w};
        }
        if constexpr (POSE) { RP_HEAD_LOAD(xa, pd + 32) } else { RP_HEAD_LOAD(xb, pd) }
    } else {
        if constexpr (POSE) {
            RP_HEAD_LOAD(xa, pa + 64)
            RP_HEAD_SKIP(xa, RP_HEAD_LOAD(xb, pa + 128), 1)
            RP_HEAD_SKIP(xb, RP_HEAD_LOAD(xa, pd + 32), 2)
        } else {
            RP_HEAD_LOAD(xa, pa)
            RP_HEAD_SKIP(xa, RP_HEAD_LOAD(xb, pa + 64), 0)
            RP_HEAD_SKIP(xb, RP_HEAD_LOAD(xa, pa + 128), 1)
            RP_HEAD_SKIP(xa, RP_HEAD_LOAD(xb, pd), 2)
        }
        if (hd.snap_mode == 1) {
#pragma unroll
            for (int m = 0; m < 3; ++m) rp_stg4(reinterpret_cast<float*>(snp + m), make_float4(a3[2 * m].x, a3[2 * m].y, a3[2 * m + 1].x, a3[2 * m + 1].y));
        }
    }
    if constexpr (POSE) {
        RP_HEAD_D2(xa, RP_HEAD_LOAD(xb, pd + 64), 1)
        RP_HEAD_D2(xb, RP_HEAD_LOAD(xa, pd + 160), 2)
    } else {
        RP_HEAD_D2(xb, RP_HEAD_LOAD(xa, pd + 32), 0)
        RP_HEAD_D2(xa, RP_HEAD_LOAD(xb, pd + 64), 1)
        RP_HEAD_D2(xb, RP_HEAD_LOAD(xa, pd + 96), 2)
    }
#undef RP_HEAD_SKIP
#undef RP_HEAD_D2
    if (!POSE) {
        RP_HEAD_LINE(xa, RP_HEAD_LOAD(xb, pd + 128), 96, 768, 24, 6, as_, 0)                                                // s
        RP_HEAD_LINE(xb, RP_HEAD_LOAD(xa, pd + 160), 128, 768 + 32 * 24, 24, 6, as_, 0)
    }
    RP_HEAD_LINE(xa, RP_HEAD_LOAD(xb, pd + 192), 160, 2304, 32, 8, af, 0)                                                   // f
    RP_HEAD_LINE(xb, , 192, 2304 + 32 * 32, 32, 8, af, 0)
#undef RP_HEAD_LINE
#undef RP_HEAD_LOAD
    // bias, tanh, store (cf is even: 8-byte stores; a lane's cf floats are contiguous in the NHWC output)
    float r[cf + 1];
#pragma unroll
    for (int o = 0; o < 3; ++o) { r[o] = POSE ? 0.f : a3[o >> 1][o & 1] + hd.bias[o]; r[3 + o] = a3[2 + (o >> 1)][o & 1] + hd.bias[3 + o]; }
    r[6] = a3[4].x + hd.bias[6];
#pragma unroll
    for (int o = 0; o < S; ++o) r[7 + o] = POSE ? 0.f : as_[o >> 1][o & 1] + hd.bias[7 + o];
#pragma unroll
    for (int k = 0; k < 32; ++k) { const float v = af[k >> 1][k & 1] + hd.bias[7 + S + k]; r[7 + S + k] = hd.use_tanh ? tanhf(v) : v; }
    float* o = hd.out + pix * cf;
#pragma unroll
    for (int k = 0; k < cf / 2; ++k) *reinterpret_cast<float2*>(o + 2 * k) = make_float2(r[2 * k], r[2 * k + 1]);
}

// Split-K reduce + BatchNorm partial sums in one pass (round 2): grid (chunk, group, member); a workgroup adds the K slices of a
// chunk of one group's rows of member z in fixed order, stores y, and leaves the float64 {sum, sum of squares} of those rows per
// channel as record [group][z * nch + chunk][C][2] for bn_finalize_kernel -- y is not read back by a separate statistics pass.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const ConvDesc* __restrict__ descs, int nch, int C, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) double sm_rs[];
    const ConvDesc d = descs[blockIdx.z];
    const int q4 = d.cout_pad >> 2, nrl = 256 / q4;
    const int cq = threadIdx.x % q4, rl = threadIdx.x / q4;
    const int g = blockIdx.y, chunk = blockIdx.x;
    const int hw = d.Hp * d.Wp, R = 2 * hw;
    const int chunk_rows = (R + nch - 1) / nch;
    const int r0 = g * R + chunk * chunk_rows, r1 = min(min(r0 + chunk_rows, (g + 1) * R), d.M);
    double sv[4] = {0, 0, 0, 0}, qv[4] = {0, 0, 0, 0};
    if (rl < nrl) {
        const int c4 = cq * 4;
        for (int m = r0 + rl; m < r1; m += nrl) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            const int img = m / hw, rem = m - img * hw;
            const int msh = (img & 1) * hw + rem;                 // the row of a shared slice (ConvDesc::shared_slices)
            for (int ks = 0; ks < d.ksplit; ++ks) {
                const float4 v = rp_ldg4(d.partial + ((size_t)ks * d.M + (((d.shared_slices >> ks) & 1) ? msh : m)) * d.cout_pad + c4);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            const int yp = rem / d.Wp, xp = rem - yp * d.Wp;
            const size_t pix = ((size_t)img * d.Hout + yp * d.osy + d.py) * d.Wout + xp * d.osx + d.px;
            float* yo = d.y + pix * d.ycstride + d.ychoff + c4;
            if (c4 + 3 < d.Cout) rp_stg4(yo, a);
            else { if (c4 < d.Cout) rp_stg(yo, a.x); if (c4 + 1 < d.Cout) rp_stg(yo + 1, a.y); if (c4 + 2 < d.Cout) rp_stg(yo + 2, a.z); }
            const double e0 = a.x, e1 = a.y, e2 = a.z, e3 = a.w;
            sv[0] += e0; sv[1] += e1; sv[2] += e2; sv[3] += e3;
            qv[0] += e0 * e0; qv[1] += e1 * e1; qv[2] += e2 * e2; qv[3] += e3 * e3;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { sm_rs[(rl * q4 + cq) * 8 + k] = sv[k]; sm_rs[(rl * q4 + cq) * 8 + 4 + k] = qv[k]; }
    }
    __syncthreads();
    if (rl == 0) {
        const size_t rec = (size_t)g * (nch * gridDim.z) + blockIdx.z * nch + chunk;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = cq * 4 + k;
            if (c >= d.Cout) continue;
            double a = 0, b = 0;
            for (int j = 0; j < nrl; ++j) { a += sm_rs[(j * q4 + cq) * 8 + k]; b += sm_rs[(j * q4 + cq) * 8 + 4 + k]; }
            double* o = partial + (rec * C + d.ychoff + c) * 2;
            o[0] = a; o[1] = b;
        }
    }
}

// ---- BatchNorm batch statistics (per group of 2 images, per channel), float64 ---------------------
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int rows_per_group, int C, int chunk_rows,
                                                          double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int q4 = C >> 2;
    const int nrl = 256 / q4;
    const int cq = threadIdx.x % q4, rl = threadIdx.x / q4;
    const int g = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (rl < nrl) {
        const int r1 = min((chunk + 1) * chunk_rows, rows_per_group);
        const float* base = x + (size_t)g * rows_per_group * C + cq * 4;
        for (int r = chunk * chunk_rows + rl; r < r1; r += nrl) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)r * C);
            const double a = v.x, b = v.y, c = v.z, d = v.w;
            s[0] += a; s[1] += b; s[2] += c; s[3] += d;
            ss[0] += a * a; ss[1] += b * b; ss[2] += c * c; ss[3] += d * d;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { sm[(rl * q4 + cq) * 8 + k] = s[k]; sm[(rl * q4 + cq) * 8 + 4 + k] = ss[k]; }
    }
    __syncthreads();
    if (rl == 0) {
        double* o = partial + (((size_t)g * nchunks + chunk) * C + cq * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double a = 0, b = 0;
            for (int j = 0; j < nrl; ++j) { a += sm[(j * q4 + cq) * 8 + k]; b += sm[(j * q4 + cq) * 8 + 4 + k]; }
            o[k * 2] = a; o[k * 2 + 1] = b;
        }
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ partial, int nchunks, int C, int rows_per_group,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float2* __restrict__ ss) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    if (c >= C) return;
    double s = 0, q = 0;
    for (int k = 0; k < nchunks; ++k) {
        const double* p = partial + (((size_t)g * nchunks + k) * C + c) * 2;
        s += p[0]; q += p[1];
    }
    const double mean = s / rows_per_group;
    double var = q / rows_per_group - mean * mean;       // biased variance (training-mode BN)
    if (var < 0) var = 0;
    const double sc = (double)gamma[c] / sqrt(var + BN_EPS);
    ss[(size_t)g * C + c] = make_float2((float)sc, (float)((double)beta[c] - mean * sc));
}

// batchnorm = 0 nets (mymodel.py:22-25): the buffer's {scale, shift} are constants, {1, conv bias} (`gamma` / `beta` hold them), for every group
__global__ void ss_fill_kernel(int C, int G, const float* __restrict__ gamma, const float* __restrict__ beta, float2* __restrict__ ss) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * G) return;
    const int c = i % C;
    ss[i] = make_float2(gamma[c], beta[c]);
}

// Same for record lists of up to a few hundred entries: a workgroup = 32 consecutive channels x 32 record parts of one group (lane =
// channel: coalesced record reads); part k adds records k, k + 32, ..., the parts are then added in order.
// skip_blk > 0 (self-stream cache): the even blocks of skip_blk channels (the self-view streams) keep the {scale, shift} they have
__global__ __launch_bounds__(1024) void bn_finalize_parts_kernel(const double* __restrict__ partial, int nchunks, int C, int rows_per_group,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float2* __restrict__ ss, int skip_blk) {
    __shared__ double red[32][32][2];
    const int cq = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cq, g = blockIdx.y;
    if (skip_blk > 0 && !(((blockIdx.x * 32) / skip_blk) & 1)) return;          // (block-uniform: skip_blk is a multiple of 32)
    double s = 0, q = 0;
    if (c < C)
        for (int k = part; k < nchunks; k += 32) {
            const double2 v = *reinterpret_cast<const double2*>(partial + (((size_t)g * nchunks + k) * C + c) * 2);
            s += v.x; q += v.y;
        }
    red[part][cq][0] = s; red[part][cq][1] = q;
    __syncthreads();
    if (part || c >= C) return;
    s = 0; q = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) { s += red[k][cq][0]; q += red[k][cq][1]; }
    const double mean = s / rows_per_group;
    double var = q / rows_per_group - mean * mean;
    if (var < 0) var = 0;
    const double sc = (double)gamma[c] / sqrt(var + BN_EPS);
    ss[(size_t)g * C + c] = make_float2((float)sc, (float)((double)beta[c] - mean * sc));
}

// Finalise BatchNorm from the per-tile records written by the conv epilogues.  A workgroup = 32 consecutive channels x 32 record
// parts of one group: lane = channel, so a wave's 16-byte record reads are 512 contiguous bytes (one wave per channel walking
// its records read 16 bytes per 1-4 KB line: 50 us per layer on D2 / D3).  Part k adds records k, k + 32, ... of every launch
// member that wrote the channel, in (member, tile) order; the 32 parts are then added in order: fixed order, deterministic.
__global__ __launch_bounds__(1024) void bn_finalize_fused_kernel(const ConvDesc* __restrict__ descs, int ndesc, int skip_blk, int C,
                                                                 int rows_per_group, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float2* __restrict__ ss) {
    __shared__ double red[32][32][2];
    const int cq = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cq, g = blockIdx.y;
    if (skip_blk > 0 && !(((blockIdx.x * 32) / skip_blk) & 1)) return;          // self-stream cache: see bn_finalize_parts_kernel
    double s = 0, q = 0;
    if (c < C) {
        for (int z = 0; z < ndesc; ++z) {
            const ConvDesc& d = descs[z];
            const int cl = c - d.ychoff;
            if (cl < 0 || cl >= d.Cout) continue;
            const int hw = d.Hp * d.Wp;
            const int BMt = d.stat_bm;
            const int t0 = (2 * g * hw) / BMt, t1 = min(((2 * g + 2) * hw - 1) / BMt, (d.M - 1) / BMt);
            for (int t = t0 + part; t <= t1; t += 32) {
                const int sl = g - (((t * BMt) / hw) >> 1);
                if (sl < 0 || sl > 1) continue;
                const double2 v = *reinterpret_cast<const double2*>(d.stat_part + (((size_t)t * 2 + sl) * d.cout_pad + cl) * 2);
                s += v.x; q += v.y;
            }
        }
    }
    red[part][cq][0] = s; red[part][cq][1] = q;
    __syncthreads();
    if (part || c >= C) return;
    s = 0; q = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) { s += red[k][cq][0]; q += red[k][cq][1]; }
    const double mean = s / rows_per_group;
    double var = q / rows_per_group - mean * mean;
    if (var < 0) var = 0;
    const double sc = (double)gamma[c] / sqrt(var + BN_EPS);
    ss[(size_t)g * C + c] = make_float2((float)sc, (float)((double)beta[c] - mean * sc));
}

// RELPOSE_FWD_ZERO_WARP: the warped-stream channel blocks (odd blocks of `blk` channels) of buffer a [n][hw][C] were computed for images
// 0 and 1 only; image i >= 2 gets a copy of image i & 1, BatchNorm group g >= 1 the {scale, shift} of group 0 for those channels.
__global__ __launch_bounds__(256) void bcast_warped_kernel(float* __restrict__ a, int n, int hw, int C, int blk, float2* __restrict__ ss, int G) {
    const int q4 = 3 * blk / 4;                                   // float4 per pixel to copy
    const size_t total = (size_t)(n - 2) * hw * q4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % q4);
        const size_t pi = i / q4;
        const int px = (int)(pi % hw), img = 2 + (int)(pi / hw);
        const int ch = (2 * (q / (blk / 4)) + 1) * blk + 4 * (q % (blk / 4));
        const float4 v = *reinterpret_cast<const float4*>(a + ((size_t)(img & 1) * hw + px) * C + ch);      // (a: a kernel argument, global loads)
        *reinterpret_cast<float4*>(a + ((size_t)img * hw + px) * C + ch) = v;
    }
    const int tsz = (G - 1) * 3 * blk;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tsz; i += gridDim.x * blockDim.x) {
        const int q = i % (3 * blk), g = 1 + i / (3 * blk);
        const int ch = (2 * (q / blk) + 1) * blk + q % blk;
        ss[(size_t)g * C + ch] = ss[ch];
    }
}

// ---- bilinear resize, align_corners=False (mymodel.py:261,379) --------------------------------------
__device__ __forceinline__ void lin_coef(int dst, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * (dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = src - i0;
    l0 = 1.f - l1;
}

// x [n,16,H,W] NCHW -> y [n,RS,RS,16] NHWC
__global__ void resize_in_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int H, int W, int c_begin) {
    const size_t total = (size_t)n * RS * RS;
    const float shh = (float)H / RS, sww = (float)W / RS;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % RS), oy = (int)((idx / RS) % RS), img = (int)(idx / ((size_t)RS * RS));
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        lin_coef(oy, shh, H, y0, y1, ly0, ly1);
        lin_coef(ox, sww, W, x0, x1, lx0, lx1);
        // c_begin = 8 (self-stream cache): channels 0:8 of X0 are already there
        float out[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < c_begin) continue;
            const float* p = x + ((size_t)img * 16 + c) * H * W;
            out[c] = ly0 * (lx0 * p[(size_t)y0 * W + x0] + lx1 * p[(size_t)y0 * W + x1]) +
                     ly1 * (lx0 * p[(size_t)y1 * W + x0] + lx1 * p[(size_t)y1 * W + x1]);
        }
        float4* o = reinterpret_cast<float4*>(y + idx * 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) if (4 * c >= c_begin) o[c] = make_float4(out[4 * c], out[4 * c + 1], out[4 * c + 2], out[4 * c + 3]);
    }
}

// x [n,RS,RS,C] NHWC -> y [n,C,H,W] NCHW.  One wave = 64 consecutive output pixels of one row.  The <=26
// source pixels x 2 source rows they interpolate from are first copied to LDS with contiguous 8-byte loads
// (C = 54/60 floats per pixel: 8-B aligned), then every lane blends its 4 taps from LDS and writes plane by
// plane (256 contiguous bytes per store instruction).
#define RO_MAXW 28
#define RO_MAXC 60                                   // channels of the LDS path (7 + S + 32 <= 60); more: the generic path
#define RO_PF ((RO_MAXW * RO_MAXC / 2 + 63) / 64)    // float2 per lane and source row
__global__ __launch_bounds__(256) void resize_out_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float smem_ro[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* srow = smem_ro + (size_t)wave * 2 * RO_MAXW * C;
    const int segs = (W + 63) / 64;
    const size_t nseg = (size_t)n * H * segs;
    const float shh = (float)RS / H, sww = (float)RS / W;
    // a wave's segment: source rows y0 / y1 and the source pixel range [sx, sx + width) of its 64 output pixels
    struct Seg { int img, oy, ox_first, y0, y1, sx, width; float ly0, ly1; bool lds; };
    auto seg_of = [&](size_t sidx) {
        Seg g;
        const int seg = (int)(sidx % segs);
        g.oy = (int)((sidx / segs) % H); g.img = (int)(sidx / ((size_t)segs * H));
        g.ox_first = seg * 64;
        const int ox_last = min(g.ox_first + 63, W - 1);
        int xa, xb, xc, xd; float t0, t1;
        lin_coef(g.oy, shh, RS, g.y0, g.y1, g.ly0, g.ly1);
        lin_coef(g.ox_first, sww, RS, xa, xb, t0, t1);
        lin_coef(ox_last, sww, RS, xc, xd, t0, t1);
        g.sx = xa; g.width = xd - xa + 1;           // <= RO_MAXW for scale <= 0.4 (224/640 = 0.35)
        g.lds = g.width <= RO_MAXW && !(C & 1) && C <= RO_MAXC;
        return g;
    };
    // The source pixels of the NEXT segment travel in registers while this segment is blended and stored (a wave did load -> blend ->
    // store strictly in turn before: 698 us per forward at 64 images).
    float2 pf[2][RO_PF];
    auto prefetch = [&](const Seg& g) {
        const int nf2 = g.width * C / 2;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float2* src = reinterpret_cast<const float2*>(x + (((size_t)g.img * RS + (r ? g.y1 : g.y0)) * RS + g.sx) * C);
#pragma unroll
            for (int k = 0; k < RO_PF; ++k) { const int i = lane + 64 * k; if (i < nf2) pf[r][k] = src[i]; }
        }
    };
    const size_t step = (size_t)gridDim.x * 4;
    size_t sidx = (size_t)blockIdx.x * 4 + wave;
    Seg cur;
    if (sidx < nseg) { cur = seg_of(sidx); if (cur.lds) prefetch(cur); }
    for (; sidx < nseg; sidx += step) {
        const Seg g = cur;
        if (!g.lds) {
            // generic shapes (W < ~560: the 64-pixel segment spans more source pixels than the LDS region holds; odd C:
            // pixel rows are not 8-byte aligned): every lane gathers its 4 taps straight from global memory
            const int ox = g.ox_first + lane;
            if (ox < W) {
                int x0, x1; float lx0, lx1;
                lin_coef(ox, sww, RS, x0, x1, lx0, lx1);
                const float* r0 = x + ((size_t)g.img * RS + g.y0) * RS * C;
                const float* r1 = x + ((size_t)g.img * RS + g.y1) * RS * C;
                const float* a = r0 + (size_t)x0 * C; const float* b = r0 + (size_t)x1 * C;
                const float* cc = r1 + (size_t)x0 * C; const float* dd = r1 + (size_t)x1 * C;
                float* o = y + (size_t)g.img * C * H * W + (size_t)g.oy * W + ox;
                for (int c = 0; c < C; ++c)
                    o[(size_t)c * H * W] = g.ly0 * (lx0 * a[c] + lx1 * b[c]) + g.ly1 * (lx0 * cc[c] + lx1 * dd[c]);
            }
            if (sidx + step < nseg) { cur = seg_of(sidx + step); if (cur.lds) prefetch(cur); }
            continue;
        }
        const int nf2 = g.width * C / 2;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float2* dst = reinterpret_cast<float2*>(srow + r * RO_MAXW * C);
#pragma unroll
            for (int k = 0; k < RO_PF; ++k) { const int i = lane + 64 * k; if (i < nf2) dst[i] = pf[r][k]; }
        }
        if (sidx + step < nseg) { cur = seg_of(sidx + step); if (cur.lds) prefetch(cur); }
        // the wave's own LDS region: make the writes visible to all of its lanes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int ox = g.ox_first + lane;
        if (ox < W) {
            int x0, x1; float lx0, lx1;
            lin_coef(ox, sww, RS, x0, x1, lx0, lx1);
            const float* a = srow + (x0 - g.sx) * C;
            const float* b = srow + (x1 - g.sx) * C;
            const float* cc = a + RO_MAXW * C;
            const float* dd = b + RO_MAXW * C;
            float* o = y + (size_t)g.img * C * H * W + (size_t)g.oy * W + ox;
            for (int c = 0; c < C; ++c)
                o[(size_t)c * H * W] = g.ly0 * (lx0 * a[c] + lx1 * b[c]) + g.ly1 * (lx0 * cc[c] + lx1 * dd[c]);
        }
        __builtin_amdgcn_wave_barrier();                 // LDS region is reused by the next segment
    }
}

// ---- host-side model ---------------------------------------------------------------------------------
struct Phase {                       // one launch of the implicit GEMM
    int ntaps; signed char offy[16], offx[16];
    int py, px, Hp, Wp;              // Hp/Wp resolved against the layer's Hout/Wout
    size_t w_off;                    // float offset into the packed-weight blob
    size_t ws_off = 0;               // the same weights split into bf16 hi / lo halves per 32-wide k-tile (bf16x3 mode; 0 = none)
    size_t wh_off = 0;               // ... into float16 hi / lo halves (f16x3 mode), pre-multiplied by 1 / wh_scale
    float wh_scale = 1.f;            // power of two
    size_t w3_off = 0;               // ... into THREE bf16 pieces per 32-wide k-tile [32 hi | 32 mid | 32 lo] (bf16x9 / bf16x6: w = hi + mid + lo exactly)
    int K;
};

struct Layer {
    std::string name;                // reference block name ("conv4", "deconv3rgb", "deconv1f" ...)
    int kind;                        // 0 conv, 1 deconv, 2 head
    int cin, cout, k, stride, pad;
    int cout_pad;
    std::vector<Phase> phases;
    size_t bias_off;                 // heads
};

struct Buf { std::string name; int H, C; size_t off /*floats per image*/, ss_off, gb_off; };

}  // namespace

struct RelposeSCNet {
    int S, use_tanh, cf;
    // constructor variants (mymodel.py:145-149): BatchNorm after every conv (else a conv bias), skip concatenations in the decoder, and which of the
    // five heads {rgb, n, d, s, f} exist (bit m of omask).  The evaluation.py configuration is bn = skip = 1, omask = 31; anything else is a
    // `variant()`: same kernels, plain plans only (no level-0 / self-stream-cache / pose-output plans)
    int bn = 1, skip = 1, omask = 31;
    bool variant() const { return !(bn && skip && omask == 31); }
    std::map<std::string, std::vector<float>> params;   // raw state dict (host)
    std::map<std::string, std::vector<int64_t>> shapes;
    bool finalized = false;
    int prec = 0;                    // RELPOSE_PREC_F32 / RELPOSE_PREC_BF16X3 (relpose_scnet_set_precision)
    int64_t nparams = 0;
    // device
    float* d_w = nullptr;        // packed weights + biases
    float* d_gb = nullptr;       // gamma/beta per activation buffer [2][C]
    float2* d_ident = nullptr;   // 16 x {1,0}: scale/shift of the raw network input
    size_t w1_off = 0;           // conv1 direct-kernel weights inside d_w
    size_t wh_off = 0, bh_off = 0;   // fused-heads weight image and bias vector inside d_w
    std::map<std::pair<void*, int>, void*> plans;   // (workspace, n) -> Plan* (each with its own device descriptor table)
    struct SelfState { uint64_t tag = 0, gen = 0; int n = 0, H = 0, W = 0; bool pose_only = false; };
    std::map<void*, SelfState> self_state;          // workspace -> whose self-view streams it holds (relpose_scnet_forward4)
    // workspace -> the forward whose RELPOSE_FWD_PART_FRONT half has been enqueued and whose _BACK half has not
    struct PendingFront { void* plan = nullptr; SelfState st; const float* x = nullptr; float* out = nullptr; int flags = 0; };
    std::map<void*, PendingFront> pending;
    std::map<std::string, Layer> layers;
    std::map<std::string, Buf> bufs;
    size_t per_image_floats = 0, ss_float2_per_group = 0;
    size_t gb_floats = 0;
    // timing
    int last_n = 0;
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_kind;
};

namespace {

struct LayerSpec { const char* name; int kind, cin, cout, k, s, p; };

std::vector<LayerSpec> layer_specs(const RelposeSCNet* net) {
    std::vector<LayerSpec> t;
    const int g = 64, S = net->S;
    const int sm = net->skip ? 2 : 1;                                  // skip_multiplier (mymodel.py:149)
    t.push_back({"conv1", 0, 16, 192, 3, 1, 1});                       // fused conv1{rgb,n,d} x {self,t2s}
    const char* mods[3] = {"rgb", "n", "d"};
    static std::string names[64];
    int ni = 0;
    for (int m = 0; m < 3; ++m) {
        names[ni] = std::string("conv2") + mods[m]; t.push_back({names[ni++].c_str(), 0, g / 2, g, 4, 2, 1});
        names[ni] = std::string("conv3") + mods[m]; t.push_back({names[ni++].c_str(), 0, g, g * 2, 4, 2, 1});
    }
    t.push_back({"conv4", 0, g * 12, g * 4, 4, 2, 1});
    t.push_back({"conv5", 0, g * 4, g * 8, 4, 2, 1});
    t.push_back({"conv6", 0, g * 8, g * 8, 4, 2, 1});
    t.push_back({"conv7", 0, g * 8, g * 8, 3, 2, 0});
    t.push_back({"conv8", 0, g * 8, g * 8, 3, 1, 1});
    t.push_back({"conv9", 0, g * 8, g * 16, 3, 1, 0});
    t.push_back({"deconv9", 1, g * 16, g * 8, 3, 1, 0});
    t.push_back({"deconv8", 1, g * 8 * sm, g * 8, 3, 1, 1});
    t.push_back({"deconv7", 1, g * 8 * sm, g * 8, 3, 2, 0});
    t.push_back({"deconv6", 1, g * 8 * sm, g * 8, 4, 2, 1});
    t.push_back({"deconv5", 1, g * 8 * sm, g * 4, 4, 2, 1});
    t.push_back({"deconv4", 1, g * 4 * sm, g * 2, 4, 2, 1});
    const int hc[5] = {3, 3, 1, S, 32};
    const char* heads[5] = {"rgb", "n", "d", "s", "f"};
    for (int m = 0; m < 5; ++m) {
        if (!((net->omask >> m) & 1)) continue;                        // (a head that was not constructed has no parameters: mymodel.py:189-243)
        const bool skip = m < 3;                                       // (rgb / n / d exist with skip connections only: relpose_scnet_create_ex)
        names[ni] = std::string("deconv3") + heads[m]; t.push_back({names[ni++].c_str(), 1, skip ? g * 4 : g * 2, g, 4, 2, 1});
        names[ni] = std::string("deconv2") + heads[m]; t.push_back({names[ni++].c_str(), 1, skip ? g * 2 : g, skip ? g / 2 : g, 4, 2, 1});
        names[ni] = std::string("deconv1") + heads[m]; t.push_back({names[ni++].c_str(), 2, g, hc[m], 1, 1, 0});
    }
    return t;
}

// index in {rgb, n, d, s, f} of the head a decoder block belongs to ("deconv3n" -> 1), -1 for the shared trunk
int head_of(const std::string& block) {
    if (block.compare(0, 7, "deconv3") && block.compare(0, 7, "deconv2") && block.compare(0, 7, "deconv1")) return -1;
    const std::string h = block.substr(7);
    const char* heads[5] = {"rgb", "n", "d", "s", "f"};
    for (int m = 0; m < 5; ++m) if (h == heads[m]) return m;
    return -1;
}

// input channels of the resized net input used by conv1 block q = 2*m + s  (mymodel.py:264-286)
void conv1_inputs(int m, int s, int* ch, int& n) {
    const int off = s * 8;
    if (m == 0) { ch[0] = off + 0; ch[1] = off + 1; ch[2] = off + 2; ch[3] = off + 7; n = 4; }
    else if (m == 1) { ch[0] = off + 3; ch[1] = off + 4; ch[2] = off + 5; ch[3] = off + 7; n = 4; }
    else { ch[0] = off + 6; ch[1] = off + 7; n = 2; }
}

bool have(RelposeSCNet* net, const std::string& key, size_t numel) {
    auto it = net->params.find(key);
    if (it == net->params.end() || it->second.size() != numel) {
        fprintf(stderr, "relpose_scnet: missing or mis-sized parameter %s (want %zu)\n", key.c_str(), numel);
        return false;
    }
    return true;
}

// round-to-nearest-even float32 -> bfloat16 (what v_cvt_pk_bf16_f32 does on the device side of the split)
inline uint16_t f32_to_bf16(float v) {
    uint32_t u; memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);       // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h) { const uint32_t u = (uint32_t)h << 16; float v; memcpy(&v, &u, 4); return v; }

int pack_layer(RelposeSCNet* net, const LayerSpec& sp, std::vector<float>& blob) {
    Layer L;
    L.name = sp.name; L.kind = sp.kind; L.cin = sp.cin; L.cout = sp.cout; L.k = sp.k; L.stride = sp.s; L.pad = sp.p;
    L.cout_pad = sp.cout >= 128 ? (sp.cout + 127) / 128 * 128 : (sp.cout > 32 ? 64 : 32);
    L.bias_off = 0;
    const int k = sp.k;
    if (L.name == "conv1") {
        // block-sparse fusion of conv1rgb/conv1n/conv1d over both streams: zero weights outside each block
        Phase P; P.ntaps = 9; P.py = P.px = 0; P.K = 9 * 16; P.w_off = blob.size();
        for (int t = 0; t < 9; ++t) { P.offy[t] = (signed char)(t / 3 - 1); P.offx[t] = (signed char)(t % 3 - 1); }
        blob.resize(blob.size() + (size_t)L.cout_pad * P.K, 0.f);
        float* w = blob.data() + P.w_off;
        const char* mods[3] = {"rgb", "n", "d"};
        for (int m = 0; m < 3; ++m) {
            const std::string key = std::string("conv1") + mods[m] + ".0.weight";
            const int cin_m = m == 2 ? 2 : 4;
            if (!have(net, key, (size_t)32 * cin_m * 9)) return RELPOSE_EINVAL;
            const float* W = net->params[key].data();
            for (int s = 0; s < 2; ++s) {
                int ch[4], n; conv1_inputs(m, s, ch, n);
                for (int co = 0; co < 32; ++co)
                    for (int ci = 0; ci < n; ++ci)
                        for (int t = 0; t < 9; ++t)
                            w[(size_t)((2 * m + s) * 32 + co) * P.K + t * 16 + ch[ci]] = W[((size_t)co * cin_m + ci) * 9 + t];
            }
        }
        L.phases.push_back(P);
    } else if (sp.kind == 0 || sp.kind == 2) {
        const std::string key = L.name + (sp.kind == 0 ? ".0.weight" : ".weight");
        if (!have(net, key, (size_t)sp.cout * sp.cin * k * k)) return RELPOSE_EINVAL;
        const float* W = net->params[key].data();
        Phase P; P.ntaps = k * k; P.py = P.px = 0; P.K = k * k * sp.cin; P.w_off = blob.size();
        for (int t = 0; t < k * k; ++t) { P.offy[t] = (signed char)(t / k - sp.p); P.offx[t] = (signed char)(t % k - sp.p); }
        blob.resize(blob.size() + (size_t)L.cout_pad * P.K, 0.f);
        float* w = blob.data() + P.w_off;
        for (int co = 0; co < sp.cout; ++co)
            for (int ci = 0; ci < sp.cin; ++ci)
                for (int t = 0; t < k * k; ++t) w[(size_t)co * P.K + t * sp.cin + ci] = W[((size_t)co * sp.cin + ci) * k * k + t];
        L.phases.push_back(P);
        if (sp.kind == 2) {
            if (!have(net, L.name + ".bias", sp.cout)) return RELPOSE_EINVAL;
            L.bias_off = blob.size();
            blob.resize(blob.size() + L.cout_pad, 0.f);
            memcpy(blob.data() + L.bias_off, net->params[L.name + ".bias"].data(), sp.cout * sizeof(float));
        }
    } else {
        // ConvTranspose2d weight [Cin][Cout][k][k]:  y = iy*s - p + ky.  Phase (py,px) = output pixels
        // congruent to (py,px) mod s; its taps are the ky with (py + p - ky) % s == 0, dy = (py+p-ky)/s.
        const std::string key = L.name + ".0.weight";
        if (!have(net, key, (size_t)sp.cin * sp.cout * k * k)) return RELPOSE_EINVAL;
        const float* W = net->params[key].data();
        for (int py = 0; py < sp.s; ++py)
            for (int px = 0; px < sp.s; ++px) {
                Phase P; P.py = py; P.px = px; P.ntaps = 0; P.w_off = blob.size();
                int kys[16], kxs[16];
                for (int ky = 0; ky < k; ++ky) {
                    if (((py + sp.p - ky) % sp.s + sp.s) % sp.s) continue;
                    for (int kx = 0; kx < k; ++kx) {
                        if (((px + sp.p - kx) % sp.s + sp.s) % sp.s) continue;
                        const int t = P.ntaps++;
                        kys[t] = ky; kxs[t] = kx;
                        P.offy[t] = (signed char)((py + sp.p - ky) / sp.s);       // exact division
                        P.offx[t] = (signed char)((px + sp.p - kx) / sp.s);
                    }
                }
                P.K = P.ntaps * sp.cin;
                blob.resize(blob.size() + (size_t)L.cout_pad * P.K, 0.f);
                float* w = blob.data() + P.w_off;
                for (int co = 0; co < sp.cout; ++co)
                    for (int t = 0; t < P.ntaps; ++t)
                        for (int ci = 0; ci < sp.cin; ++ci)
                            w[(size_t)co * P.K + t * sp.cin + ci] = W[(((size_t)ci * sp.cout + co) * k + kys[t]) * k + kxs[t]];
                L.phases.push_back(P);
            }
    }
    // bf16x3 mode: every [CoutPad][K] row re-packed k-tile by k-tile as [32 x bf16 hi | 32 x bf16 lo] (w ~= hi + lo to
    // 2^-16; same 128 bytes per k-tile, so the B loader and the LDS geometry are those of the fp32 kernel)
    if (L.name != "conv1")
        for (Phase& P : L.phases) {
            if (P.K % 32) continue;
            P.ws_off = blob.size();
            blob.resize(blob.size() + (size_t)L.cout_pad * P.K, 0.f);
            const float* w = blob.data() + P.w_off;
            uint16_t* ws = reinterpret_cast<uint16_t*>(blob.data() + P.ws_off);
            for (int n_ = 0; n_ < L.cout_pad; ++n_)
                for (int kt = 0; kt < P.K / 32; ++kt)
                    for (int e = 0; e < 32; ++e) {
                        const float v = w[(size_t)n_ * P.K + kt * 32 + e];
                        const uint16_t hi = f32_to_bf16(v);
                        const uint16_t lo = f32_to_bf16(v - bf16_to_f32(hi));
                        uint16_t* o = ws + ((size_t)n_ * P.K + kt * 32) * 2;
                        o[e] = hi; o[32 + e] = lo;
                    }
            // float16 split (f16x3 mode): 11 + 11 mantissa bits, subnormal lo halves keep an absolute error <= 2^-25
            P.wh_off = blob.size();
            blob.resize(blob.size() + (size_t)L.cout_pad * P.K, 0.f);
            w = blob.data() + P.w_off;
            _Float16* wh = reinterpret_cast<_Float16*>(blob.data() + P.wh_off);
            // exact power-of-two pre-scale so that max |w| lands in [512, 1024): the lo halves (~2^-12 |w|) of all but
            // the tiniest weights are then NORMAL float16 numbers; the kernel multiplies the accumulators by wh_scale
            float wmax = 0.f;
            for (size_t i = 0; i < (size_t)L.cout_pad * P.K; ++i) wmax = std::max(wmax, fabsf(w[i]));
            int ex = 0;
            if (wmax > 0.f) { (void)frexpf(wmax, &ex); ex = 10 - ex; }        // wmax * 2^ex in [512, 1024)
            ex = std::max(-14, std::min(ex, 24));
            const float up = ldexpf(1.f, ex);
            P.wh_scale = ldexpf(1.f, -ex);
            for (int n_ = 0; n_ < L.cout_pad; ++n_)
                for (int kt = 0; kt < P.K / 32; ++kt)
                    for (int e = 0; e < 32; ++e) {
                        const float v = w[(size_t)n_ * P.K + kt * 32 + e] * up;
                        const _Float16 hi = (_Float16)v;
                        const _Float16 lo = (_Float16)(v - (float)hi);
                        _Float16* o = wh + ((size_t)n_ * P.K + kt * 32) * 2;
                        o[e] = hi; o[32 + e] = lo;
                    }
            // three-piece bf16 split (bf16x9 / bf16x6): 8 + 8 + 8 significand bits, w = hi + mid + lo EXACTLY (both remainders are exact
            // fp32 differences; bf16 has the fp32 exponent range, so no pre-scale); 192 bytes per k-tile and row
            P.w3_off = blob.size();
            blob.resize(blob.size() + (size_t)L.cout_pad * P.K * 3 / 2, 0.f);
            w = blob.data() + P.w_off;
            uint16_t* w3 = reinterpret_cast<uint16_t*>(blob.data() + P.w3_off);
            for (int n_ = 0; n_ < L.cout_pad; ++n_)
                for (int kt = 0; kt < P.K / 32; ++kt)
                    for (int e = 0; e < 32; ++e) {
                        const float v = w[(size_t)n_ * P.K + kt * 32 + e];
                        const uint16_t hi = f32_to_bf16(v);
                        const float r1 = v - bf16_to_f32(hi);
                        const uint16_t mid = f32_to_bf16(r1);
                        const uint16_t lo = f32_to_bf16(r1 - bf16_to_f32(mid));
                        uint16_t* o = w3 + ((size_t)n_ * P.K + kt * 32) * 3;
                        o[e] = hi; o[32 + e] = mid; o[64 + e] = lo;
                    }
        }
    net->layers[L.name] = L;
    return 0;
}

struct BufSpec { const char* name; int H, C; };
const BufSpec BUFS[] = {{"X0", 224, 16},  {"A1", 224, 192}, {"A2", 112, 384}, {"A3", 56, 768},  {"A4", 28, 256}, {"A5", 14, 512},
                        {"A6", 7, 512},   {"A7", 3, 512},   {"A8", 3, 512},   {"A9", 1, 1024},  {"D9", 3, 512},  {"D8", 3, 512},
                        {"D7", 7, 512},   {"D6", 14, 512},  {"D5", 28, 256},  {"D4", 56, 128},  {"D3", 112, 320}, {"D2", 224, 224},
                        {"OUT", 224, 0}};

// gamma/beta source blocks of each activation buffer: list of (bn key prefix, channels)
std::vector<std::pair<std::string, int>> bn_blocks(const std::string& b) {
    typedef std::pair<std::string, int> P;
    if (b == "A1") return {P("conv1rgb", 32), P("conv1rgb", 32), P("conv1n", 32), P("conv1n", 32), P("conv1d", 32), P("conv1d", 32)};
    if (b == "A2") return {P("conv2rgb", 64), P("conv2rgb", 64), P("conv2n", 64), P("conv2n", 64), P("conv2d", 64), P("conv2d", 64)};
    if (b == "A3") return {P("conv3rgb", 128), P("conv3rgb", 128), P("conv3n", 128), P("conv3n", 128), P("conv3d", 128), P("conv3d", 128)};
    if (b == "A4") return {P("conv4", 256)};
    if (b == "A5") return {P("conv5", 512)};
    if (b == "A6") return {P("conv6", 512)};
    if (b == "A7") return {P("conv7", 512)};
    if (b == "A8") return {P("conv8", 512)};
    if (b == "A9") return {P("conv9", 1024)};
    if (b == "D9") return {P("deconv9", 512)};
    if (b == "D8") return {P("deconv8", 512)};
    if (b == "D7") return {P("deconv7", 512)};
    if (b == "D6") return {P("deconv6", 512)};
    if (b == "D5") return {P("deconv5", 256)};
    if (b == "D4") return {P("deconv4", 128)};
    if (b == "D3") return {P("deconv3rgb", 64), P("deconv3n", 64), P("deconv3d", 64), P("deconv3s", 64), P("deconv3f", 64)};
    if (b == "D2") return {P("deconv2rgb", 32), P("deconv2n", 32), P("deconv2d", 32), P("deconv2s", 64), P("deconv2f", 64)};
    return {};
}

// ---- static launch plan ------------------------------------------------------------------------------
// Everything about a forward except the in/out pointers is fixed by (n, workspace): the ConvDesc table is
// built once, uploaded, and the forward replays the op list.  Independent convs of equal tile shape
// (sub-pixel phases of a transposed conv, the six shared-weight encoder streams, parallel heads) are
// merged into ONE grid (blockIdx.z = member) so the 256 CUs see thousands of tiles per launch instead
// of a few hundred (wave quantisation); layers with few output tiles are split along K.
enum { OP_CONV = 0, OP_REDUCE = 1, OP_STATS = 2, OP_CONV1 = 3, OP_STATS_FUSED = 4, OP_HEADS = 5, OP_DECONV_TILE = 6, OP_CONV_S2 = 7, OP_CONV_STRIP = 8, OP_BCAST = 9, OP_NOP = 10, OP_SS_FILL = 11 };
struct Op { int type; int first, count, cfg; dim3 grid; std::string buf; int sslds = 0, uni = 0, split = 0; int ninner = 1, mt_max = 1; int skip_blk = 0; };

struct Plan {
    int n = 0; void* ws = nullptr;
    bool zero_warp = false;      // RELPOSE_FWD_ZERO_WARP plan
    bool pose_only = false;      // RELPOSE_FWD_POSE_OUTPUTS plan
    bool self_cached = false;    // self-stream cache plan (relpose_scnet_forward4): the self-view encoder streams are not recomputed
    size_t persist_floats = 0;   // conv4's split-K partial sums (a region of their own: the self slices outlive the forward)
    size_t snap_floats = 0;      // accumulator snapshots of the skip-connection halves of deconv3 / deconv2 (ConvDesc::snap) and the heads
    size_t heads_snap_off = 0;
    int snap_mode = 0;           // 0 = plain forward, 1 = fills the snapshots (a tagged forward), 2 = self-cached forward
    std::vector<ConvDesc> descs;
    std::vector<Op> ops;
    size_t splitk_floats = 0;
    size_t stat_doubles = 0;     // per-tile BatchNorm records of all fused-statistics groups
    ConvDesc* d_descs = nullptr; // device copy of `descs`
    int tail_first = -1;         // first op of the forward's tail (the heads; resize_out follows): relpose_scnet_forward2 moves it to a second stream
    int head_count = 0;          // ops of the forward's head (conv1 + its BatchNorm finalize; resize_in precedes): second stream as well
    hipEvent_t tail_ev = nullptr, head_ev = nullptr;
    // the bottleneck chain (conv4's split-K reduction .. deconv6's BatchNorm finalize): ops [mid_first, mid_end) may run on a third stream, and a
    // forward may be enqueued in two calls cut at mid_end (RELPOSE_FWD_PART_FRONT / _BACK)
    int mid_first = -1, mid_end = -1;
    hipEvent_t mid_ev0 = nullptr, mid_ev1 = nullptr;
};

struct Builder {
    RelposeSCNet* net; int n, G;
    float* act; float2* ss; float* splitk; double* statp;   // may be null for a sizing dry run
    int pend_first = -1, pend_count = 0, pend_bm = 0; bool pend_ok = true;   // conv groups since the last stats() call
    int pend_reduce = -1, pend_groups = 0;   // ... the split-K reduce op of the (only) producer group, number of producer groups
    Plan* plan;
    int rc = 0;
    int group_first = -1;
    bool zero_warp = false;     // RELPOSE_FWD_ZERO_WARP plan
    bool pose_only = false;     // RELPOSE_FWD_POSE_OUTPUTS plan: no rgb / semantic decoder branches
    bool self_cached = false;   // self-stream cache plan: only the warped-view members of conv1 / conv2 / conv3 and K slices of conv4
    float* persist = nullptr;   // conv4's split-K partial sums
    float* snapbuf = nullptr;   // accumulator snapshots (ConvDesc::snap), `snap_mode` as in Plan
    int skip_slices = 0;        // (conv4, strip kernel) K slices left from the forward that filled the cache
    int snap_mode = 0;
    int force_ksplit = 0, shared_slices = 0;   // conv4: 6 K slices = the six 128-channel stream blocks of A3 (in EVERY plan: same numerics);
                                                // zero-warp plans mark the warped streams' slices shared (ConvDesc::shared_slices)
    int nimg = 0;               // images of the members added by conv() (0 = n): RELPOSE_FWD_ZERO_WARP plans run the warped streams on 2

    float* buf(const std::string& b);
    float2* ssb(const std::string& b);
    Src src(const std::string& b, int choff, int C);
    void begin_group() { group_first = (int)plan->descs.size(); }
    void end_group();
    void conv(const std::string& layer, Src s0, const Src* s1, int Hin, const std::string& out, int ochoff);
    void stats(const std::string& b);
};

size_t partial_doubles(int G) { return (size_t)G * std::max(64 * 1024 * 2, C1_PASSES_PER_GROUP * 192 * 2); }
constexpr int MAX_DESCS = 256;

void Builder::stats(const std::string& b) {
    Op o; o.buf = b; o.cfg = 0;
    if (!net->bn) {
        // batchnorm = 0: nothing to measure -- the buffer's {scale, shift} are the constants {1, conv bias} (relpose_scnet_finalize)
        if (pend_first >= 0) for (int i = pend_first; i < pend_first + pend_count; ++i) plan->descs[i].stat_part = nullptr;
        o.type = OP_SS_FILL; o.first = o.count = 0;
        plan->ops.push_back(o);
        pend_first = -1; pend_count = 0; pend_reduce = -1; pend_groups = 0;
        return;
    }
    if (pend_first >= 0 && pend_ok) { o.type = OP_STATS_FUSED; o.first = pend_first; o.count = pend_count; o.cfg = pend_bm; }
    else {
        o.type = OP_STATS; o.first = o.count = 0;
        // a mixed / ineligible producer set: drop the per-tile records again (they would be written for nothing)
        if (pend_first >= 0) for (int i = pend_first; i < pend_first + pend_count; ++i) plan->descs[i].stat_part = nullptr;
        static const bool no_rs = RP_ENV("RELPOSE_NO_REDUCE_STATS") != nullptr;
        if (!no_rs && pend_groups == 1 && pend_reduce >= 0) {
            // one split-K producer group: its reduce kernel also leaves the BatchNorm partial sums (chunk count fixed per layer,
            // independent of the batch: results stay bitwise batch-invariant)
            Op& r = plan->ops[pend_reduce];
            const ConvDesc& d = plan->descs[r.first];
            const int R = 2 * d.Hp * d.Wp;
            // one or two rows per row-lane of a workgroup (the sum over up to 128 K slices is the long loop), bounded by the record buffer
            const int nrl = std::max(1, 256 / (d.cout_pad / 4));
            const int cap = std::max(1, std::min(128, 65536 / (net->bufs[b].C * r.count)));
            int nch = std::min(cap, std::max(1, (R + nrl - 1) / nrl));
            const int chunk_rows = (R + nch - 1) / nch;
            nch = (R + chunk_rows - 1) / chunk_rows;
            if ((size_t)nch * r.count * net->bufs[b].C * 2 <= (size_t)64 * 1024 * 2 && d.cout_pad <= 1024) {
                r.cfg = 1; r.ninner = nch; r.grid = dim3(nch, G, r.count); r.buf = b;
                o.cfg = 3; o.count = nch * r.count;          // finalize only, from the reduce kernel's records
            }
        }
    }
    plan->ops.push_back(o);
    pend_first = -1; pend_count = 0; pend_reduce = -1; pend_groups = 0;
}

float* Builder::buf(const std::string& b) { return act ? act + net->bufs[b].off * n : nullptr; }
float2* Builder::ssb(const std::string& b) { return ss ? ss + net->bufs[b].ss_off * G : nullptr; }

Src Builder::src(const std::string& b, int choff, int C) {
    const Buf& B = net->bufs[b];
    Src r; r.x = buf(b) + choff; r.cstride = B.C; r.C = C; r.sstride = B.C;
    if (b == "X0") { r.ss = net->d_ident; r.sstride = 0; r.slope = 1.f; }
    else { r.ss = ssb(b) + choff; r.slope = LRELU; }
    return r;
}

void Builder::conv(const std::string& layer, Src s0, const Src* s1, int Hin, const std::string& out, int ochoff) {
    if (rc) return;
    const Layer& L = net->layers[layer];
    const Buf& O = net->bufs[out];
    const int Hout = O.H;
    for (const Phase& P : L.phases) {
        ConvDesc d;
        memset(&d, 0, sizeof(d));
        d.src[0] = s0; d.nsrc = 1;
        if (s1) { d.src[1] = *s1; d.nsrc = 2; }
        d.Cin = s0.C + (s1 ? s1->C : 0);
        d.Nimg = nimg > 0 ? nimg : n; d.Hin = Hin; d.Win = Hin;
        if (L.kind == 1) {
            d.sy = d.sx = 1; d.osy = d.osx = L.stride; d.py = P.py; d.px = P.px;
            d.Hp = (Hout - P.py + L.stride - 1) / L.stride; d.Wp = (Hout - P.px + L.stride - 1) / L.stride;
        } else {
            d.sy = d.sx = L.stride; d.osy = d.osx = 1; d.py = d.px = 0; d.Hp = Hout; d.Wp = Hout;
        }
        d.ntaps = P.ntaps;
        memcpy(d.offy, P.offy, 16); memcpy(d.offx, P.offx, 16);
        const bool f16w = (net->prec == RELPOSE_PREC_F16X3 || net->prec == RELPOSE_PREC_F16) && P.wh_off;
        d.w = net->d_w + ((net->prec == 1 && P.ws_off) ? P.ws_off : f16w ? P.wh_off : (net->prec >= RELPOSE_PREC_BF16X9 && P.w3_off) ? P.w3_off : P.w_off);
        d.wscale = f16w ? P.wh_scale : 1.f;
        d.Cout = L.cout; d.cout_pad = L.cout_pad;
        d.y = buf(out); d.Hout = Hout; d.Wout = Hout; d.ycstride = (out == "OUT") ? net->cf : O.C; d.ychoff = ochoff;
        d.bias = (L.kind == 2) ? net->d_w + L.bias_off : nullptr;
        d.tanh_out = (L.kind == 2 && layer == "deconv1f" && net->use_tanh) ? 1 : 0;
        d.M = d.Nimg * d.Hp * d.Wp; d.K = P.K;
        d.ksplit = 1; d.kt_per = d.K / BK; d.partial = nullptr;
        {   // K order: tap-inner for the 2x2-tap phases of transposed convs (RELPOSE_TAP_INNER=0 none / 2 every conv: experiments)
            static const int ti = RP_ENV("RELPOSE_TAP_INNER") ? atoi(RP_ENV("RELPOSE_TAP_INNER")) : 1;
            d.tap_inner = (ti == 2 || (ti == 1 && d.osy == 2)) ? 1 : 0;
        }
        if (d.K != d.ntaps * d.Cin || d.Cin % BK || (s1 && s0.C % BK)) { rc = RELPOSE_EINVAL; return; }
        plan->descs.push_back(d);
    }
}

// close a group: all members share cout_pad (= tile config); choose split-K from the tile count
void Builder::end_group() {
    if (rc || group_first < 0) return;
    const int first = group_first, count = (int)plan->descs.size() - first;
    group_first = -1;
    if (count <= 0) return;
    if ((int)plan->descs.size() > MAX_DESCS) { rc = RELPOSE_EINVAL; return; }
    const int cp = plan->descs[first].cout_pad;
    int big_m = 0;
    for (int i = first; i < first + count; ++i) big_m = std::max(big_m, plan->descs[i].M / plan->descs[i].Nimg * 64);   // at the nominal batch
    // tile configs: 0 = 128x128 (4 waves), 3 = 256x128 (8 waves, 4 waves/SIMD at 2 blocks/CU), 1 = 256x64, 2 = 256x32
    // (3 measured within 1 % of 0 on conv3/conv4/deconv4-6 but needs twice the split-K: off unless RELPOSE_8WAVE is set)
    static const bool tile128 = RP_ENV("RELPOSE_TILE128") != nullptr;      // experiment: 128-row tiles, 4 workgroups per CU
    int cfg = cp >= 128 ? ((big_m >= 8192 && RP_ENV("RELPOSE_8WAVE")) ? 3 : 0) : (cp == 64 ? (tile128 ? 4 : 1) : (tile128 ? 5 : 2));
    // 6 = 128 x 256 tiles (2 x 2 waves of 64 x 128: 8 accumulators per wave, 2 workgroups per CU) for Cout >= 256 (RELPOSE_TILE256N)
    static const bool tile256n = RP_ENV("RELPOSE_TILE256N") != nullptr;
    if (tile256n && cp >= 256 && cp % 256 == 0) cfg = 6;
    {   // 256-row tiles that would straddle BatchNorm groups where 128-row tiles would not: take the 128-row variant
        // (same throughput per tile shape, but it gets the uniform-group loader)
        static const bool no_auto128 = RP_ENV("RELPOSE_NO_AUTO128") != nullptr;
        bool u256 = true, u128 = true;
        for (int i = first; i < first + count; ++i) {
            const int rows = 2 * plan->descs[i].Hp * plan->descs[i].Wp;
            u256 = u256 && rows % 256 == 0; u128 = u128 && rows % 128 == 0;
        }
        if (!no_auto128 && !u256 && u128 && (cfg == 1 || cfg == 2)) cfg = cfg == 1 ? 4 : 5;
    }
    // 7 = 64 x 64 tiles (2 x 2 waves of one 32 x 32 block) for the bottleneck layers (conv7-9, deconv9-7: <= 1024 output rows at the
    // nominal batch): with 128 x 128 tiles they have 1-9 M tiles and are split 64-128 ways along K -- 3 k-tiles per workgroup, 75 MB
    // of partial sums per layer for the reduce pass to add up; 64 x 64 tiles give 4x the tiles, an 8x smaller split and partial buffer
    static const bool no_small = RP_ENV("RELPOSE_NO_TILE64") != nullptr;
    const bool small = !no_small && cfg == 0 && big_m <= 1024 && cp % 64 == 0;
    if (small) cfg = 7;
    int BMt = cfg == 7 ? 64 : ((cfg == 0 || cfg >= 4) ? 128 : 256);
    const int BNt = cfg == 7 ? 64 : (cfg == 6 ? 256 : ((cfg == 0 || cfg == 3) ? 128 : cp));
    // Fused-phase kernel (deconv_tile_kernel): the 4 phases of stride-2 4x4 transposed convs with Cout 32 / 64 whose input grid
    // tiles into 16 x 16 (Cout 32) / 8 x 16 (Cout 64) patches -- deconv2 (112 x 112); fp32 products only.
    bool dtile = false;
    int dt_cfg = -1;
    {
        static const bool no_dt = RP_ENV("RELPOSE_NO_DECONV_TILE") != nullptr;
        // variants (RELPOSE_DT_VARIANT overrides; measured at 64 images, profiles/r02_conv_experiments.txt):
        //   0: <MI 2, NI 1> 16 x 16 patches, 2 workgroups per CU      1: <1, 2> 8 x 16 patches, 2 per CU
        //   2: <1, 1> 8 x 16 patches, 3 per CU, Cout 64 as two N tiles 3: <1, 1> 4 x 56 strips of 7 waves (56-wide grids)
        static const int dt_var = RP_ENV("RELPOSE_DT_VARIANT") ? atoi(RP_ENV("RELPOSE_DT_VARIANT")) : -1;
        const int Wg = plan->descs[first].Win;
        static const bool dt_strip = RP_ENV("RELPOSE_DT_STRIP") != nullptr;   // 56-wide grids (deconv3): no gain measured (one 7-wave workgroup per CU)
        static const bool no_pair = RP_ENV("RELPOSE_DT_NO_PAIR") != nullptr;
        //   4: <1, 1> pairs of 8 x 8 patches (grids that tile into 8 x 8 only: deconv3, 56 x 56), 3 per CU, Cout 64 as two N tiles
        // (three-piece bf16 rows: the <1, 2> variant's two weight tiles would need 95 KB of LDS -- Cout 64 as two N tiles of 32 there)
        dt_cfg = Wg % 16 == 0 ? ((cp == 32 || net->prec >= RELPOSE_PREC_BF16X9) ? 2 : 1) : ((Wg == 56 && dt_strip) ? 3 : ((Wg % 8 == 0 && !no_pair) ? 4 : -1));
        if (dt_var >= 0 && dt_var <= 2 && Wg % 16 == 0 && !(dt_var == 0 && cp != 32) && !(dt_var == 1 && cp != 64)) dt_cfg = dt_var;
        const int PRt = dt_cfg == 0 ? 16 : (dt_cfg == 3 ? 4 : 8), PWt = dt_cfg == 3 ? 56 : (dt_cfg == 4 ? 8 : 16);
        dtile = !no_dt && dt_cfg >= 0 && net->prec != 1 && (cp == 32 || cp == 64) && count % 4 == 0;
        for (int i = first; i < first + count && dtile; ++i) {
            const ConvDesc& d = plan->descs[i];
            const ConvDesc& d0 = plan->descs[first + ((i - first) & ~3)];
            dtile = d.osy == 2 && d.osx == 2 && d.sy == 1 && d.ntaps == 4 && d.Hin % PRt == 0 && d.Win % PWt == 0 && d.Hp == d.Hin && d.Wp == d.Win &&
                    d.Cin <= 512 && d.src[0].sstride != 0 && !d.bias && d.src[0].x == d0.src[0].x && d.y == d0.y && d.ychoff == d0.ychoff && d.Cin == d0.Cin;
        }
        if (dtile && dt_cfg == 4) dtile = ((plan->descs[first].Hin / 8) * (Wg / 8) * 2) % 2 == 0 && n % 2 == 0;
        if (dtile) BMt = dt_cfg == 4 ? 128 : PRt * PWt;              // (4: a workgroup = two 8 x 8 patches)
    }
    // Parity-plane kernel (conv_s2_tile_kernel): 4x4 stride-2 pad-1 convs of one source with Cout 64 on 16 x 16 output patches
    // (conv2) or Cout 128 on pairs of 8 x 8 patches (conv3); fp32 products only.
    int s2_cfg = -1;
    {
        static const bool no_s2 = RP_ENV("RELPOSE_NO_CONV_S2") != nullptr;
        const ConvDesc& d0 = plan->descs[first];
        if (!no_s2 && !dtile && net->prec != 1) {
            static const bool s2_small = RP_ENV("RELPOSE_S2_SMALL") != nullptr;     // experiment: 8 x 16 patches, 4 workgroups per CU
            if (cp == 64 && d0.Hp % 16 == 0 && d0.Wp % 16 == 0) s2_cfg = s2_small ? 2 : 0;
            else if (cp == 128 && d0.Hp % 8 == 0 && d0.Wp % 8 == 0 && ((d0.Hp / 8) * (d0.Wp / 8) * 2) % 2 == 0 && n % 2 == 0) s2_cfg = 1;
        }
        for (int i = first; i < first + count && s2_cfg >= 0; ++i) {
            const ConvDesc& d = plan->descs[i];
            const bool ok = d.osy == 1 && d.osx == 1 && d.sy == 2 && d.sx == 2 && d.ntaps == 16 && d.offy[0] == -1 && d.offx[0] == -1 && d.offy[15] == 2 &&
                            d.offx[15] == 2 && d.nsrc == 1 && d.Cin <= 128 && d.src[0].sstride != 0 && !d.bias && d.Hin == 2 * d.Hp && d.Win == 2 * d.Wp &&
                            d.Hp == d0.Hp && d.Wp == d0.Wp && d.Cout == cp;
            if (!ok) s2_cfg = -1;
        }
        if (s2_cfg >= 0) BMt = s2_cfg == 0 ? 256 : 128;              // (0: 16 x 16 patches; 1: pairs of 8 x 8; 2: 8 x 16)
    }
    int max_mt = 0, min_kt = 1 << 30;
    long tiles = 0;
    for (int i = first; i < first + count; ++i) {
        ConvDesc& d = plan->descs[i];
        if (d.cout_pad != cp) { rc = RELPOSE_EINVAL; return; }
        const int mt = (d.M + BMt - 1) / BMt;
        max_mt = std::max(max_mt, mt);
        min_kt = std::min(min_kt, d.K / BK);
        // the split factor must not depend on the batch size (results are bitwise batch-invariant), so the
        // tile count is evaluated at a nominal batch of 64 images (32 scan pairs)
        tiles += (long)((d.M / d.Nimg * 64 + BMt - 1) / BMt) * (cp / BNt);
    }
    // split along K until the launch has >= ~3000 tiles (>= 4 waves of resident blocks), keeping >= 8 k-tiles per slice
    int ksplit = 1;
    // (64 x 64 tiles: ~2 workgroups per CU are enough -- these launches are latency-bound chains, not throughput -- with >= 16 k-tiles each)
    // (experiments build: RELPOSE_WANT_TILES / RELPOSE_WANT_TILES64 / RELPOSE_MIN_SLICE override the split-K rule below)
    static const long wt_env = RP_ENV("RELPOSE_WANT_TILES") ? atol(RP_ENV("RELPOSE_WANT_TILES")) : 0;
    static const long wt64_env = RP_ENV("RELPOSE_WANT_TILES64") ? atol(RP_ENV("RELPOSE_WANT_TILES64")) : 0;
    static const int ms_env = RP_ENV("RELPOSE_MIN_SLICE") ? atoi(RP_ENV("RELPOSE_MIN_SLICE")) : 0;
    // (the three-piece bf16 kernels finish a tile in about 3/4 of the fp32 kernels' time, and the split-K reduction they then wait for is the same: half the
    // split pays in the loop -- configs[1], same box, 2 x 40 steps: 3000 tiles 785.3 pairs/s, 2000: 785.2, 1500: 790.5, 1000: 789.6, 700: 789.0; the 64 x 64
    // bottleneck tiles stay at 2048: 1024 / 512 / 256 / 64 give 785.9 / 784.5 / 778.1 / 741.3 against 785.5)
    const long want_tiles = cfg == 7 ? (wt64_env > 0 ? wt64_env : 2048) : (wt_env > 0 ? wt_env : (net->prec >= RELPOSE_PREC_BF16X9 ? 1500 : 3000));
    const int min_slice = ms_env > 0 ? ms_env : 8;
    while (!dtile && s2_cfg < 0 && tiles * ksplit < want_tiles && ksplit < 64 && min_kt / (ksplit * 2) >= min_slice) ksplit *= 2;
    if (force_ksplit && !dtile && s2_cfg < 0) ksplit = force_ksplit;
    size_t pf = 0;
    for (int i = first; i < first + count; ++i) {
        ConvDesc& d = plan->descs[i];
        d.ntiles_n = cp / BNt;
        d.ksplit = ksplit;
        d.kt_per = (d.K / BK + ksplit - 1) / ksplit;
        if (ksplit > 1) {
            // conv4 (force_ksplit: its K slices are the stream blocks) keeps its partial sums in a region no other layer writes: the
            // self-view slices are re-used by the following self-cached forwards of the same workspace
            float* base = force_ksplit ? persist : splitk;
            d.partial = base ? base + pf : nullptr; pf += (size_t)ksplit * d.M * cp;
        }
    }
    if (force_ksplit) plan->persist_floats = std::max(plan->persist_floats, pf);
    else plan->splitk_floats = std::max(plan->splitk_floats, pf);
    // fused BatchNorm statistics: no split-K and every tile inside <= 2 groups (2*hw >= BM)
    bool fuse = (ksplit == 1);
    for (int i = first; i < first + count; ++i) fuse = fuse && (2 * plan->descs[i].Hp * plan->descs[i].Wp >= BMt) && !plan->descs[i].bias;
    if (pend_first < 0) { pend_first = first; pend_count = 0; pend_bm = BMt; pend_ok = true; }
    pend_count += count;
    ++pend_groups;
    pend_ok = pend_ok && fuse;               // (the groups feeding one BatchNorm may use different tile heights: ConvDesc::stat_bm)
    if (fuse) {
        for (int i = first; i < first + count; ++i) {
            ConvDesc& d = plan->descs[i];
            d.stat_bm = BMt;
            const size_t nd = (size_t)((d.M + BMt - 1) / BMt) * 2 * cp * 2;
            d.stat_part = statp ? statp + plan->stat_doubles : (double*)(uintptr_t)8;   // non-null marker in the dry run
            plan->stat_doubles += nd;
        }
    }
    if (s2_cfg >= 0) {
        Op o; o.type = OP_CONV_S2; o.first = first; o.count = count; o.cfg = s2_cfg; o.split = net->prec;
        o.grid = dim3((unsigned)(plan->descs[first].M / BMt), count, 1);
        plan->ops.push_back(o);
        return;
    }
    if (dtile) {
        Op o; o.type = OP_DECONV_TILE; o.first = first; o.count = count; o.cfg = dt_cfg; o.split = net->prec;
        o.grid = dim3((unsigned)(plan->descs[first].M / BMt), count / 4, dt_cfg >= 2 ? cp / 32 : 1);
        if (dt_cfg != 3 && dt_cfg != 1) {
            // heads with a skip source (the self-view block of A3 / A2): accumulator snapshots for the self-stream cache -- the region
            // is laid out (and its offsets are the same) in every plan; only tagged forwards write it, only self-cached ones read it
            const int mini = (dt_cfg == 0 || dt_cfg == 1) ? 2 : 1;                     // MI * NI of the variant
            for (int i = first; i < first + count; i += 4) {
                ConvDesc& d = plan->descs[i];
                if (d.nsrc != 2) continue;
                const size_t nfl = (size_t)o.grid.x * o.grid.z * 256 * 64 * mini;
                d.snap = snapbuf ? snapbuf + plan->snap_floats : nullptr;
                d.snap_mode = snapbuf ? snap_mode : 0;
                plan->snap_floats += nfl;
            }
        }
        plan->ops.push_back(o);
        return;
    }
    {   // split-K stride-2 4x4 convs of one source with Cout a multiple of 128 (conv4, conv5): the strip kernel
        static const bool no_strip = RP_ENV("RELPOSE_NO_CONV_STRIP") != nullptr;
        const ConvDesc& d = plan->descs[first];
        const int hw = d.Hp * d.Wp;
        bool ok = !no_strip && net->prec != 1 && count == 1 && ksplit > 1 && cfg == 0 && cp % 128 == 0 && d.osy == 1 && d.osx == 1 && d.sy == 2 && d.sx == 2 &&
                  d.ntaps == 16 && d.offy[0] == -1 && d.offx[0] == -1 && d.offy[15] == 2 && d.offx[15] == 2 && d.nsrc == 1 && d.src[0].sstride != 0 && !d.bias &&
                  d.Hin == 2 * d.Hp && d.Win == 2 * d.Wp && hw >= 128;
        // staged positions of a 128-pixel tile: 127 + row wraps + one image crossing + the taps' reach
        ok = ok && 127 + (127 + d.Wp - 1) / d.Wp + (d.Wp + 1) + (d.Wp + 1) + 2 <= 224;
        if (ok) {
            if (shared_slices && (d.Cin / BK) % ksplit == 0) plan->descs[first].shared_slices = shared_slices;     // (slices = whole channel ranges here)
            if (skip_slices && (d.Cin / BK) % ksplit == 0) plan->descs[first].skip_slices = skip_slices;
            Op o; o.type = OP_CONV_STRIP; o.first = first; o.count = 1; o.cfg = 0; o.split = net->prec;
            o.grid = dim3((unsigned)((d.M + 127) / 128), (cp / 128) * ksplit, 1);
            plan->ops.push_back(o);
            Op r; r.type = OP_REDUCE; r.first = first; r.count = count; r.cfg = 0; r.grid = dim3(256, 1, count);
            plan->ops.push_back(r);
            pend_reduce = (int)plan->ops.size() - 1;
            return;
        }
    }
    Op o; o.type = OP_CONV; o.first = first; o.count = count; o.cfg = cfg;
    // runs of 4 consecutive members that are the phases of one stride-2 transposed conv share their input tile
    o.ninner = 1; o.mt_max = max_mt;
    if (count % 4 == 0) {
        bool ph = true;
        for (int i = first; i < first + count; i += 4)
            for (int k = 1; k < 4; ++k)
                ph = ph && plan->descs[i + k].src[0].x == plan->descs[i].src[0].x && plan->descs[i + k].osy == 2 && plan->descs[i].osy == 2;
        // the 4 phase tiles of one spatial tile run back to back on the same XCD and share their input lines in its
        // L2 (deconv2: -10 %, deconv3: -2 %); small layers lose to the grid padding (8 tiles), so only above 512 tiles
        static const bool no_pi = RP_ENV("RELPOSE_NO_PHASE_INTERLEAVE") != nullptr;
        if (ph && !no_pi && max_mt >= 512) o.ninner = 4;
    }
    o.grid = (o.ninner == 1) ? dim3(max_mt * count, (cp / BNt) * ksplit, 1)
                             : dim3(((max_mt + 7) / 8) * 8 * count, (cp / BNt) * ksplit, 1);
    // LDS scale/shift table: every member must fit (groups spanned by a tile) x Cin entries in 1024
    o.sslds = 1; o.uni = (cfg != 3);
    for (int i = first; i < first + count; ++i) {
        const ConvDesc& d = plan->descs[i];
        const int hw = d.Hp * d.Wp;
        const int ng = (BMt - 1) / (2 * hw) + 2;
        if ((long)ng * d.Cin > ((cfg == 0 || cfg == 3 || cfg == 6 || cfg == 7) ? 2048 : 512) || d.src[0].sstride == 0) o.sslds = 0;
        if ((2 * hw) % BMt) o.uni = 0;             // some tile would straddle two BatchNorm groups
    }
    if (!o.sslds) o.uni = 0;
    o.split = (cfg != 3) ? net->prec : 0;
    plan->ops.push_back(o);
    if (ksplit > 1) {
        Op r; r.type = OP_REDUCE; r.first = first; r.count = count; r.cfg = 0; r.grid = dim3(256, 1, count);
        plan->ops.push_back(r);
        pend_reduce = (int)plan->ops.size() - 1;
    }
}

void build_plan(RelposeSCNet* net, int n, Builder& R) {
    const char* mods[3] = {"rgb", "n", "d"};
    const char* heads[5] = {"rgb", "n", "d", "s", "f"};
    auto one = [&](const std::string& layer, Src s0, const Src* s1, int Hin, const std::string& out, int ochoff) {
        R.begin_group(); R.conv(layer, s0, s1, Hin, out, ochoff); R.end_group();
    };
    // encoder, three modalities x two streams in concatenated buffers (mymodel.py:266-291)
    { Op o; o.type = OP_CONV1; o.first = o.count = o.cfg = 0; R.plan->ops.push_back(o); }   // direct kernel
    R.stats("A1");
    if (net->bn) R.plan->ops.back().cfg = 1;      // partial records already written by conv1_direct_kernel
    if (R.self_cached) R.plan->ops.back().skip_blk = 32;
    R.plan->head_count = (int)R.plan->ops.size();
    if (R.self_cached) {
        // Self-stream cache (relpose_scnet_forward4): channels 0:8 of the input are those of the forward that filled the cache, and the
        // reference runs the self-view streams as module calls of their own with their own batch statistics (mymodel.py:266-276 vs
        // :278-288) -- so the self blocks of A1 / A2 / A3 (even q), their {scale, shift} and conv4's self K slices would come out
        // bitwise as they already are in this workspace.  Only the warped-view members run (same kernels, same tiles, same records as
        // in the full plan), the BatchNorm finalize leaves the self blocks' table entries alone, and conv4 computes its odd K slices.
        for (int L = 2; L <= 3; ++L) {
            const std::string name = L == 2 ? "conv2" : "conv3", in = L == 2 ? "A1" : "A2", out = L == 2 ? "A2" : "A3";
            const int cin = L == 2 ? 32 : 64, cout = L == 2 ? 64 : 128, Hin = L == 2 ? 224 : 112;
            R.begin_group();
            for (int q = 1; q < 6; q += 2) R.conv(name + mods[q / 2], R.src(in, q * cin, cin), nullptr, Hin, out, q * cout);
            R.end_group();
            R.stats(out);
            if (R.plan->ops.back().type != OP_STATS_FUSED) R.rc = RELPOSE_EINVAL;     // (a statistics pass over the buffer would redo the self blocks)
            R.plan->ops.back().skip_blk = cout;
        }
    } else if (!R.zero_warp) {
        R.begin_group();
        for (int q = 0; q < 6; ++q) R.conv(std::string("conv2") + mods[q / 2], R.src("A1", q * 32, 32), nullptr, 224, "A2", q * 64);
        R.end_group(); R.stats("A2");
        R.begin_group();
        for (int q = 0; q < 6; ++q) R.conv(std::string("conv3") + mods[q / 2], R.src("A2", q * 64, 64), nullptr, 112, "A3", q * 128);
        R.end_group(); R.stats("A3");
    } else {
        // level 0: the warped view is all zeros, so the warped streams (odd q) see the same input in every image -- conv1 of zeros is
        // an exact 0 everywhere -- and their conv2 / conv3 run for the first BatchNorm group (2 images) only: same kernels, same tiles,
        // same BatchNorm records as in the full plan, hence bitwise the same values; OP_BCAST then copies the conv3 outputs and their
        // {scale, shift} to the other images / groups (the A2 blocks of the warped streams are read by those conv3 members only)
        for (int L = 2; L <= 3; ++L) {
            const std::string name = L == 2 ? "conv2" : "conv3", in = L == 2 ? "A1" : "A2", out = L == 2 ? "A2" : "A3";
            const int cin = L == 2 ? 32 : 64, cout = L == 2 ? 64 : 128, Hin = L == 2 ? 224 : 112;
            for (int stream = 0; stream < 2; ++stream) {
                R.nimg = stream ? 2 : 0;
                R.begin_group();
                for (int q = stream; q < 6; q += 2) R.conv(name + mods[q / 2], R.src(in, q * cin, cin), nullptr, Hin, out, q * cout);
                R.end_group();
            }
            R.nimg = 0;
            R.stats(out);
            if (R.plan->ops.back().type != OP_STATS_FUSED) R.rc = RELPOSE_EINVAL;     // (the records of the 2-image members: fused statistics only)
        }
        { Op o; o.type = OP_BCAST; o.first = o.count = o.cfg = 0; o.buf = "A3"; R.plan->ops.push_back(o); }
    }
    R.force_ksplit = 6; R.shared_slices = R.zero_warp ? 0x2a : 0;      // slice ks = stream block ks of A3 (odd = warped view)
    R.skip_slices = R.self_cached ? 0x15 : 0;
    const size_t conv4_desc = R.plan->descs.size();
    one("conv4", R.src("A3", 0, 768), nullptr, 56, "A4", 0);
    // the bottleneck chain starts with conv4's split-K reduction (when it has one) and ends in front of deconv5
    R.plan->mid_first = (int)R.plan->ops.size() - ((!R.plan->ops.empty() && R.plan->ops.back().type == OP_REDUCE) ? 1 : 0);
    R.stats("A4");
    R.force_ksplit = 0; R.shared_slices = 0; R.skip_slices = 0;
    // (the strip kernel took the shared slices: nothing reads the warped blocks of A3 beyond the first image pair, no copies needed)
    if (R.zero_warp && !R.rc && R.plan->descs[conv4_desc].shared_slices)
        for (Op& o : R.plan->ops) if (o.type == OP_BCAST) o.type = OP_NOP;
    one("conv5", R.src("A4", 0, 256), nullptr, 28, "A5", 0); R.stats("A5");
    one("conv6", R.src("A5", 0, 512), nullptr, 14, "A6", 0); R.stats("A6");
    one("conv7", R.src("A6", 0, 512), nullptr, 7, "A7", 0); R.stats("A7");
    one("conv8", R.src("A7", 0, 512), nullptr, 3, "A8", 0); R.stats("A8");
    one("conv9", R.src("A8", 0, 512), nullptr, 3, "A9", 0); R.stats("A9");
    // decoder with skip concatenations (mymodel.py:302-307)
    // (skipLayer = 0, mymodel.py:335-340: the same chain without the second source)
    Src sk;
    const Src* skp = net->skip ? &sk : nullptr;
    one("deconv9", R.src("A9", 0, 1024), nullptr, 1, "D9", 0); R.stats("D9");
    sk = R.src("A8", 0, 512); one("deconv8", R.src("D9", 0, 512), skp, 3, "D8", 0); R.stats("D8");
    sk = R.src("A7", 0, 512); one("deconv7", R.src("D8", 0, 512), skp, 3, "D7", 0); R.stats("D7");
    sk = R.src("A6", 0, 512); one("deconv6", R.src("D7", 0, 512), skp, 7, "D6", 0); R.stats("D6");
    R.plan->mid_end = (int)R.plan->ops.size();
    sk = R.src("A5", 0, 512); one("deconv5", R.src("D6", 0, 512), skp, 14, "D5", 0); R.stats("D5");
    sk = R.src("A4", 0, 256); one("deconv4", R.src("D5", 0, 256), skp, 28, "D4", 0); R.stats("D4");
    // heads (mymodel.py:309-376): rgb/n/d with skips from the self stream, s/f without
    // (RELPOSE_FWD_POSE_OUTPUTS: the rgb and semantic branches feed nothing the pose path reads; their blocks of D3 / D2 stay unwritten)
    // (a net constructed without a head -- outputType, mymodel.py:189-243 -- leaves that head's blocks to the memset in front of the forward)
    auto wanted = [&](int m) { return ((net->omask >> m) & 1) && (!R.pose_only || (m != 0 && m != 3)); };
    R.begin_group();
    for (int m = 0; m < 5; ++m) {
        if (!wanted(m)) continue;
        if (m < 3) { sk = R.src("A3", 2 * m * 128, 128); R.conv(std::string("deconv3") + heads[m], R.src("D4", 0, 128), &sk, 56, "D3", m * 64); }
        else R.conv(std::string("deconv3") + heads[m], R.src("D4", 0, 128), nullptr, 56, "D3", m * 64);
    }
    R.end_group(); R.stats("D3");
    const int d2off[5] = {0, 32, 64, 96, 160};
    R.begin_group();
    for (int m = 0; m < 3; ++m) if (wanted(m)) { sk = R.src("A2", 2 * m * 64, 64); R.conv(std::string("deconv2") + heads[m], R.src("D3", m * 64, 64), &sk, 112, "D2", d2off[m]); }
    R.end_group();
    R.begin_group();
    for (int m = 3; m < 5; ++m) if (wanted(m)) R.conv(std::string("deconv2") + heads[m], R.src("D3", m * 64, 64), nullptr, 112, "D2", d2off[m]);
    R.end_group(); R.stats("D2");
    R.plan->tail_first = (int)R.plan->ops.size();
    if (RP_ENV("RELPOSE_GEMM_HEADS") || (net->S != 15 && net->S != 21)) {   // generic implicit-GEMM path (5 members)
        if (R.pose_only) R.rc = RELPOSE_EINVAL;                               // (pose-only plans need the fused heads kernel: S = 15 / 21)
        const int ooff[5] = {0, 3, 6, 7, 7 + net->S};
        R.begin_group();
        for (int m = 0; m < 5; ++m) {
            if (!wanted(m)) continue;
            if (m < 3) { sk = R.src("A1", 2 * m * 32, 32); R.conv(std::string("deconv1") + heads[m], R.src("D2", d2off[m], 32), &sk, 224, "OUT", ooff[m]); }
            else R.conv(std::string("deconv1") + heads[m], R.src("D2", d2off[m], 64), nullptr, 224, "OUT", ooff[m]);
        }
        R.end_group();
    } else {
        Op o; o.type = OP_HEADS; o.first = o.count = o.cfg = 0; R.plan->ops.push_back(o);
        R.plan->heads_snap_off = R.plan->snap_floats;              // the heads' skip-half accumulators (HeadsDesc::snap): 12 floats per pixel
        R.plan->snap_floats += (size_t)n * RS * RS * 12;
    }
    (void)n;
}

// snap_mode of the transposed conv that phase member i belongs to (the builder records it on the first of the 4 phase members)
inline int plan_snap_mode(const Plan& p, int i) {
    for (const Op& op : p.ops)
        if (op.type == OP_DECONV_TILE && i >= op.first && i < op.first + op.count) return p.descs[op.first + ((i - op.first) & ~3)].snap_mode;
    return 0;
}

void free_plan(RelposeSCNet* net) {
    for (auto& kv : net->plans) {
        Plan* p = (Plan*)kv.second;
        if (p->d_descs) (void)hipFree(p->d_descs);
        if (p->tail_ev) (void)hipEventDestroy(p->tail_ev);
        if (p->head_ev) (void)hipEventDestroy(p->head_ev);
        if (p->mid_ev0) (void)hipEventDestroy(p->mid_ev0);
        if (p->mid_ev1) (void)hipEventDestroy(p->mid_ev1);
        delete p;
    }
    net->plans.clear();
    net->self_state.clear();       // (new weights / precision: nothing cached is valid)
    net->pending.clear();          // (a half-enqueued forward loses its plan: its BACK call returns RELPOSE_EINVAL)
}

struct WsOffsets { size_t act, ss, partial, splitk, statp, persist, snap, total; };

WsOffsets ws_offsets(RelposeSCNet* net, int n) {
    Plan dry;
    Builder B; B.net = net; B.n = n; B.G = n / 2; B.act = nullptr; B.ss = nullptr; B.splitk = nullptr; B.statp = nullptr; B.plan = &dry;
    build_plan(net, n, B);
    WsOffsets o;
    size_t off = 0;
    o.act = off; off += rp_align(net->per_image_floats * n * sizeof(float));
    o.ss = off; off += rp_align(net->ss_float2_per_group * (n / 2) * sizeof(float2));
    o.partial = off; off += rp_align(partial_doubles(n / 2) * sizeof(double));
    o.splitk = off; off += rp_align(dry.splitk_floats * sizeof(float));
    o.statp = off; off += rp_align(dry.stat_doubles * sizeof(double));
    o.persist = off; off += rp_align(dry.persist_floats * sizeof(float));
    o.snap = off; off += rp_align(dry.snap_floats * sizeof(float));
    o.total = off;
    return o;
}

}  // namespace

extern "C" {

RelposeSCNet* relpose_scnet_create(int32_t snumclass, int32_t use_tanh) {
    if (snumclass < 1 || snumclass > 64) return nullptr;
    RelposeSCNet* net = new RelposeSCNet();
    net->S = snumclass; net->use_tanh = use_tanh; net->cf = 7 + snumclass + 32;
    return net;
}

RelposeSCNet* relpose_scnet_create_ex(const RelposeSCNetConfig* cfg) {
    if (!cfg || cfg->struct_size < sizeof(RelposeSCNetConfig)) return nullptr;
    const int om = cfg->output_mask;
    if (om <= 0 || om > 31) return nullptr;
    // without skip connections the reference can only build the s / f heads: its 1x1 rgb / n / d output convs always take 64 channels, 32 of them
    // the skip (mymodel.py:192,200,208 vs :344-360 -- that combination raises inside torch); 'k' reads an undefined xsift (:328) in either mode
    if (!cfg->skip_layer && (om & (RELPOSE_OUT_RGB | RELPOSE_OUT_N | RELPOSE_OUT_D))) return nullptr;
    RelposeSCNet* net = relpose_scnet_create(cfg->snumclass, cfg->use_tanh);
    if (!net) return nullptr;
    net->bn = cfg->batchnorm ? 1 : 0; net->skip = cfg->skip_layer ? 1 : 0; net->omask = om;
    return net;
}

void relpose_scnet_destroy(RelposeSCNet* net) {
    if (!net) return;
    if (net->d_w) (void)hipFree(net->d_w);
    if (net->d_gb) (void)hipFree(net->d_gb);
    if (net->d_ident) (void)hipFree(net->d_ident);
    free_plan(net);
    delete net;
}

int relpose_scnet_set_param(RelposeSCNet* net, const char* key, const float* data, size_t numel) {
    if (!net || !key || !data) return RELPOSE_EINVAL;
    net->params[key] = std::vector<float>(data, data + numel);
    net->finalized = false;
    free_plan(net);                  // cached launch plans hold absolute pointers into the packed-weight blob
    return 0;
}

int64_t relpose_scnet_num_params(const RelposeSCNet* net) {
    if (!net) return 0;
    int64_t n = 0;
    for (auto& kv : net->params) n += (int64_t)kv.second.size();
    return n;
}

int relpose_scnet_set_precision(RelposeSCNet* net, int32_t mode) {
    if (!net || mode < RELPOSE_PREC_F32 || mode > RELPOSE_PREC_BF16X6) return RELPOSE_EINVAL;
    if (net->prec != mode) { net->prec = mode; free_plan(net); }     // the launch plans hold weight pointers and kernel variants
    return 0;
}

int relpose_scnet_finalize(RelposeSCNet* net) {
    if (!net) return RELPOSE_EINVAL;
    free_plan(net);                  // descriptors of an earlier state dict point into the blob that is re-allocated below
    std::vector<float> blob;
    net->layers.clear();
    for (const LayerSpec& sp : layer_specs(net)) {
        int rc = pack_layer(net, sp, blob);
        if (rc) return rc;
    }
    {   // conv1 direct kernel weights: [6][9][4][32]; depth block: channel 0 = depth, 1 = mask, 2..3 = 0
        net->w1_off = blob.size();
        blob.resize(blob.size() + 6 * 9 * 4 * 32, 0.f);
        float* w1 = blob.data() + net->w1_off;
        const char* mods[3] = {"rgb", "n", "d"};
        for (int m = 0; m < 3; ++m) {
            const int cin_m = m == 2 ? 2 : 4;
            const float* W = net->params[std::string("conv1") + mods[m] + ".0.weight"].data();
            for (int s2 = 0; s2 < 2; ++s2)
                for (int t = 0; t < 9; ++t)
                    for (int c = 0; c < cin_m; ++c)
                        for (int o = 0; o < 32; ++o)
                            w1[(((size_t)(2 * m + s2) * 9 + t) * 4 + c) * 32 + o] = W[((size_t)o * cin_m + c) * 9 + t];
        }
    }
    {   // fused heads: weight image (see heads_kernel) + bias in output-channel order
        net->wh_off = blob.size();
        blob.resize(blob.size() + HEADS_W, 0.f);
        net->bh_off = blob.size();
        blob.resize(blob.size() + 64, 0.f);
        float* wh = blob.data() + net->wh_off;
        float* bh = blob.data() + net->bh_off;
        const char* hn[5] = {"rgb", "n", "d", "s", "f"};
        const int hc[5] = {3, 3, 1, net->S, 32};
        if (net->S > 24) return RELPOSE_EINVAL;
        int ob = 0;
        for (int m = 0; m < 5; ++m) {
            if (!((net->omask >> m) & 1)) { ob += hc[m]; continue; }       // a head that was not constructed: zero weights and bias, its channels come out 0
            const float* W = net->params[std::string("deconv1") + hn[m] + ".weight"].data();     // [Cout][64]
            const float* Bv = net->params[std::string("deconv1") + hn[m] + ".bias"].data();
            for (int o = 0; o < hc[m]; ++o) {
                bh[ob + o] = Bv[o];
                for (int ci = 0; ci < 64; ++ci) {
                    size_t idx;
                    if (m < 3) idx = (size_t)((ci < 32 ? m * 32 + ci : 96 + m * 32 + (ci - 32)) * 4 + o);
                    else if (m == 3) idx = 768 + (size_t)ci * 24 + o;
                    else idx = 2304 + (size_t)ci * 32 + o;
                    wh[idx] = W[(size_t)o * 64 + ci];
                }
            }
            ob += hc[m];
        }
    }
    // activation buffers
    net->bufs.clear();
    size_t off = 0, ssoff = 0, gboff = 0;
    std::vector<float> gb;
    for (const BufSpec& bs : BUFS) {
        Buf b; b.name = bs.name; b.H = bs.H; b.C = bs.C ? bs.C : net->cf;
        b.off = off; off += (size_t)b.H * b.H * b.C;
        b.ss_off = ssoff; b.gb_off = gboff;
        auto blocks = bn_blocks(b.name);
        if (!blocks.empty()) {
            ssoff += b.C;
            gb.resize(gboff + 2 * (size_t)b.C, 0.f);
            int c = 0;
            for (auto& blk : blocks) {
                const int hm = head_of(blk.first);
                if (hm >= 0 && !((net->omask >> hm) & 1)) { c += blk.second; continue; }   // no such head: {gamma, beta} = 0, the block's activations read as 0
                if (net->bn) {
                    if (!have(net, blk.first + ".1.weight", blk.second) || !have(net, blk.first + ".1.bias", blk.second)) return RELPOSE_EINVAL;
                    memcpy(gb.data() + gboff + c, net->params[blk.first + ".1.weight"].data(), blk.second * sizeof(float));
                    memcpy(gb.data() + gboff + b.C + c, net->params[blk.first + ".1.bias"].data(), blk.second * sizeof(float));
                } else {
                    // batchnorm = 0 (mymodel.py:22-25): conv + bias + LeakyReLU.  The consumer's loader computes lrelu(scale * y + shift) anyway,
                    // so the table holds {1, bias} -- 1 * y + bias is the bias add, rounded once like the reference's -- and OP_SS_FILL copies it
                    // where the BatchNorm finalize would have written {gamma / sigma, beta - mean * gamma / sigma}
                    if (!have(net, blk.first + ".0.bias", blk.second)) return RELPOSE_EINVAL;
                    for (int i = 0; i < blk.second; ++i) gb[gboff + c + i] = 1.f;
                    memcpy(gb.data() + gboff + b.C + c, net->params[blk.first + ".0.bias"].data(), blk.second * sizeof(float));
                }
                c += blk.second;
            }
            if (c != b.C) return RELPOSE_EINVAL;
            gboff += 2 * (size_t)b.C;
        }
        net->bufs[b.name] = b;
    }
    net->per_image_floats = off; net->ss_float2_per_group = ssoff; net->gb_floats = gboff;
    if (net->d_w) { (void)hipFree(net->d_w); net->d_w = nullptr; }
    if (net->d_gb) { (void)hipFree(net->d_gb); net->d_gb = nullptr; }
    RP_HIP(hipMalloc((void**)&net->d_w, blob.size() * sizeof(float)));
    RP_HIP(hipMemcpy(net->d_w, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    RP_HIP(hipMalloc((void**)&net->d_gb, gb.size() * sizeof(float)));
    RP_HIP(hipMemcpy(net->d_gb, gb.data(), gb.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!net->d_ident) {
        float2 id[16];
        for (int i = 0; i < 16; ++i) id[i] = make_float2(1.f, 0.f);
        RP_HIP(hipMalloc((void**)&net->d_ident, sizeof(id)));
        RP_HIP(hipMemcpy(net->d_ident, id, sizeof(id), hipMemcpyHostToDevice));
    }
    net->finalized = true;
    return 0;
}

size_t relpose_scnet_workspace_bytes(const RelposeSCNet* net, int32_t n, int32_t H, int32_t W) {
    if (!net || !net->finalized || n <= 0 || (n & 1) || H <= 0 || W <= 0) return 0;
    return ws_offsets(const_cast<RelposeSCNet*>(net), n).total;
}

// The four historical entry points are thin wrappers of relpose_scnet_forward_ex (one argument block, include/relpose.h).
static RelposeForwardArgs rp_fwd_args(const float* x, float* out, int32_t n, int32_t H, int32_t W, void* workspace, size_t workspace_bytes, void* stream,
                                      void* tail_stream, int32_t flags, uint64_t self_tag) {
    RelposeForwardArgs a;
    memset(&a, 0, sizeof(a));
    a.struct_size = (uint32_t)sizeof(a); a.flags = flags; a.x = x; a.out = out; a.n_images = n; a.H = H; a.W = W;
    a.workspace = workspace; a.workspace_bytes = workspace_bytes; a.stream = stream; a.tail_stream = tail_stream; a.self_tag = self_tag;
    return a;
}

int relpose_scnet_forward(RelposeSCNet* net, const float* x, float* out, int32_t n, int32_t H, int32_t W, void* workspace,
                          size_t workspace_bytes, void* stream) {
    const RelposeForwardArgs a = rp_fwd_args(x, out, n, H, W, workspace, workspace_bytes, stream, stream, 0, 0);
    return relpose_scnet_forward_ex(net, &a);
}

int relpose_scnet_forward_ex(RelposeSCNet* net, const RelposeForwardArgs* args) {
    // (fields beyond the caller's struct_size take their defaults: a caller compiled against an older header keeps working)
    if (!net || !args || args->struct_size < offsetof(RelposeForwardArgs, self_tag)) return RELPOSE_EINVAL;
    const float* x = args->x; float* out = args->out;
    const int32_t n = args->n_images, H = args->H, W = args->W, flags = args->flags;
    void* workspace = args->workspace; const size_t workspace_bytes = args->workspace_bytes;
    // a constructor variant (relpose_scnet_create_ex) runs the plain plan on one stream: the level-0 / pose-output / self-stream-cache plans and
    // the two-stream form are built for the evaluation.py configuration only, and every one of them is "bitwise the plain forward" by contract
    const bool variant = net->variant();
    void* stream = args->stream; void* tail_stream = (args->tail_stream && !variant) ? args->tail_stream : args->stream;
    const uint64_t self_tag = (!variant && args->struct_size >= offsetof(RelposeForwardArgs, self_tag) + sizeof(uint64_t)) ? args->self_tag : 0;
    const uint64_t ws_gen = args->struct_size >= offsetof(RelposeForwardArgs, workspace_generation) + sizeof(uint64_t) ? args->workspace_generation : 0;
    if (!net->finalized || !x || !out || !workspace || n <= 0 || (n & 1) || H <= 0 || W <= 0 || n >= (1 << 24)) return RELPOSE_EINVAL;
    if (flags & ~(RELPOSE_FWD_ZERO_WARP | RELPOSE_FWD_POSE_OUTPUTS | RELPOSE_FWD_NEW_WORKSPACE | RELPOSE_FWD_PART_FRONT | RELPOSE_FWD_PART_BACK)) return RELPOSE_EINVAL;
    const int part = flags & (RELPOSE_FWD_PART_FRONT | RELPOSE_FWD_PART_BACK);
    if (part == (RELPOSE_FWD_PART_FRONT | RELPOSE_FWD_PART_BACK)) return RELPOSE_EINVAL;
#ifdef RP_EXPERIMENTS
    void* mid_stream = (args->struct_size >= offsetof(RelposeForwardArgs, reserved1) + sizeof(void*) && args->reserved1) ? args->reserved1 : args->stream;
#else
    // product build: one call = one forward, the bottleneck chain stays on `stream` (the two-part / third-stream forms lost their A/Bs, round 5)
    if (part || (args->struct_size >= offsetof(RelposeForwardArgs, reserved1) + sizeof(void*) && args->reserved1)) return RELPOSE_EINVAL;
    void* mid_stream = args->stream;
#endif
    const int G = n / 2;
    // (nothing to share with one BatchNorm group; the tile kernels' patch pairing wants the 2-image members' patch count even as well)
    const bool zero_warp = (flags & RELPOSE_FWD_ZERO_WARP) && n > 2 && !variant;
    const bool pose_only = (flags & RELPOSE_FWD_POSE_OUTPUTS) != 0 && !variant;
    Plan* plan = nullptr;
    RelposeSCNet::SelfState commit;          // what the workspace holds once this forward is through
    commit.tag = self_tag; commit.n = n; commit.H = H; commit.W = W; commit.pose_only = pose_only; commit.gen = ws_gen;
    if (part == RELPOSE_FWD_PART_BACK) {
        // the second half of a forward whose FRONT call chose the plan (and invalidated the self-stream record): same arguments, or nothing runs
        auto it = net->pending.find(workspace);
        if (it == net->pending.end()) return RELPOSE_EINVAL;
        const RelposeSCNet::PendingFront pf = it->second;
        net->pending.erase(it);
        if (pf.x != x || pf.out != out || pf.flags != (flags & ~(RELPOSE_FWD_PART_BACK | RELPOSE_FWD_NEW_WORKSPACE)) || pf.st.n != n || pf.st.H != H || pf.st.W != W ||
            pf.st.tag != self_tag || pf.st.gen != ws_gen)
            return RELPOSE_EINVAL;
        plan = (Plan*)pf.plan;
    } else {
    net->pending.erase(workspace);           // (a FRONT that is never followed by its BACK: dropped by whatever forward comes next)
    if (flags & RELPOSE_FWD_NEW_WORKSPACE) net->self_state.erase(workspace);      // the memory behind this pointer is not what the last forward left
    // Self-stream cache: the previous forward on this workspace carried the same non-zero tag (and shape, and -- when the caller names its
    // allocations -- the same workspace generation) -> the self-view encoder streams it left in the workspace are what this forward would
    // compute.  Anything else runs (and re-fills) them.  The record is INVALIDATED here and committed only after this forward's kernels
    // have been enqueued: a forward that returns an error leaves no claim on the workspace's content.
    RelposeSCNet::SelfState prev = net->self_state[workspace];
    net->self_state[workspace] = RelposeSCNet::SelfState();
    // (the accumulator snapshots are laid out per plan family: a pose-outputs forward has fewer decoder heads)
    bool self_cached = self_tag != 0 && prev.tag == self_tag && prev.n == n && prev.H == H && prev.W == W && prev.pose_only == pose_only && !zero_warp &&
                       prev.gen == ws_gen;
    int snap_mode = 0;
    for (int attempt = 0; attempt < 2 && !plan; ++attempt) {
        // a tagged forward that computes the self streams also leaves the accumulator snapshots of the skip-connection halves
        snap_mode = self_cached ? 2 : (self_tag != 0 ? 1 : 0);
        const int plan_key = (int)n | (zero_warp ? 1 << 24 : 0) | (pose_only ? 1 << 25 : 0) | (self_cached ? 1 << 26 : 0) | (snap_mode == 1 ? 1 << 27 : 0);
        auto it = net->plans.find(std::make_pair(workspace, plan_key));
        if (it != net->plans.end()) { plan = (Plan*)it->second; break; }
        const WsOffsets o = ws_offsets(net, n);
        if (workspace_bytes < o.total) return RELPOSE_ENOMEM;
        if (net->plans.size() >= 32 && net->pending.empty()) {       // callers keep a few long-lived workspaces; bound the cache
            free_plan(net);
            self_cached = false;             // (free_plan drops every workspace's self-stream record with the plans)
            snap_mode = self_tag != 0 ? 1 : 0;
        }
        Plan* p = new Plan();
        p->n = n; p->ws = workspace; p->zero_warp = zero_warp; p->pose_only = pose_only; p->self_cached = self_cached;
        char* ws = (char*)workspace;
        Builder B; B.net = net; B.n = n; B.G = G; B.plan = p; B.zero_warp = zero_warp; B.pose_only = pose_only; B.self_cached = self_cached;
        B.act = (float*)(ws + o.act); B.ss = (float2*)(ws + o.ss); B.splitk = (float*)(ws + o.splitk); B.statp = (double*)(ws + o.statp);
        B.persist = (float*)(ws + o.persist);
        B.snapbuf = (float*)(ws + o.snap); B.snap_mode = snap_mode; p->snap_mode = snap_mode;
        build_plan(net, n, B);
        if (B.rc) {
            delete p;
            // a self-cached plan that cannot be built (e.g. conv2 / conv3 statistics not on the fused path in some kernel selection) is not an
            // error of the call: the full forward computes the same output and refills the cache
            if (self_cached) { self_cached = false; continue; }
            return B.rc;
        }
        RP_HIP(hipMalloc((void**)&p->d_descs, MAX_DESCS * sizeof(ConvDesc)));
        RP_HIP(hipMemcpy(p->d_descs, p->descs.data(), p->descs.size() * sizeof(ConvDesc), hipMemcpyHostToDevice));
        const int key2 = (int)n | (zero_warp ? 1 << 24 : 0) | (pose_only ? 1 << 25 : 0) | (self_cached ? 1 << 26 : 0) | (snap_mode == 1 ? 1 << 27 : 0);
        net->plans[std::make_pair(workspace, key2)] = p;
        plan = p;
    }
    }   // (part != BACK)
    if (!plan) return RELPOSE_EINVAL;
    net->last_n = n;
    // two-stream mode: the HBM-bound head (resize_in, conv1) and tail (heads, resize_out) run on tail_stream, the MFMA-bound middle on `stream`
    static const bool head_side = RP_ENV("RELPOSE_NO_HEAD_OVERLAP") == nullptr;
    const bool two = tail_stream != stream;
    hipStream_t s = (two && head_side && plan->head_count > 0) ? (hipStream_t)tail_stream : (hipStream_t)stream;
    const WsOffsets o = ws_offsets(net, n);
    char* ws = (char*)workspace;
    float* act = (float*)(ws + o.act);
    float2* ssp = (float2*)(ws + o.ss);
    double* partial = (double*)(ws + o.partial);
    auto mark = [&](int kind) {
        if (!net->profiling) return;
        hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, s);
        net->ev.push_back(e); net->ev_kind.push_back(kind);
    };
    const bool mid_split = plan->mid_first >= 0 && plan->mid_end > plan->mid_first && plan->mid_first >= plan->head_count && plan->mid_end <= plan->tail_first;
    if (part && !mid_split) return RELPOSE_EINVAL;
    if (net->omask != 31) {
        // heads the net was constructed without: their blocks of D3 / D2 (and, on the generic heads path, their channels of OUT) are written by
        // nobody -- zero them so that what the heads kernel multiplies by its zero weights is finite; their output channels come out 0
        for (const char* bn_ : {"D3", "D2", "OUT"}) {
            const Buf& B = net->bufs[bn_];
            RP_HIP(hipMemsetAsync(act + B.off * n, 0, (size_t)n * B.H * B.H * B.C * sizeof(float), s));
        }
    }
    if (part != RELPOSE_FWD_PART_BACK) {
        mark(3);
        hipLaunchKernelGGL(resize_in_kernel, dim3(2048), dim3(256), 0, s, x, act + net->bufs["X0"].off * n, n, H, W, plan->self_cached ? 8 : 0);
        mark(-3);
    } else {
        // the decoder continues on `stream` behind the chain (wherever FRONT enqueued it)
        s = (hipStream_t)stream;
        RP_HIP(hipStreamWaitEvent(s, plan->mid_ev1, 0));
    }
    int op_index = -1;
    for (const Op& op : plan->ops) {
        ++op_index;
        if (part == RELPOSE_FWD_PART_BACK && op_index < plan->mid_end) continue;
        if (mid_split && op_index == plan->mid_first && part != RELPOSE_FWD_PART_BACK && (hipStream_t)mid_stream != s) {
            if (!plan->mid_ev0) RP_HIP(hipEventCreateWithFlags(&plan->mid_ev0, hipEventDisableTiming));
            RP_HIP(hipEventRecord(plan->mid_ev0, s));
            s = (hipStream_t)mid_stream;
            RP_HIP(hipStreamWaitEvent(s, plan->mid_ev0, 0));
        }
        if (mid_split && op_index == plan->mid_end && part != RELPOSE_FWD_PART_BACK) {
            if (part == RELPOSE_FWD_PART_FRONT) {
                // the first half ends here: BACK (the next call on this workspace) orders the decoder behind this event
                if (!plan->mid_ev1) RP_HIP(hipEventCreateWithFlags(&plan->mid_ev1, hipEventDisableTiming));
                RP_HIP(hipEventRecord(plan->mid_ev1, s));
                RP_CHECK_LAUNCH();
                RelposeSCNet::PendingFront pf;
                pf.plan = plan; pf.st = commit; pf.x = x; pf.out = out; pf.flags = flags & ~(RELPOSE_FWD_PART_FRONT | RELPOSE_FWD_NEW_WORKSPACE);
                net->pending[workspace] = pf;
                return 0;
            }
            if (s != (hipStream_t)stream) {
                if (!plan->mid_ev1) RP_HIP(hipEventCreateWithFlags(&plan->mid_ev1, hipEventDisableTiming));
                RP_HIP(hipEventRecord(plan->mid_ev1, s));
                s = (hipStream_t)stream;
                RP_HIP(hipStreamWaitEvent(s, plan->mid_ev1, 0));
            }
        }
        if (op.type == OP_NOP) continue;
        if (op_index == plan->head_count && s != (hipStream_t)stream) {      // head done: the convolutions continue on `stream`
            if (!plan->head_ev) RP_HIP(hipEventCreateWithFlags(&plan->head_ev, hipEventDisableTiming));
            RP_HIP(hipEventRecord(plan->head_ev, s));
            s = (hipStream_t)stream;
            RP_HIP(hipStreamWaitEvent(s, plan->head_ev, 0));
        }
        if (op_index == plan->tail_first && two) {
            // the HBM-bound tail (heads, resize_out) continues on the caller's second stream, ordered behind everything above;
            // `stream` is free for the next forward (which must use another workspace)
            if (!plan->tail_ev) RP_HIP(hipEventCreateWithFlags(&plan->tail_ev, hipEventDisableTiming));
            RP_HIP(hipEventRecord(plan->tail_ev, s));
            s = (hipStream_t)tail_stream;
            RP_HIP(hipStreamWaitEvent(s, plan->tail_ev, 0));
        }
        if (op.type == OP_CONV) {
            mark(1);
            const ConvDesc* dd = plan->d_descs + op.first;
            // 4-wave tiles: cfg 0 = 128x128 (2x2 waves of 64x64), 1 = 256x64, 2 = 256x32, 4 = 128x64, 5 = 128x32
#define RP_LAUNCH_V(WM_, WN_, MI_, NI_, SS_, UNI_, SP_) \
            hipLaunchKernelGGL((conv_igemm_kernel<WM_, WN_, MI_, NI_, SS_, UNI_, SP_>), op.grid, dim3(256), 0, s, dd, op.ninner, op.mt_max)
#define RP_LAUNCH_T(WM_, WN_, MI_, NI_)                                                                        \
            do {                                                                                               \
                if (op.split == 1) {                                                                           \
                    if (op.uni) RP_LAUNCH_V(WM_, WN_, MI_, NI_, true, true, 1);                                \
                    else if (op.sslds) RP_LAUNCH_V(WM_, WN_, MI_, NI_, true, false, 1);                        \
                    else RP_LAUNCH_V(WM_, WN_, MI_, NI_, false, false, 1);                                     \
                } else if (op.split == 2) {                                                                    \
                    if (op.uni) RP_LAUNCH_V(WM_, WN_, MI_, NI_, true, true, 2);                                \
                    else if (op.sslds) RP_LAUNCH_V(WM_, WN_, MI_, NI_, true, false, 2);                        \
                    else RP_LAUNCH_V(WM_, WN_, MI_, NI_, false, false, 2);                                     \
                } else if (op.split == 3) {                                                                    \
                    if (op.uni) RP_LAUNCH_V(WM_, WN_, MI_, NI_, true, true, 3);                                \
                    else if (op.sslds) RP_LAUNCH_V(WM_, WN_, MI_, NI_, true, false, 3);                        \
                    else RP_LAUNCH_V(WM_, WN_, MI_, NI_, false, false, 3);                                     \
                } else if (op.split == 4) {                                                                    \
                    if (op.uni) RP_LAUNCH_V(WM_, WN_, MI_, NI_, true, true, 4);                                \
                    else if (op.sslds) RP_LAUNCH_V(WM_, WN_, MI_, NI_, true, false, 4);                        \
                    else RP_LAUNCH_V(WM_, WN_, MI_, NI_, false, false, 4);                                     \
                } else if (op.split == 5) {                                                                    \
                    if (op.uni) RP_LAUNCH_V(WM_, WN_, MI_, NI_, true, true, 5);                                \
                    else if (op.sslds) RP_LAUNCH_V(WM_, WN_, MI_, NI_, true, false, 5);                        \
                    else RP_LAUNCH_V(WM_, WN_, MI_, NI_, false, false, 5);                                     \
                } else {                                                                                       \
                    if (op.uni) RP_LAUNCH_V(WM_, WN_, MI_, NI_, true, true, 0);                                \
                    else if (op.sslds) RP_LAUNCH_V(WM_, WN_, MI_, NI_, true, false, 0);                        \
                    else RP_LAUNCH_V(WM_, WN_, MI_, NI_, false, false, 0);                                     \
                }                                                                                              \
            } while (0)
            if (op.cfg == 3) {
                if (op.sslds) hipLaunchKernelGGL((conv_igemm_kernel<4, 2, 2, 2, true>), op.grid, dim3(512), 0, s, dd, op.ninner, op.mt_max);
                else hipLaunchKernelGGL((conv_igemm_kernel<4, 2, 2, 2, false>), op.grid, dim3(512), 0, s, dd, op.ninner, op.mt_max);
            }
            else if (op.cfg == 0) RP_LAUNCH_T(2, 2, 2, 2);
            else if (op.cfg == 1) RP_LAUNCH_T(4, 1, 2, 2);
            else if (op.cfg == 2) RP_LAUNCH_T(4, 1, 2, 1);
            else if (op.cfg == 4) RP_LAUNCH_T(4, 1, 1, 2);
            else if (op.cfg == 6) RP_LAUNCH_T(2, 2, 2, 4);
            else if (op.cfg == 7) RP_LAUNCH_T(2, 2, 1, 1);
            else RP_LAUNCH_T(4, 1, 1, 1);
#undef RP_LAUNCH_T
#undef RP_LAUNCH_V
            mark(-1);
        } else if (op.type == OP_CONV_STRIP) {
            mark(1);
            // (op.split: 0 = fp32 products, 2 = f16x3, 3 = f16 -- the tile kernels stage 16-bit operands themselves; bf16x3 stays on conv_igemm_kernel)
#define RP_TILE_SPLIT(LAUNCH_)                                            \
            do {                                                          \
                if (op.split == 2) { LAUNCH_(2); }                        \
                else if (op.split == 3) { LAUNCH_(3); }                   \
                else if (op.split == 4) { LAUNCH_(4); }                   \
                else if (op.split == 5) { LAUNCH_(5); }                   \
                else { LAUNCH_(0); }                                      \
            } while (0)
            /* (variants that exist for the experiment log only: no three-piece instantiations) */ \
#define RP_TILE_SPLIT_X(LAUNCH_)                                          \
            do {                                                          \
                if (op.split == 2) { LAUNCH_(2); }                        \
                else if (op.split == 3) { LAUNCH_(3); }                   \
                else if (op.split >= 4) return RELPOSE_EINVAL;            \
                else { LAUNCH_(0); }                                      \
            } while (0)
#define RP_L_STRIP(SP_) hipLaunchKernelGGL((conv_s2_strip_kernel<4, SP_>), op.grid, dim3(256), 0, s, plan->d_descs + op.first)
            RP_TILE_SPLIT(RP_L_STRIP);
#undef RP_L_STRIP
            mark(-1);
        } else if (op.type == OP_CONV_S2) {
            mark(1);
#define RP_L_S2A(SP_) hipLaunchKernelGGL((conv_s2_tile_kernel<2, 2, 16, false, SP_>), op.grid, dim3(256), 0, s, plan->d_descs + op.first)
#define RP_L_S2B(SP_) hipLaunchKernelGGL((conv_s2_tile_kernel<1, 2, 16, false, SP_>), op.grid, dim3(256), 0, s, plan->d_descs + op.first)
#define RP_L_S2C(SP_) hipLaunchKernelGGL((conv_s2_tile_kernel<1, 4, 8, true, SP_>), op.grid, dim3(256), 0, s, plan->d_descs + op.first)
            if (op.cfg == 0) RP_TILE_SPLIT(RP_L_S2A);
            else if (op.cfg == 2) RP_TILE_SPLIT_X(RP_L_S2B);
            else RP_TILE_SPLIT(RP_L_S2C);
#undef RP_L_S2A
#undef RP_L_S2B
#undef RP_L_S2C
            mark(-1);
        } else if (op.type == OP_DECONV_TILE) {
            mark(1);
#define RP_L_DT0(SP_) hipLaunchKernelGGL((deconv_tile_kernel<2, 1, 4, 16, false, SP_>), op.grid, dim3(256), 0, s, plan->d_descs + op.first)
#define RP_L_DT1(SP_) hipLaunchKernelGGL((deconv_tile_kernel<1, 2, 4, 16, false, SP_>), op.grid, dim3(256), 0, s, plan->d_descs + op.first)
#define RP_L_DT2(SP_) hipLaunchKernelGGL((deconv_tile_kernel<1, 1, 4, 16, false, SP_>), op.grid, dim3(256), 0, s, plan->d_descs + op.first)
#define RP_L_DT4(SP_) hipLaunchKernelGGL((deconv_tile_kernel<1, 1, 4, 8, true, SP_>), op.grid, dim3(256), 0, s, plan->d_descs + op.first)
#define RP_L_DT3(SP_) hipLaunchKernelGGL((deconv_tile_kernel<1, 1, 7, 8, false, SP_>), op.grid, dim3(448), 0, s, plan->d_descs + op.first)
            if (op.cfg == 0) RP_TILE_SPLIT_X(RP_L_DT0);
            else if (op.cfg == 1) RP_TILE_SPLIT_X(RP_L_DT1);
            else if (op.cfg == 2) RP_TILE_SPLIT(RP_L_DT2);
            else if (op.cfg == 4) RP_TILE_SPLIT(RP_L_DT4);
            else RP_TILE_SPLIT_X(RP_L_DT3);
#undef RP_L_DT0
#undef RP_L_DT1
#undef RP_L_DT2
#undef RP_L_DT3
#undef RP_L_DT4
#undef RP_TILE_SPLIT
#undef RP_TILE_SPLIT_X
            mark(-1);
        } else if (op.type == OP_CONV1) {
            mark(1);
            static const bool c1_direct = RP_ENV("RELPOSE_CONV1_DIRECT") != nullptr;      // the round-1 VALU kernel (A/B switch)
            if (c1_direct)
                hipLaunchKernelGGL(conv1_direct_kernel, dim3(1024), dim3(256), 0, s, act + net->bufs["X0"].off * n, net->d_w + net->w1_off,
                                   act + net->bufs["A1"].off * n, partial, n);
            else
                hipLaunchKernelGGL(conv1_mfma_kernel, dim3(196 * n), dim3(256), 0, s, act + net->bufs["X0"].off * n, net->d_w + net->w1_off,
                                   act + net->bufs["A1"].off * n, partial, n, (plan->zero_warp ? 1 : 0) | (plan->self_cached ? 2 : 0));
            mark(-1);
        } else if (op.type == OP_STATS_FUSED) {
            const Buf& B = net->bufs[op.buf];
            mark(2);
            hipLaunchKernelGGL(bn_finalize_fused_kernel, dim3((B.C + 31) / 32, G), dim3(1024), 0, s, plan->d_descs + op.first, op.count, op.skip_blk,
                               B.C, 2 * B.H * B.H, net->d_gb + B.gb_off, net->d_gb + B.gb_off + B.C, ssp + B.ss_off * G);
            mark(-2);
        } else if (op.type == OP_SS_FILL) {
            const Buf& B = net->bufs[op.buf];
            mark(2);
            hipLaunchKernelGGL(ss_fill_kernel, dim3((B.C * G + 255) / 256), dim3(256), 0, s, B.C, G, net->d_gb + B.gb_off, net->d_gb + B.gb_off + B.C,
                               ssp + B.ss_off * G);
            mark(-2);
        } else if (op.type == OP_BCAST) {
            const Buf& B = net->bufs[op.buf];
            mark(2);
            hipLaunchKernelGGL(bcast_warped_kernel, dim3(2048), dim3(256), 0, s, act + B.off * n, n, B.H * B.H, B.C, B.C / 6, ssp + B.ss_off * G, G);
            mark(-2);
        } else if (op.type == OP_HEADS) {
            HeadsDesc hd;
            hd.d2 = act + net->bufs["D2"].off * n; hd.a1 = act + net->bufs["A1"].off * n;
            hd.ss_d2 = ssp + net->bufs["D2"].ss_off * G; hd.ss_a1 = ssp + net->bufs["A1"].ss_off * G;
            hd.w = net->d_w + net->wh_off; hd.bias = net->d_w + net->bh_off; hd.out = act + net->bufs["OUT"].off * n;
            hd.n = n; hd.S = net->S; hd.cf = net->cf; hd.use_tanh = net->use_tanh;
            hd.snap_mode = plan->snap_mode; hd.snap = (float*)(ws + o.snap) + plan->heads_snap_off;
            mark(1);
            const dim3 hg((unsigned)((size_t)n * RS * RS / 256));
            if (plan->pose_only) {
                if (net->S == 15) hipLaunchKernelGGL((heads_kernel<15, true>), hg, dim3(256), 0, s, hd);
                else hipLaunchKernelGGL((heads_kernel<21, true>), hg, dim3(256), 0, s, hd);
            } else if (net->S == 15) hipLaunchKernelGGL((heads_kernel<15, false>), hg, dim3(256), 0, s, hd);
            else hipLaunchKernelGGL((heads_kernel<21, false>), hg, dim3(256), 0, s, hd);
            mark(-1);
        } else if (op.type == OP_REDUCE) {
            mark(4);
            if (op.cfg == 1) {
                const ConvDesc& d0 = plan->descs[op.first];
                const int q4 = d0.cout_pad / 4, nrl = 256 / q4;
                hipLaunchKernelGGL(splitk_reduce_stats_kernel, op.grid, dim3(256), (size_t)std::max(nrl, 1) * q4 * 8 * sizeof(double), s,
                                   plan->d_descs + op.first, op.ninner, net->bufs[op.buf].C, partial);
            } else hipLaunchKernelGGL(splitk_reduce_kernel, op.grid, dim3(256), 0, s, plan->d_descs + op.first);
            mark(-4);
        } else {
            const Buf& B = net->bufs[op.buf];
            const int rows = 2 * B.H * B.H;
            if (op.cfg == 1) {                     // A1: finalise the per-pass records of conv1_direct_kernel
                mark(2);
                hipLaunchKernelGGL(bn_finalize_parts_kernel, dim3((B.C + 31) / 32, G), dim3(1024), 0, s, partial, C1_PASSES_PER_GROUP, B.C, rows,
                                   net->d_gb + B.gb_off, net->d_gb + B.gb_off + B.C, ssp + B.ss_off * G, op.skip_blk);
                mark(-2);
                continue;
            }
            if (op.cfg == 3) {                     // the split-K reduce kernel left op.count records per group: finalize only
                mark(2);
                hipLaunchKernelGGL(bn_finalize_parts_kernel, dim3((B.C + 31) / 32, G), dim3(1024), 0, s, partial, op.count, B.C, rows,
                                   net->d_gb + B.gb_off, net->d_gb + B.gb_off + B.C, ssp + B.ss_off * G, 0);
                mark(-2);
                continue;
            }
            int nch = (2048 + G - 1) / G;
            if (nch > (rows + 63) / 64) nch = (rows + 63) / 64;
            if (nch > 64) nch = 64;
            if (nch < 1) nch = 1;
            const int chunk_rows = (rows + nch - 1) / nch;
            nch = (rows + chunk_rows - 1) / chunk_rows;
            const int q4 = B.C / 4, nrl = 256 / q4;
            mark(2);
            hipLaunchKernelGGL(bn_partial_kernel, dim3(nch, G), dim3(256), (size_t)nrl * q4 * 8 * sizeof(double), s, act + B.off * n, rows,
                               B.C, chunk_rows, partial);
            hipLaunchKernelGGL(bn_finalize_kernel, dim3((B.C + 255) / 256, G), dim3(256), 0, s, partial, nch, B.C, rows,
                               net->d_gb + B.gb_off, net->d_gb + B.gb_off + B.C, ssp + B.ss_off * G);
            mark(-2);
        }
    }
    mark(3);
    hipLaunchKernelGGL(resize_out_kernel, dim3(4096), dim3(256), net->cf <= RO_MAXC ? (size_t)4 * 2 * RO_MAXW * net->cf * sizeof(float) : 0, s,
                       act + net->bufs["OUT"].off * n, out, n, net->cf, H, W);
    mark(-3);
    RP_CHECK_LAUNCH();
    {   // everything is enqueued: the workspace now holds (stream-ordered) the self-view streams of `self_tag`
        net->self_state[workspace] = commit;
    }
    return 0;
}

// Multiply-accumulates one forward of the given plan family EXECUTES (descriptor-based: rows x K x Cout of every launched member, with the
// slices / chunks / members the level-0 plan and the self-stream cache leave out removed) -- bench.py's plan-aware in-loop roofline divides
// it by the same count of the full plan.  Host-only (a dry-run plan build: no device memory is touched).
int relpose_scnet_plan_macs(RelposeSCNet* net, int32_t n, int32_t flags, int32_t self_cached, double* macs_out) {
    if (!net || !net->finalized || !macs_out || n <= 0 || (n & 1)) return RELPOSE_EINVAL;
    if (net->variant()) { flags = 0; self_cached = 0; }          // (constructor variants run the plain plan whatever the flags)
    const bool zero_warp = (flags & RELPOSE_FWD_ZERO_WARP) && n > 2 && !self_cached;
    const bool pose_only = (flags & RELPOSE_FWD_POSE_OUTPUTS) != 0;
    Plan dry;
    Builder B; B.net = net; B.n = n; B.G = n / 2; B.act = nullptr; B.ss = nullptr; B.splitk = nullptr; B.statp = nullptr; B.plan = &dry;
    B.zero_warp = zero_warp; B.pose_only = pose_only; B.self_cached = self_cached != 0;
    B.snapbuf = (float*)(uintptr_t)4096;       // (never dereferenced: makes the builder record ConvDesc::snap_mode in the dry run)
    B.snap_mode = self_cached ? 2 : 1; dry.snap_mode = B.snap_mode;
    build_plan(net, n, B);
    if (B.rc) return B.rc;
    double macs = 0;
    for (const Op& op : dry.ops) {
        if (op.type == OP_CONV || op.type == OP_CONV_S2 || op.type == OP_CONV_STRIP || op.type == OP_DECONV_TILE) {
            for (int i = op.first; i < op.first + op.count; ++i) {
                const ConvDesc& d = dry.descs[i];
                double rowsK;
                if (op.type == OP_CONV_STRIP && d.ksplit > 1 && (d.shared_slices || d.skip_slices)) {
                    rowsK = 0;
                    for (int ks = 0; ks < d.ksplit; ++ks) {
                        if ((d.skip_slices >> ks) & 1) continue;
                        const double rows = ((d.shared_slices >> ks) & 1) ? 2.0 * d.Hp * d.Wp : (double)d.M;
                        rowsK += rows * d.K / d.ksplit;
                    }
                } else if (op.type == OP_DECONV_TILE && d.nsrc == 2 && plan_snap_mode(dry, i) == 2) {
                    rowsK = (double)d.M * d.ntaps * d.src[0].C;          // the skip source's chunks come from the snapshot
                } else rowsK = (double)d.M * d.K;
                macs += rowsK * d.Cout;
            }
        } else if (op.type == OP_CONV1) {
            // six 3x3 blocks (Cin 4 / 4 / 2 -> 32), self + warped stream each; level 0 runs the warped blocks on the first image pair only,
            // a self-cached forward the warped blocks only
            const double blk = 224.0 * 224 * 32 * (36 + 36 + 18);
            const double self_imgs = self_cached ? 0 : n, warp_imgs = zero_warp ? 2 : n;
            macs += blk * (self_imgs + warp_imgs);
        } else if (op.type == OP_HEADS) {
            // 1x1 heads: rgb / n / d read cat(D2 32, A1 skip 32), s / f read 64 channels of D2; a self-cached forward takes the skip halves' sums
            // from the snapshot; the pose-outputs plan skips the rgb and semantic heads
            const double px = (double)n * 224 * 224;
            const int hc[5] = {3, 3, 1, net->S, 32};
            for (int m = 0; m < 5; ++m) {
                if (pose_only && (m == 0 || m == 3)) continue;
                const int cin = (m < 3 && self_cached) ? 32 : 64;
                macs += px * cin * hc[m];
            }
        }
    }
    *macs_out = macs;
    return 0;
}

int64_t relpose_scnet_read_tap(RelposeSCNet* net, const char* name, float* out, void* workspace, void* stream) {
    if (!net || !net->finalized || !name || net->last_n <= 0) return -1;
    auto it = net->bufs.find(name);
    if (it == net->bufs.end()) return -1;
    const Buf& B = it->second;
    const int64_t count = (int64_t)net->last_n * B.H * B.H * B.C;
    if (out) {
        if (!workspace) return -1;
        const float* src = (const float*)workspace + B.off * net->last_n;
        if (hipMemcpyAsync(out, src, count * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return -1;
    }
    return count;
}

int relpose_scnet_profile(RelposeSCNet* net, const float* x, float* out, int32_t n, int32_t H, int32_t W, void* workspace,
                          size_t workspace_bytes, int32_t iters, double* ms_gemm, double* ms_other, int64_t* n_gemm, void* stream) {
    if (!net || iters <= 0) return RELPOSE_EINVAL;
    double tg = 0, to = 0; int64_t ng = 0;
    for (int it = 0; it < iters; ++it) {
        net->profiling = true; net->ev.clear(); net->ev_kind.clear();
        int rc = relpose_scnet_forward(net, x, out, n, H, W, workspace, workspace_bytes, stream);
        net->profiling = false;
        if (rc) return rc;
        RP_HIP(hipStreamSynchronize((hipStream_t)stream));
        for (size_t i = 0; i + 1 < net->ev.size(); i += 2) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, net->ev[i], net->ev[i + 1]) != hipSuccess) ms = 0.f;
            if (net->ev_kind[i] == 1) { tg += ms; ++ng; } else to += ms;
        }
        for (auto e : net->ev) (void)hipEventDestroy(e);
        net->ev.clear(); net->ev_kind.clear();
    }
    if (ms_gemm) *ms_gemm = tg / iters;
    if (ms_other) *ms_other = to / iters;
    if (n_gemm) *n_gemm = ng / iters;
    return 0;
}

}  // extern "C"
