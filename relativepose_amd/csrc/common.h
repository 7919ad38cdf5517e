// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/relpose.h"

#define RP_WAVE 64

// A/B switches of the experiment log exist only in builds with -DRP_EXPERIMENTS (tools/build_variant.py);
// the product library reads no environment variable.
#ifdef RP_EXPERIMENTS
#include <stdlib.h>
#define RP_ENV(name) getenv(name)
#else
#define RP_ENV(name) ((const char*)nullptr)
#endif

#define RP_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return -(1000 + (int)e__);   \
    } while (0)

#define RP_HIP(x)                                           \
    do {                                                    \
        hipError_t e__ = (x);                               \
        if (e__ != hipSuccess) return -(1000 + (int)e__);   \
    } while (0)

// Global-address-space accessors.  Pointers that reach a kernel through a descriptor struct in memory are
// "generic" to the compiler, which then emits FLAT loads/stores; FLAT ops also count against lgkmcnt, so every
// wait for an LDS read would drain the outstanding global loads as well.  These casts give plain
// global_load / global_store (vmcnt only).
typedef float rp_v4f __attribute__((ext_vector_type(4)));
typedef float rp_v2f __attribute__((ext_vector_type(2)));
#define RP_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ float4 rp_ldg4(const float* p) { const rp_v4f v = *(RP_GLOBAL const rp_v4f*)p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float2 rp_ldg2(const float* p) { const rp_v2f v = *(RP_GLOBAL const rp_v2f*)p; return make_float2(v.x, v.y); }
__device__ __forceinline__ float rp_ldg(const float* p) { return *(RP_GLOBAL const float*)p; }
__device__ __forceinline__ double rp_ldg(const double* p) { return *(RP_GLOBAL const double*)p; }
__device__ __forceinline__ void rp_stg4(float* p, float4 v) { const rp_v4f w = {v.x, v.y, v.z, v.w}; *(RP_GLOBAL rp_v4f*)p = w; }
__device__ __forceinline__ void rp_stg2(float* p, float2 v) { const rp_v2f w = {v.x, v.y}; *(RP_GLOBAL rp_v2f*)p = w; }
__device__ __forceinline__ void rp_stg(float* p, float v) { *(RP_GLOBAL float*)p = v; }
__device__ __forceinline__ void rp_stg(double* p, double v) { *(RP_GLOBAL double*)p = v; }

__device__ __forceinline__ int rp_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ double rp_shfl_xor_d(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rp_shfl_d(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}

// DPP lane exchanges (no LDS): bound_ctrl on, lanes without a source read 0
template <int CTRL>
__device__ __forceinline__ int rp_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ double rp_dpp_d(double v) {
    return __hiloint2double(rp_dpp<CTRL>(__double2hiint(v)), rp_dpp<CTRL>(__double2loint(v)));
}
#define RP_ROW_SHR(n) (0x110 + (n))
#define RP_ROW_SHL(n) (0x100 + (n))
__device__ __forceinline__ double rp_readlane_d(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// Fixed-order sum over the 64 lanes of a wave (deterministic), result in every lane, without LDS traffic: butterfly inside
// every row of 16 lanes with DPP (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then ((r0 + r1) + (r2 + r3)) of the
// four row sums through SGPRs.  (The ds_bpermute butterfly this replaces cost ~1.2 k cycles of LDS latency per double.)
__device__ __forceinline__ double rp_wave_sum(double v) {
    v += rp_dpp_d<0xB1>(v);
    v += rp_dpp_d<0x4E>(v);
    v += rp_dpp_d<0x141>(v);
    v += rp_dpp_d<0x140>(v);
    return (rp_readlane_d(v, 0) + rp_readlane_d(v, 16)) + (rp_readlane_d(v, 32) + rp_readlane_d(v, 48));
}
__device__ __forceinline__ int rp_wave_sum_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Deterministic block-wide sum of `nv` doubles per thread; result broadcast to all
// threads.  `red` is LDS scratch of at least nv*(blockDim/64) doubles.
template <int NV>
__device__ __forceinline__ void rp_block_sum(double (&v)[NV], double* red) {
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = rp_wave_sum(v[i]);
    __syncthreads();
    if (rp_lane() == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[i * nw + wave] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += red[i * nw + w];
        v[i] = s;
    }
}

static inline size_t rp_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
