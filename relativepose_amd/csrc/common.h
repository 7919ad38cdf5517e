// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/relpose.h"

#define RP_WAVE 64

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

// Fixed-order butterfly sum over the 64 lanes of a wave (deterministic).
__device__ __forceinline__ double rp_wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += rp_shfl_xor_d(v, m);
    return v;
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
