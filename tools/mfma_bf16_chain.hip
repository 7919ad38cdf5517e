// Micro-benchmark (round 6): v_mfma_f32_32x32x16_bf16 issue rate per SIMD as a function of how many INDEPENDENT accumulators a wave rotates
// through (1 = every MFMA depends on the previous one: the shape of the MI = NI = 1 tile kernels in the split modes), next to the fp32 MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_chain.hip -o /tmp/mfma_bf16_chain && /tmp/mfma_bf16_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, bool F32>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0) {
    floatx16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + threadIdx.x * 1e-3f + e); b[e] = (__bf16)(a0 * 0.5f + threadIdx.x * 2e-3f - e); }
    const float fa = a0 + threadIdx.x * 1e-3f, fb = a0 * 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if constexpr (F32) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool F32>
void run(float* d, int wpb) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * wpb, iters = F32 ? 4000 : 8000;
    k<NACC, F32><<<blocks, 256>>>(d, 100, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<NACC, F32><<<blocks, 256>>>(d, iters, 1.f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas_per_simd = (double)wpb * iters * 32;              // one wave of every block per SIMD
    const double flop = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * (F32 ? 2 : 16);
    printf("%s accumulators=%d waves/SIMD=%d: %8.1f TFLOP/s, %.1f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz)\n", F32 ? "f32 32x32x2  " : "bf16 32x32x16",
           NACC, wpb, flop / ms / 1e9, ms * 1e6 / mfmas_per_simd, ms * 1e6 / mfmas_per_simd * 2.4);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 2048 * 4);
    for (int wpb : {1, 2, 3}) { run<1, false>(d, wpb); run<2, false>(d, wpb); run<4, false>(d, wpb); }
    for (int wpb : {1, 2}) { run<1, true>(d, wpb); run<4, true>(d, wpb); }
    return 0;
}
