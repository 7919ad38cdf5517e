// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate on this GPU (calibrates the fp32-MFMA roofline
// under the chip's real clock/power behaviour).  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpb : {1, 2, 3}) {
        const int blocks = 256 * wpb, iters = 20000;
        k<<<blocks, 256>>>(d, 100, 1.f, 2.f);
        hipDeviceSynchronize();
        hipEventRecord(e0); k<<<blocks, 256>>>(d, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 /*waves*/ * iters * 32 /*mfma*/ * 2.0 * 32 * 32 * 2;
        printf("blocks/CU=%d: %.1f TFLOP/s (%.2f ms)  -> implied clock %.2f GHz\n", wpb, flops / ms / 1e9, ms,
               flops / ms / 1e9 / 157.3 * 2.4);
    }
    return 0;
}
