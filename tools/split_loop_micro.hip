// Micro-benchmark (round 6): what does the INNER LOOP of the split-mode tile kernels cost on its own?
// A workgroup of 4 waves keeps an A tile [128 rows x 96 bf16 (hi|mid|lo)] and a B tile [NI*32 rows x 96] in LDS (208-byte rows, like csrc/scnet.hip) and runs
// `steps` x { fragment reads of two 16-channel steps + the six bf16x6 MFMA terms per (i, j) } on MI x NI accumulators -- no global traffic, no staging.
// Variants: PIPE (next step's reads before this step's MFMAs), BAR (a __syncthreads pair per step, as the kernels have), workgroups per CU by dynamic LDS padding.
//   hipcc --offload-arch=gfx950 -O3 tools/split_loop_micro.hip -o /tmp/split_loop_micro && /tmp/split_loop_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LD = 52;   // floats per LDS row
template <int MI, int NI, bool PIPE, bool BAR, int OCC>
__global__ __launch_bounds__(256, OCC) void k(float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* At = lds;                       // 128 * MI rows
    float* Bt = lds + 128 * MI * LD;       // NI * 32 rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    for (int i = tid; i < (128 * MI + NI * 32) * LD; i += 256) lds[i] = __int_as_float(0x3f803f80 + (i * 2654435761u >> 20));   // two plausible bf16 per float
    __syncthreads();
    floatx16 acc[MI][NI];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int arow[MI], brow[NI];
    for (int i = 0; i < MI; ++i) arow[i] = ((wave * MI + i) * 32 + l31) * LD + h * 4;
    for (int j = 0; j < NI; ++j) brow[j] = (j * 32 + l31) * LD + h * 4;
    bf16x8 a[2][3][MI], b[2][3][NI];
    auto load = [&](int buf, int st) {
        for (int pc = 0; pc < 3; ++pc) {
            for (int i = 0; i < MI; ++i) a[buf][pc][i] = *reinterpret_cast<const bf16x8*>(&At[arow[i] + pc * 16 + st * 8]);
            for (int j = 0; j < NI; ++j) b[buf][pc][j] = *reinterpret_cast<const bf16x8*>(&Bt[brow[j] + pc * 16 + st * 8]);
        }
    };
    constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};
    auto mma = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[buf][ta[t]][i], b[buf][tb[t]][j], acc[i][j], 0, 0, 0);
    };
    if (PIPE) load(0, 0);
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (PIPE) {
                load((st + 1) & 1, (st + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(st & 1);
                __builtin_amdgcn_sched_barrier(0);
            } else { load(0, st); mma(0); }
        }
        if (BAR) { __syncthreads(); __syncthreads(); }
    }
    float sum = 0;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = sum;
}
template <int MI, int NI, bool PIPE, bool BAR, int OCC>
void run(float* d, const char* name) {
    const size_t need = (size_t)(128 * MI + NI * 32) * LD * 4;
    const size_t lds = OCC == 1 ? 100 * 1024 : (OCC == 2 ? 70 * 1024 : 50 * 1024);      // pad so that exactly OCC workgroups fit a CU
    if (need > lds) { printf("%s: tile does not fit\n", name); return; }
    (void)hipFuncSetAttribute((const void*)k<MI, NI, PIPE, BAR, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * OCC * 4, steps = 400;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MI, NI, PIPE, BAR, OCC><<<blocks, 256, lds>>>(d, 20);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MI, NI, PIPE, BAR, OCC><<<blocks, 256, lds>>>(d, steps); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 4 * steps * 2 * 6 * MI * NI;
    const double tf = mfma * 2.0 * 32 * 32 * 16 / ms / 1e9;
    printf("%-58s %7.1f bf16 TFLOP/s = %4.1f %% of 2500 (%.2f ms)\n", name, tf, tf / 25.0, ms);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 16 * 256 * 4);
    run<1, 1, false, false, 2>(d, "MI1 NI1, 2 WG/CU, reads then MFMAs");
    run<1, 1, true, false, 2>(d, "MI1 NI1, 2 WG/CU, pipelined reads");
    run<1, 1, false, true, 2>(d, "MI1 NI1, 2 WG/CU, reads then MFMAs, barrier pair per step");
    run<1, 1, true, true, 2>(d, "MI1 NI1, 2 WG/CU, pipelined, barrier pair per step");
    run<1, 1, false, false, 3>(d, "MI1 NI1, 3 WG/CU, reads then MFMAs");
    run<1, 1, false, false, 1>(d, "MI1 NI1, 1 WG/CU, reads then MFMAs");
    run<1, 1, true, false, 1>(d, "MI1 NI1, 1 WG/CU, pipelined reads");
    run<1, 4, false, false, 2>(d, "MI1 NI4, 2 WG/CU, reads then MFMAs");
    run<1, 4, true, false, 2>(d, "MI1 NI4, 2 WG/CU, pipelined reads");
    run<1, 4, false, true, 2>(d, "MI1 NI4, 2 WG/CU, reads then MFMAs, barrier pair per step");
    run<2, 2, false, false, 2>(d, "MI2 NI2, 2 WG/CU, reads then MFMAs");
    run<2, 2, true, false, 2>(d, "MI2 NI2, 2 WG/CU, pipelined reads");
    run<2, 2, true, false, 1>(d, "MI2 NI2, 1 WG/CU, pipelined reads");
    run<2, 1, true, false, 1>(d, "MI2 NI1, 1 WG/CU, pipelined reads");
    return 0;
}
