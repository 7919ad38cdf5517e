// Micro-benchmark (round 6, last experiment): would PRODUCER / CONSUMER waves fill the bf16 pipe where the present tile kernels (every wave stages, then
// multiplies, two barriers per chunk) reach 50-59 %?  Synthetic but sized like deconv_tile_kernel<1,1,4,16,.,bf16x6>:
//   per K chunk a workgroup stages NLD float4 of activations per staging thread (global load -> BatchNorm fma + LeakyReLU + three-piece bf16 split, ~10 vector
//   instructions per element -> ds_write) and NB 16-byte weight loads (copied to LDS as they are); a computing wave then runs NSTEP x {3 + 3 fragment reads,
//   6 MFMA terms} x PH accumulators from that chunk.
//   MODE 0: the present structure -- 4 waves, each stages its share, barrier, each multiplies, barrier (single-buffered LDS).
//   MODE 1: 8 waves, waves 0-3 only multiply from buffer b while waves 4-7 only stage buffer b^1; ONE barrier per chunk (double-buffered LDS).
//   hipcc --offload-arch=gfx950 -O3 tools/split_specialized_micro.hip -o /tmp/ssm && /tmp/ssm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int LD = 52;                     // floats per LDS row (208 bytes: 32 x {hi, mid, lo} bf16 + pad)
constexpr int AROWS = 160, BROWS = 128;    // staged pixels (8 x 16 patch + halo), weight rows (4 phases x 32 output channels)
constexpr int TILE = (AROWS + BROWS) * LD; // floats per LDS buffer (59.9 KB)
constexpr int LD16 = 28;                   // 16-channel chunks: 96 bytes of pieces + 16 pad

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 r = __builtin_convertvector((v2f){a, b}, bf2);
    return __builtin_bit_cast(unsigned, r);
}
// stage one thread's share of a chunk into `buf`: NLD activation float4 (with the transform) + NB weight float4 (plain copy)
// STG 1: PRE-SPLIT activations (what an own pass or the producer layer's epilogue would have written: 6 bytes per element): loaded and stored as they are
template <int NLD, int NB>
__device__ __forceinline__ void stage_copy(float* buf, const float4* __restrict__ ga, const float4* __restrict__ gb, size_t off, int st_tid) {
    constexpr int NC = (NLD * 3 + 1) / 2;                    // 16-byte loads for the same elements at 6 bytes each
    float4 x[NC > 0 ? NC : 1], w[NB > 0 ? NB : 1];
#pragma unroll
    for (int i = 0; i < NC; ++i) x[i] = ga[off + (size_t)i * 256 + st_tid];
#pragma unroll
    for (int i = 0; i < NB; ++i) w[i] = gb[(off & 0xffff) + (size_t)i * 256 + st_tid];
#pragma unroll
    for (int i = 0; i < NC; ++i) *reinterpret_cast<float4*>(buf + ((i * 256 + st_tid) * 4) % (AROWS * LD - 4)) = x[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(buf + AROWS * LD + ((i * 256 + st_tid) * 4) % (BROWS * LD - 4)) = w[i];
}
template <int NLD, int NB>
__device__ __forceinline__ void stage(float* buf, const float4* __restrict__ ga, const float4* __restrict__ gb, size_t off, int st_tid, float sc, float sh) {
    float4 x[NLD > 0 ? NLD : 1], w[NB > 0 ? NB : 1];
#pragma unroll
    for (int i = 0; i < NLD; ++i) x[i] = ga[off + (size_t)i * 256 + st_tid];
#pragma unroll
    for (int i = 0; i < NB; ++i) w[i] = gb[(off & 0xffff) + (size_t)i * 256 + st_tid];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        float v[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
        unsigned hi[2], mid[2], lo[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float a = fmaf(v[2 * p], sc, sh), b = fmaf(v[2 * p + 1], sc, sh);
            a = fmaxf(a, 0.1f * a); b = fmaxf(b, 0.1f * b);
            hi[p] = pk_bf16(a, b);
            const float ra = a - __uint_as_float(hi[p] << 16), rb = b - __uint_as_float(hi[p] & 0xffff0000u);
            mid[p] = pk_bf16(ra, rb);
            lo[p] = pk_bf16(ra - __uint_as_float(mid[p] << 16), rb - __uint_as_float(mid[p] & 0xffff0000u));
        }
        const int e = (i * 256 + st_tid) * 4;                  // element index in the chunk: row = e / 32, channel = e % 32
        float* row = buf + (e >> 5) % AROWS * LD + ((e & 31) >> 1);
        *reinterpret_cast<uint2*>(row) = make_uint2(hi[0], hi[1]);
        *reinterpret_cast<uint2*>(row + 16) = make_uint2(mid[0], mid[1]);
        *reinterpret_cast<uint2*>(row + 32) = make_uint2(lo[0], lo[1]);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(buf + AROWS * LD + ((i * 256 + st_tid) * 4) % (BROWS * LD - 4)) = w[i];
}
template <int PH, int NSTEP>
__device__ __forceinline__ void multiply(const float* buf, floatx16 (&acc)[PH], int wave, int lane) {
    const int h = lane >> 5, l31 = lane & 31;
    constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
#pragma unroll
        for (int p = 0; p < PH; ++p) {
            bf16x8 a[3], b[3];
            const float* ar = buf + ((wave * 32 + l31 + s + p) % AROWS) * LD + h * 4 + (s & 1) * 8;
            const float* br = buf + AROWS * LD + (p * 32 + l31) * LD + h * 4 + (s & 1) * 8;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) { a[pc] = *reinterpret_cast<const bf16x8*>(ar + pc * 16); b[pc] = *reinterpret_cast<const bf16x8*>(br + pc * 16); }
#pragma unroll
            for (int t = 0; t < 6; ++t) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta[t]], b[tb[t]], acc[p], 0, 0, 0);
        }
    }
}
// MODE 2: the present structure with pre-split activations (stage_copy); MODE 3: the same, software-pipelined over chunks in double-buffered LDS (the next chunk's
// loads are issued before this chunk's MFMAs, its LDS writes after them: ONE barrier per chunk)
template <int MODE, int PH, int NSTEP, int NLD, int NB, int OCC>
__global__ __launch_bounds__(256, OCC) void k2(float* out, const float4* ga, const float4* gb, int chunks, size_t gmask) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (MODE == 3 ? 2 : 1) * TILE; i += blockDim.x) lds[i] = __int_as_float(0x3f803f80 + (i * 2654435761u >> 20));
    __syncthreads();
    floatx16 acc[PH];
    for (int p = 0; p < PH; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    size_t off = ((size_t)blockIdx.x * 7919 * 4096) & gmask;
    if (MODE == 2) {
        for (int c = 0; c < chunks; ++c) {
            stage_copy<NLD, NB>(lds, ga, gb, off, tid);
            off = (off + NLD * 256) & gmask;
            __syncthreads();
            multiply<PH, NSTEP>(lds, acc, wave, lane);
            __syncthreads();
        }
    } else {
        constexpr int NC = (NLD * 3 + 1) / 2;
        float4 x[NC > 0 ? NC : 1], w[NB > 0 ? NB : 1];
        stage_copy<NLD, NB>(lds, ga, gb, off, tid);
        off = (off + NLD * 256) & gmask;
        __syncthreads();
        for (int c = 0; c < chunks; ++c) {
            float* cur = lds + (c & 1) * TILE;
            float* nxt = lds + ((c + 1) & 1) * TILE;
#pragma unroll
            for (int i = 0; i < NC; ++i) x[i] = ga[off + (size_t)i * 256 + tid];
#pragma unroll
            for (int i = 0; i < NB; ++i) w[i] = gb[(off & 0xffff) + (size_t)i * 256 + tid];
            multiply<PH, NSTEP>(cur, acc, wave, lane);
#pragma unroll
            for (int i = 0; i < NC; ++i) *reinterpret_cast<float4*>(nxt + ((i * 256 + tid) * 4) % (AROWS * LD - 4)) = x[i];
#pragma unroll
            for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(nxt + AROWS * LD + ((i * 256 + tid) * 4) % (BROWS * LD - 4)) = w[i];
            off = (off + NLD * 256) & gmask;
            __syncthreads();
        }
    }
    float sum = 0;
    for (int p = 0; p < PH; ++p) for (int r = 0; r < 16; ++r) sum += acc[p][r];
    out[blockIdx.x * 256 + tid] = sum;
}
template <int MODE, int PH, int NSTEP, int NLD, int NB, int OCC>
void run2(float* d, const float4* ga, const float4* gb, size_t gmask, const char* name) {
    const size_t lds = (size_t)(MODE == 3 ? 2 : 1) * TILE * 4;
    (void)hipFuncSetAttribute((const void*)k2<MODE, PH, NSTEP, NLD, NB, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * OCC * 4, chunks = 96;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k2<MODE, PH, NSTEP, NLD, NB, OCC><<<blocks, 256, lds>>>(d, ga, gb, 8, gmask);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k2<MODE, PH, NSTEP, NLD, NB, OCC><<<blocks, 256, lds>>>(d, ga, gb, chunks, gmask); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    const double mfma = (double)blocks * 4 * chunks * NSTEP * PH * 6;
    const double tf = mfma * 2.0 * 32 * 32 * 16 / ms / 1e9;
    printf("%-86s %7.1f bf16 TFLOP/s = %4.1f %% of 2500 (%.2f ms)%s\n", name, tf, tf / 25.0, ms, err ? " ERROR" : "");
}
template <int MODE, int PH, int NSTEP, int NLD, int NB, int OCC>
__global__ __launch_bounds__(MODE ? 512 : 256, OCC) void k(float* out, const float4* ga, const float4* gb, int chunks, size_t gmask) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (MODE ? 2 : 1) * TILE; i += blockDim.x) lds[i] = __int_as_float(0x3f803f80 + (i * 2654435761u >> 20));
    __syncthreads();
    floatx16 acc[PH];
    for (int p = 0; p < PH; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    size_t off = ((size_t)blockIdx.x * 7919 * 4096) & gmask;
    if (MODE == 0) {
        for (int c = 0; c < chunks; ++c) {
            stage<NLD, NB>(lds, ga, gb, off, tid, 1.01f, 0.01f);
            off = (off + NLD * 256) & gmask;
            __syncthreads();
            multiply<PH, NSTEP>(lds, acc, wave, lane);
            __syncthreads();
        }
    } else {
        const bool producer = wave >= 4;
        if (producer) stage<NLD, NB>(lds, ga, gb, off, tid - 256, 1.01f, 0.01f);
        off = (off + NLD * 256) & gmask;
        __syncthreads();
        for (int c = 0; c < chunks; ++c) {
            float* cur = lds + (c & 1) * TILE;
            float* nxt = lds + ((c + 1) & 1) * TILE;
            if (producer) { if (c + 1 < chunks) stage<NLD, NB>(nxt, ga, gb, off, tid - 256, 1.01f, 0.01f); }
            else multiply<PH, NSTEP>(cur, acc, wave, lane);
            off = (off + NLD * 256) & gmask;
            __syncthreads();
        }
    }
    float sum = 0;
    for (int p = 0; p < PH; ++p) for (int r = 0; r < 16; ++r) sum += acc[p][r];
    if (MODE == 0 || wave < 4) out[blockIdx.x * 256 + (tid & 255)] = sum;
}
template <int MODE, int PH, int NSTEP, int NLD, int NB, int OCC>
void run(float* d, const float4* ga, const float4* gb, size_t gmask, const char* name) {
    const size_t lds = (size_t)(MODE ? 2 : 1) * TILE * 4;
    (void)hipFuncSetAttribute((const void*)k<MODE, PH, NSTEP, NLD, NB, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * OCC * 4, chunks = 96;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, PH, NSTEP, NLD, NB, OCC><<<blocks, MODE ? 512 : 256, lds>>>(d, ga, gb, 8, gmask);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MODE, PH, NSTEP, NLD, NB, OCC><<<blocks, MODE ? 512 : 256, lds>>>(d, ga, gb, chunks, gmask); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    const double mfma = (double)blocks * 4 * chunks * NSTEP * PH * 6;
    const double tf = mfma * 2.0 * 32 * 32 * 16 / ms / 1e9;
    printf("%-86s %7.1f bf16 TFLOP/s = %4.1f %% of 2500 (%.2f ms)%s\n", name, tf, tf / 25.0, ms, err ? " ERROR" : "");
}

// ---- generalised producer / consumer: NCW computing waves (32 rows each), NPW staging waves, 16-channel chunks (one k-step of 6 terms per phase per chunk),
//      double-buffered LDS of (32 * NCW + 32 + BROWS) rows x 112 bytes per buffer
template <int NLD, int NB, int NPT, int AR>
__device__ __forceinline__ void stage16(float* buf, const float4* __restrict__ ga, const float4* __restrict__ gb, size_t off, int st_tid, float sc, float sh) {
    float4 x[NLD > 0 ? NLD : 1], w[NB > 0 ? NB : 1];
#pragma unroll
    for (int i = 0; i < NLD; ++i) x[i] = ga[off + (size_t)i * NPT + st_tid];
#pragma unroll
    for (int i = 0; i < NB; ++i) w[i] = gb[(off & 0xffff) + (size_t)i * NPT + st_tid];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        float v[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
        unsigned hi[2], mid[2], lo[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float a = fmaf(v[2 * p], sc, sh), b = fmaf(v[2 * p + 1], sc, sh);
            a = fmaxf(a, 0.1f * a); b = fmaxf(b, 0.1f * b);
            hi[p] = pk_bf16(a, b);
            const float ra = a - __uint_as_float(hi[p] << 16), rb = b - __uint_as_float(hi[p] & 0xffff0000u);
            mid[p] = pk_bf16(ra, rb);
            lo[p] = pk_bf16(ra - __uint_as_float(mid[p] << 16), rb - __uint_as_float(mid[p] & 0xffff0000u));
        }
        const int e = (i * NPT + st_tid) * 4;                  // row = e / 16, channel = e % 16
        float* row = buf + (e >> 4) % AR * LD16 + ((e & 15) >> 1);
        *reinterpret_cast<uint2*>(row) = make_uint2(hi[0], hi[1]);
        *reinterpret_cast<uint2*>(row + 8) = make_uint2(mid[0], mid[1]);
        *reinterpret_cast<uint2*>(row + 16) = make_uint2(lo[0], lo[1]);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(buf + AR * LD16 + ((i * NPT + st_tid) * 4) % (BROWS * LD16 - 4)) = w[i];
}
template <int PH, int AR>
__device__ __forceinline__ void multiply16(const float* buf, floatx16 (&acc)[PH], int wave, int lane) {
    const int h = lane >> 5, l31 = lane & 31;
    constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};
    bf16x8 a[PH][3], b[PH][3];
#pragma unroll
    for (int p = 0; p < PH; ++p) {
        const float* ar = buf + ((wave * 32 + l31 + p) % AR) * LD16 + h * 4;
        const float* br = buf + AR * LD16 + (p * 32 + l31) * LD16 + h * 4;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) { a[p][pc] = *reinterpret_cast<const bf16x8*>(ar + pc * 8); b[p][pc] = *reinterpret_cast<const bf16x8*>(br + pc * 8); }
    }
#pragma unroll
    for (int p = 0; p < PH; ++p)
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[p][ta[t]], b[p][tb[t]], acc[p], 0, 0, 0);
}
template <int NCW, int NPW, int PH, int NLD, int NB, int OCC>
__global__ __launch_bounds__((NCW + NPW) * 64, OCC) void k16(float* out, const float4* ga, const float4* gb, int chunks, size_t gmask) {
    constexpr int AR = 32 * NCW + 32, T16 = (AR + BROWS) * LD16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * T16; i += blockDim.x) lds[i] = __int_as_float(0x3f803f80 + (i * 2654435761u >> 20));
    __syncthreads();
    floatx16 acc[PH];
    for (int p = 0; p < PH; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    size_t off = ((size_t)blockIdx.x * 7919 * 4096) & gmask;
    const bool producer = wave >= NCW;
    if (producer) stage16<NLD, NB, NPW * 64, AR>(lds, ga, gb, off, tid - NCW * 64, 1.01f, 0.01f);
    off = (off + NLD * NPW * 64) & gmask;
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        float* cur = lds + (c & 1) * T16;
        float* nxt = lds + ((c + 1) & 1) * T16;
        if (producer) { if (c + 1 < chunks) stage16<NLD, NB, NPW * 64, AR>(nxt, ga, gb, off, tid - NCW * 64, 1.01f, 0.01f); }
        else multiply16<PH, AR>(cur, acc, wave, lane);
        off = (off + NLD * NPW * 64) & gmask;
        __syncthreads();
    }
    float sum = 0;
    for (int p = 0; p < PH; ++p) for (int r = 0; r < 16; ++r) sum += acc[p][r];
    if (!producer) out[(blockIdx.x * NCW * 64 + tid) & 0xfffff] = sum;
}
// all waves stage AND multiply (the present division of labour), but 16-channel chunks in double-buffered LDS with ONE barrier per chunk: a wave issues the next
// chunk's loads, multiplies this chunk, transforms + writes the next chunk into the other buffer, barrier.  MI computing tiles of 32 rows per wave.
template <int MI, int PH, int NLD, int NB, int OCC>
__global__ __launch_bounds__(256, OCC) void k16u(float* out, const float4* ga, const float4* gb, int chunks, size_t gmask) {
    constexpr int AR = 128 * MI + 32, T16 = (AR + BROWS) * LD16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * T16; i += blockDim.x) lds[i] = __int_as_float(0x3f803f80 + (i * 2654435761u >> 20));
    __syncthreads();
    floatx16 acc[MI][PH];
    for (int m = 0; m < MI; ++m) for (int p = 0; p < PH; ++p) for (int r = 0; r < 16; ++r) acc[m][p][r] = 0.f;
    size_t off = ((size_t)blockIdx.x * 7919 * 4096) & gmask;
    stage16<NLD, NB, 256, AR>(lds, ga, gb, off, tid, 1.01f, 0.01f);
    off = (off + NLD * 256) & gmask;
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        const float* cur = lds + (c & 1) * T16;
        float* nxt = lds + ((c + 1) & 1) * T16;
        float4 x[NLD], w[NB];
#pragma unroll
        for (int i = 0; i < NLD; ++i) x[i] = ga[off + (size_t)i * 256 + tid];
#pragma unroll
        for (int i = 0; i < NB; ++i) w[i] = gb[(off & 0xffff) + (size_t)i * 256 + tid];
#pragma unroll
        for (int m = 0; m < MI; ++m) multiply16<PH, AR>(cur, acc[m], wave * MI + m, lane);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            float v[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
            unsigned hi[2], mid[2], lo[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float a = fmaf(v[2 * p], 1.01f, 0.01f), b = fmaf(v[2 * p + 1], 1.01f, 0.01f);
                a = fmaxf(a, 0.1f * a); b = fmaxf(b, 0.1f * b);
                hi[p] = pk_bf16(a, b);
                const float ra = a - __uint_as_float(hi[p] << 16), rb = b - __uint_as_float(hi[p] & 0xffff0000u);
                mid[p] = pk_bf16(ra, rb);
                lo[p] = pk_bf16(ra - __uint_as_float(mid[p] << 16), rb - __uint_as_float(mid[p] & 0xffff0000u));
            }
            const int e = (i * 256 + tid) * 4;
            float* row = nxt + (e >> 4) % AR * LD16 + ((e & 15) >> 1);
            *reinterpret_cast<uint2*>(row) = make_uint2(hi[0], hi[1]);
            *reinterpret_cast<uint2*>(row + 8) = make_uint2(mid[0], mid[1]);
            *reinterpret_cast<uint2*>(row + 16) = make_uint2(lo[0], lo[1]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(nxt + AR * LD16 + ((i * 256 + tid) * 4) % (BROWS * LD16 - 4)) = w[i];
        off = (off + NLD * 256) & gmask;
        __syncthreads();
    }
    float sum = 0;
    for (int m = 0; m < MI; ++m) for (int p = 0; p < PH; ++p) for (int r = 0; r < 16; ++r) sum += acc[m][p][r];
    out[blockIdx.x * 256 + tid] = sum;
}
template <int MI, int PH, int NLD, int NB, int OCC>
void run16u(float* d, const float4* ga, const float4* gb, size_t gmask, const char* name) {
    constexpr int AR = 128 * MI + 32, T16 = (AR + BROWS) * LD16;
    const size_t lds = (size_t)2 * T16 * 4;
    (void)hipFuncSetAttribute((const void*)k16u<MI, PH, NLD, NB, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * OCC * 4, chunks = 192;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k16u<MI, PH, NLD, NB, OCC><<<blocks, 256, lds>>>(d, ga, gb, 8, gmask);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k16u<MI, PH, NLD, NB, OCC><<<blocks, 256, lds>>>(d, ga, gb, chunks, gmask); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    const double mfma = (double)blocks * 4 * MI * chunks * PH * 6;
    const double tf = mfma * 2.0 * 32 * 32 * 16 / ms / 1e9;
    printf("%-86s %7.1f bf16 TFLOP/s = %4.1f %% of 2500 (%.2f ms, %zu KB LDS)%s\n", name, tf, tf / 25.0, ms, lds / 1024, err ? " ERROR" : "");
}
template <int NCW, int NPW, int PH, int NLD, int NB, int OCC>
void run16(float* d, const float4* ga, const float4* gb, size_t gmask, const char* name) {
    constexpr int AR = 32 * NCW + 32, T16 = (AR + BROWS) * LD16;
    const size_t lds = (size_t)2 * T16 * 4;
    (void)hipFuncSetAttribute((const void*)k16<NCW, NPW, PH, NLD, NB, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * OCC * 4, chunks = 192;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k16<NCW, NPW, PH, NLD, NB, OCC><<<blocks, (NCW + NPW) * 64, lds>>>(d, ga, gb, 8, gmask);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k16<NCW, NPW, PH, NLD, NB, OCC><<<blocks, (NCW + NPW) * 64, lds>>>(d, ga, gb, chunks, gmask); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    const double mfma = (double)blocks * NCW * chunks * PH * 6;
    const double tf = mfma * 2.0 * 32 * 32 * 16 / ms / 1e9;
    printf("%-86s %7.1f bf16 TFLOP/s = %4.1f %% of 2500 (%.2f ms, %zu KB LDS)%s\n", name, tf, tf / 25.0, ms, lds / 1024, err ? " ERROR" : "");
}

int main() {
    float* d; (void)hipMalloc(&d, 256 * 16 * 256 * 4);
    const size_t nflt4 = (size_t)1 << 26;                      // 1 GiB of activations: HBM, not L2
    float4 *ga, *gb; (void)hipMalloc(&ga, (nflt4 + 65536) * 16); (void)hipMalloc(&gb, (size_t)(1 << 17) * 16);
    (void)hipMemset(ga, 0x3c, (nflt4 + 65536) * 16); (void)hipMemset(gb, 0x3c, (size_t)(1 << 17) * 16);
    const size_t gmask = nflt4 - 1 - 8 * 256;
    // per chunk and computing wave: 4 phases x 2 k-steps x 6 terms = 48 MFMAs (deconv_tile's 32-channel chunk); staging thread: 5 float4 activations + 6 float4 weights
    run<0, 4, 2, 5, 6, 2>(d, ga, gb, gmask & ~(size_t)255, "present structure   (4 waves stage + multiply, 2 barriers/chunk), 2 WG/CU");
    run<1, 4, 2, 5, 6, 1>(d, ga, gb, gmask & ~(size_t)255, "producer / consumer (4 + 4 waves, 1 barrier/chunk, double-buffered LDS), 1 WG/CU");
    run<0, 4, 2, 0, 0, 2>(d, ga, gb, gmask & ~(size_t)255, "present structure, NO staging work (loop + barriers only), 2 WG/CU");
    run<1, 4, 2, 0, 0, 1>(d, ga, gb, gmask & ~(size_t)255, "producer / consumer, NO staging work, 1 WG/CU");
    run<0, 4, 4, 10, 12, 1>(d, ga, gb, gmask & ~(size_t)255, "present structure, chunk twice as long (96 MFMAs), 1 WG/CU");
    run<1, 4, 4, 10, 12, 1>(d, ga, gb, gmask & ~(size_t)255, "producer / consumer, chunk twice as long (96 MFMAs; LDS as above), 1 WG/CU");
    run<1, 4, 2, 3, 3, 1>(d, ga, gb, gmask & ~(size_t)255, "producer / consumer, 60 % of the staging work, 1 WG/CU");
    run2<2, 4, 2, 5, 6, 2>(d, ga, gb, gmask & ~(size_t)255, "present structure, PRE-SPLIT activations (staging = copy, no vector work), 2 WG/CU");
    run2<3, 4, 2, 5, 6, 1>(d, ga, gb, gmask & ~(size_t)255, "pre-split + software-pipelined over chunks (1 barrier/chunk, 2 LDS buffers), 1 WG/CU");
    run2<2, 4, 2, 5, 0, 2>(d, ga, gb, gmask & ~(size_t)255, "present structure, pre-split activations, NO weight staging, 2 WG/CU");
    // 16-channel chunks (24 MFMAs per computing wave and chunk), TWO computing waves per SIMD; a staging thread's share of a chunk scales with 1 / staging threads
    run16<4, 2, 4, 5, 6, 2>(d, ga, gb, gmask & ~(size_t)255, "16-ch chunks: 4 computing + 2 staging waves, 2 WG/CU");
    run16<4, 4, 4, 3, 3, 2>(d, ga, gb, gmask & ~(size_t)255, "16-ch chunks: 4 computing + 4 staging waves, 2 WG/CU");
    run16<8, 4, 4, 5, 3, 1>(d, ga, gb, gmask & ~(size_t)255, "16-ch chunks: 8 computing + 4 staging waves (256-row tile), 1 WG/CU");
    run16<8, 8, 4, 3, 2, 1>(d, ga, gb, gmask & ~(size_t)255, "16-ch chunks: 8 computing + 8 staging waves (256-row tile), 1 WG/CU");
    run16u<1, 4, 3, 3, 2>(d, ga, gb, gmask & ~(size_t)255, "16-ch chunks, every wave stages + multiplies, 1 barrier/chunk (2 LDS buffers), 2 WG/CU");
    run16u<1, 4, 3, 3, 3>(d, ga, gb, gmask & ~(size_t)255, "the same, 3 WG/CU allowed");
    run16u<2, 4, 5, 3, 1>(d, ga, gb, gmask & ~(size_t)255, "the same, TWO 32-row tiles per wave (128 accumulators: half the weight-fragment reads), 1 WG/CU");
    run16u<2, 4, 5, 3, 2>(d, ga, gb, gmask & ~(size_t)255, "the same, two tiles per wave, 2 WG/CU");
    run16<4, 2, 4, 0, 0, 2>(d, ga, gb, gmask & ~(size_t)255, "16-ch chunks: 4 + 2 waves, NO staging work, 2 WG/CU");
    run16<8, 4, 4, 0, 0, 1>(d, ga, gb, gmask & ~(size_t)255, "16-ch chunks: 8 + 4 waves, NO staging work, 1 WG/CU");
    return 0;
}
