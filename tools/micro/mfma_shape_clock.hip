// Does the clock the chip holds under bf16 MFMA load depend on the MFMA shape? Same FLOPs, same LDS traffic pattern class, random operands:
//   A  v_mfma_f32_32x32x16_bf16: wave tile 32 x 64 per k16: 1 A + 2 B fragment reads (ds_read_b128), 2 MFMAs (32 cycles each)
//   B  v_mfma_f32_16x16x32_bf16: wave tile 32 x 64 per k32: 2 A + 4 B fragment reads, 8 MFMAs (16 cycles each) — same FLOP per byte of LDS traffic
// 16 waves per CU (4 per SIMD), every CU busy, operands re-read from a 32 KB LDS image of random bf16. Prints time, TFLOP/s and the s_memtime rate.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape_clock.hip -o tools/micro/bin/mfma_shape_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ unsigned long long g_clk[8][4];

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void mfma_loop(const f4 *src, float *sink, int iters)
{
    __shared__ __attribute__((aligned(16))) f4 lds[2048];              // 32 KB
    for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = src[i];
    __syncthreads();
    const bool clk = blockIdx.x < 8 && threadIdx.x == 0;
    if (clk) { g_clk[blockIdx.x][0] = __builtin_amdgcn_s_memtime(); g_clk[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime(); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float total = 0.f;
    if (SHAPE == 0) {
        f16v acc0, acc1;
        for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {                               // one k64: 4 x (3 reads, 2 MFMAs)
                const int o = (it * 4 + g) * 7 + wave * 64;
                const f4 a = lds[(o + lane) & 2047], b0 = lds[(o + 64 + lane) & 2047], b1 = lds[(o + 128 + lane) & 2047];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b0), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b1), acc1, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; r++) total += acc0[r] + acc1[r];
    } else {
        f4 acc[2][4];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) acc[i][j] = f4{ 0.f, 0.f, 0.f, 0.f };
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int g = 0; g < 2; g++) {                               // one k64: 2 x (6 reads, 8 MFMAs)
                const int o = (it * 2 + g) * 7 + wave * 64;
                f4 a[2], b[4];
#pragma unroll
                for (int i = 0; i < 2; i++) a[i] = lds[(o + i * 64 + lane) & 2047];
#pragma unroll
                for (int j = 0; j < 4; j++) b[j] = lds[(o + 128 + j * 64 + lane) & 2047];
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a[i]), __builtin_bit_cast(bf8, b[j]), acc[i][j], 0, 0, 0);
            }
        }
        for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) total += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
    }
    if (clk) { g_clk[blockIdx.x][1] = __builtin_amdgcn_s_memtime(); g_clk[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime(); }
    if (total == 123.456f) sink[threadIdx.x] = total;
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    f4 *src; float *sink;
    CK(hipMalloc(&src, 32768)); CK(hipMalloc(&sink, 4096));
    std::vector<unsigned short> h(16384);
    srand(1);
    for (auto &v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    CK(hipMemcpy(src, h.data(), 32768, hipMemcpyHostToDevice));
    const int iters = 4000;                                             // k64 steps per wave
    const double flops = 512.0 * 8 * iters * (2.0 * 32 * 64 * 64);      // 512 workgroups x 8 waves x iters x (32 x 64 x k64)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 6; rep++)
        for (int shape = 0; shape < 2; shape++) {
            CK(hipEventRecord(e0, st));
            if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(512), dim3(512), 0, st, src, sink, iters);
            else hipLaunchKernelGGL(mfma_loop<1>, dim3(512), dim3(512), 0, st, src, sink, iters);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long c[8][4];
            CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c)));
            double g = 0;
            for (int i = 0; i < 8; i++) g += (double)(c[i][1] - c[i][0]) / ((double)(c[i][3] - c[i][2]) / 100e6) / 1e9 / 8;
            printf("%s  %.3f ms  %7.0f TFLOP/s  s_memtime rate %.2f GHz (mean of 8 XCDs)\n", shape == 0 ? "32x32x16" : "16x16x32", ms, flops / ms / 1e9, g);
        }
    return 0;
}
