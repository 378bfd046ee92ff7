// fp32 counterpart of mfma_shape_clock.hip: v_mfma_f32_32x32x2_f32 (64 cycles) against v_mfma_f32_16x16x4_f32 (32 cycles) on the same FLOPs and LDS bytes,
// 16 waves per CU, random operands. A wave tile of 32 x 32 per 8 k: shape A: 1 A + 1 B float4 read, 4 MFMAs; shape B: (2 + 2 reads, 8 MFMAs) per 16 k.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape_clock_f32.hip -o tools/micro/bin/mfma_shape_clock_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ unsigned long long g_clk[8][4];

template <int SHAPE>
__global__ __launch_bounds__(256, 4) void mfma_loop(const f4 *src, float *sink, int iters)
{
    __shared__ __attribute__((aligned(16))) f4 lds[2048];              // 32 KB
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = src[i];
    __syncthreads();
    const bool clk = blockIdx.x < 8 && threadIdx.x == 0;
    if (clk) { g_clk[blockIdx.x][0] = __builtin_amdgcn_s_memtime(); g_clk[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime(); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float total = 0.f;
    if (SHAPE == 0) {
        f16v acc;
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {                               // k32: 4 x (2 reads, 4 MFMAs of 32x32x2)
                const int o = (it * 4 + g) * 7 + wave * 64;
                const f4 a = lds[(o + lane) & 2047], b = lds[(o + 64 + lane) & 2047];
#pragma unroll
                for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; r++) total += acc[r];
    } else {
        f4 acc[2][2];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) acc[i][j] = f4{ 0.f, 0.f, 0.f, 0.f };
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int g = 0; g < 2; g++) {                               // k32: 2 x (4 reads, 16 MFMAs of 16x16x4)
                const int o = (it * 2 + g) * 7 + wave * 64;
                f4 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; i++) { a[i] = lds[(o + i * 64 + lane) & 2047]; b[i] = lds[(o + 128 + i * 64 + lane) & 2047]; }
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
            }
        }
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) total += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
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
    std::vector<float> h(8192);
    srand(1);
    for (auto &v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    CK(hipMemcpy(src, h.data(), 32768, hipMemcpyHostToDevice));
    const int iters = 2000;                                             // k32 steps per wave
    const double flops = 1024.0 * 4 * iters * (2.0 * 32 * 32 * 32);     // 1024 workgroups x 4 waves x iters x (32 x 32 x k32)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 6; rep++)
        for (int shape = 0; shape < 2; shape++) {
            CK(hipEventRecord(e0, st));
            if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(1024), dim3(256), 0, st, src, sink, iters);
            else hipLaunchKernelGGL(mfma_loop<1>, dim3(1024), dim3(256), 0, st, src, sink, iters);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long c[8][4];
            CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c)));
            double g = 0;
            for (int i = 0; i < 8; i++) g += (double)(c[i][1] - c[i][0]) / ((double)(c[i][3] - c[i][2]) / 100e6) / 1e9 / 8;
            printf("%s  %.3f ms  %7.1f TFLOP/s  s_memtime rate %.2f GHz (mean of 8 XCDs)\n", shape == 0 ? "32x32x2 " : "16x16x4 ", ms, flops / ms / 1e9, g);
        }
    return 0;
}
